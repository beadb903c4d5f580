#!/usr/bin/env python
"""Benchmark of the render hot path on MI355X.

    python bench.py [--workload cornell|dragon|matpreview-rc|matpreview-rd|volumetric]
                    [--gpus N] [--steps K] [--warmup W] [--weak] [--no-also]

A "step" is one complete render of the workload's frame (all pixels, all spp) with the scene already
resident in HBM.  At N = 1 a step is the BLOCKING drop-in call, `mcpt_renderer_draw` = csrt::Renderer::Draw
(t_render includes the final device-to-host copy of the frame; the host buffer is
pinned).  The workloads are the configurations BASELINE.json names, at their stated film sizes
(monte-carlo-path-tracing_amd/workloads.py); the default is configs[1], cornell-box 512x512 spp 256, and the
default N = 1 line carries an `also` block with north_star's SECOND target, dragon/scene.xml 1280x720 spp 256,
measured the same way in the same process (value, ms_per_step, first_draw_ms, roofline with counters,
cpu_baseline, parity).  For N > 1 (launched by torch.distributed.run, one rank per GPU) the frame's 8x8 tiles
are dealt round-robin to the ranks, every rank renders its tiles into a packed device buffer, ONE RCCL gather
brings them to rank 0, which scatters them into the frame and copies it to the host — all inside the timed
region.  The film is the workload's own for every N: the total work is fixed ("strong" scaling).  `--weak`
(cornell only) grows the film side with sqrt(N) instead.

Rank 0 prints ONE JSON line: metric Msamples/s (W*H*spp / t / 1e6, whole job) plus
  first_draw_ms what the FIRST draw of a fresh renderer costs (buffer allocation and, for scenes outside
                LDS, the calibration of the kernel configuration): what a one-shot `mcpt_cli` run pays.
  roofline      the render kernel against the roof that binds it.  Counts per sample come from the kernel's
                counting instantiation (run outside the timed region), the kernel time from HIP events on the
                launch stream, hardware counters from rocprofv3 --pmc passes that bench.py runs on the same
                workload right after the timed region (one pass per counter group, --kernel-trace only), the
                HBM stream bandwidth from a device copy / reduction timed in this process.  `valu` holds BOTH
                readings of the VALU roof: `frac_at_16_lanes_per_clk` (SQ's accounting: a wave64 instruction
                takes 4 issue cycles) and `frac_at_32_lanes_per_clk` (the guide's: 2 cycles), and — where the
                counters exist — the instruction-class mix that decides between them.
  cpu_baseline  at N = 1: the compiled reference (oracle/_ref, "reference") and the oracle port ("port") timed
                on this host's cores on a bounded sample of the same workload, and `parity`: the GPU frame
                against that CPU frame at the same film / spp.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_SPEC_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E, 8 TB/s
# VALU roof, the guide's reading (MI355X_MICROARCH.md:52-54): a wave64 VALU instruction issues over 2 cycles = 32 lanes
# per clock and SIMD; 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T fp32 lane-operations / s.  `roofline.frac` is quoted
# against THAT.  SQ's own accounting (4 issue cycles per wave64 instruction = 16 lanes per clock) and the reading by this
# repository's measured per-class issue costs stay in the line as extra keys.
VALU_LANES_PER_SIMD_GUIDE = 32
VALU_LANES_PER_SIMD_SQ = 16
GUIDE_CLOCK_GHZ = 2.4
N_XCD = 8

PMC_GROUPS = [
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
     "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    # instruction-class mix (optional: a box whose rocprofv3 lacks one of these reports the pass under `failed`)
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
     "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"],
]
# memory side, scenes outside LDS only (round 5): vector-cache and L2 hit rates, the cycles the vector cache spends stalled on
# pending misses, the L2's requests to the fabric — what says whether a mesh kernel waits for bandwidth, for a queue or for latency
PMC_GROUPS_MEMORY = [
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_REQ_sum"],
]
# LDS side, kernels whose traversal data and ray records live in LDS only (round 6; VERDICT round 5, task 4): how many of the
# wave-cycles wait for an LDS instruction, and how much of the LDS's active time is bank conflicts (12-word ray records read with
# ds_read_b128 at data-dependent record indices).  Two passes; a box whose rocprofv3 lacks a counter reports the pass under `failed`.
PMC_GROUPS_LDS = [
    ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"],
]
KERNEL_WORDS = ("render_kernel", "sorted_kernel", "stream_kernel", "primary_kernel", "queued_")
# Issue cost per wave64 VALU instruction by class, from this repository's microbenchmark on MI355X
# (tools/microbench/valu_rate.hip, profiles/r02_experiments/valu_issue_rate_microbench.jsonl): plain fp32
# add / sub / mul, moves and bit operations issue every ~2.7 cycles per SIMD, compares / selects / min / max / fma /
# conversions / integer multiplies every ~4.5, transcendentals (rcp, sqrt, ...) every ~8.5.
CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.7, "SQ_INSTS_VALU_MUL_F32": 2.7, "SQ_INSTS_VALU_FMA_F32": 4.5,
                "SQ_INSTS_VALU_TRANS_F32": 8.5, "SQ_INSTS_VALU_INT32": 2.7, "SQ_INSTS_VALU_CVT": 4.5}


def algorithmic_bytes_per_sample(counts, spp):
    """32 B per box test (the ordered walk reads one 64-byte record per node visit and
    tests its two boxes), 36 B per primitive test, 132 B of attributes per shaded hit, 12 B/spp for the
    pixel store.  Path state stays in registers / LDS (and, for the stream kernel's shadow rays on meshes,
    in a cache-resident scratch region), so the wavefront-state term is 0."""
    s = float(counts["samples"])
    return (counts["node_tests"] * 32.0 + counts["prim_tests"] * 36.0 +
            counts["shaded_hits"] * 132.0) / s + 12.0 / spp


def stream_bandwidth(torch, device):
    """HBM stream rates of this GPU, measured here: a 1 GiB device copy (read + write) and a 2 GiB
    reduction (read only).  GB/s."""
    n = 1 << 28  # floats = 1 GiB
    a = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    b = torch.empty_like(a)
    out = {}

    def timed(fn, nbytes, reps=8):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out["copy_gbs"] = timed(lambda: b.copy_(a), 8.0 * n)
    big = torch.cat([a, b])
    out["read_gbs"] = timed(lambda: big.sum(), 4.0 * big.numel())
    del a, b, big
    torch.cuda.empty_cache()
    return out


def pmc_leg(workload, film, choice, timeout_s=300, memory_side=False, lds_side=False):
    """Hardware counters of the render kernel for one full-size launch of the workload: rocprofv3 runs
    tools/render_scene.py (one draw) once per counter group.  Returns None when rocprofv3 is not there."""
    if shutil.which("rocprofv3") is None:
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    counters, kernels, failed, duration_ns = {}, set(), [], []
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for gi, group in enumerate(PMC_GROUPS + (PMC_GROUPS_MEMORY if memory_side else []) + (PMC_GROUPS_LDS if lds_side else [])):
            d = os.path.join(tmp, f"pass{gi}")
            target = [sys.executable, os.path.join(ROOT, "tools", "render_scene.py"), f"workload:{workload}",
                      "--film", *map(str, film), "--draws", "1", "--kernel-mode", str(choice[0]), "--work", str(choice[1]),
                      "--prepass", str(choice[2])]   # the timed run's (calibrated) choice, made explicit: no calibration launches
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *group, "--output-format", "csv", "-d", d, "-o", "p",
                   "--", *target]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp")
            except subprocess.TimeoutExpired:
                failed.append({"group": group, "error": "timeout"})
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                failed.append({"group": group, "rc": r.returncode, "stderr": r.stderr[-300:]})
                continue
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"]
                if not any(k in name for k in KERNEL_WORDS):
                    continue
                kernels.add(name.replace("(anonymous namespace)::", "").split("(")[0][-90:])
                counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if any(k in row["Kernel_Name"] for k in KERNEL_WORDS):
                        if gi == 0:
                            duration_ns.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    # one draw = [camera-ray pre-pass +] render kernel: the step's kernel time is their sum
    return {"counters": counters, "kernels": sorted(kernels), "failed": failed,
            "kernel_ns": float(np.sum(duration_ns)) if duration_ns else None}


def cpu_baseline(pkg, workload, W, H, SPP, budget_s=12.0):
    """Times both CPU checkers on a bounded sample of the workload (its own film when the budget buys at
    least 2 spp there, otherwise a quarter-size film) and returns (records, mcsd path dir, frame, film)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import checkers
    cores = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(dir="/tmp")
    oracle = checkers.Oracle()

    def scene_file(w, h, spp):
        path = os.path.join(tmp, f"s_{w}x{h}_{spp}.mcsd")
        pkg.workloads.config(workload, w, h, spp).save_mcsd(path)
        return path

    pw, ph = max(W // 4 // 8 * 8, 8), max(H // 4 // 8 * 8, 8)
    _, info = oracle.render(scene_file(pw, ph, 2))
    rate = pw * ph * 2 / max(info["seconds"], 1e-6)
    w, h = (W, H) if budget_s * rate / (W * H) >= 2.0 else (pw, ph)
    spp = int(max(1, min(SPP, budget_s * rate / (w * h))))
    path = scene_file(w, h, spp)
    recs = {}
    frame, info = oracle.render(path)
    samples = w * h * spp
    sample = f"{pkg.workloads.DESCRIPTION[workload].split(' (')[0].rsplit(' ', 2)[0]} at {w}x{h} spp={spp} " \
             f"({samples / 1e6:.1f} Msamples, all host threads)"
    recs["port"] = {"value": samples / info["seconds"] / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                    "seconds": info["seconds"], "sample": sample}
    if checkers.reference_available():
        # the compiled reference prints its progress under a mutex (renderer.cpp:700-715), which limits
        # how far it scales over host threads: it gets a third of the port's sample count
        rspp = max(1, spp // 3)
        rpath = scene_file(w, h, rspp)
        rframe, rinfo = checkers.Reference().render(rpath, w, h)
        recs["reference"] = {"value": w * h * rspp / rinfo["seconds"] / 1e6, "unit": "Msamples/s", "cores": cores,
                             "kind": "reference", "seconds": rinfo["seconds"],
                             "sample": sample.replace(f"spp={spp} ", f"spp={rspp} ").replace(
                                 f"({samples / 1e6:.1f}", f"({w * h * rspp / 1e6:.1f}")}
        if rspp == spp:
            recs["reference"]["equals_port_frame"] = bool(np.array_equal(rframe, frame))
    return recs, frame, (w, h, spp)


def measure(pkg, torch, dist, args, name, film, world, rank, local_rank, device, primary):
    """One workload, measured: the timed loop, the throughput-mode loop (primary line only), roofline, CPU baseline
    and parity.  Returns the JSON record on rank 0 (None elsewhere)."""
    W, H, SPP = film
    weak = args.weak and world > 1
    use_gather = world > 1 or args.force_gather
    kernel_mode = {"auto": -1, "stream": 1, "lanes": 0, "queued": 5}[args.kernel]

    def make_renderer(spp):
        r = pkg.capi.Renderer(pkg.workloads.config(name, W, H, spp), device=local_rank)
        r.set_kernel(kernel_mode)
        if args.rng != "reference":
            r.set_rng({"pcg": 1, "sobol": 2}[args.rng], seed=1, sample_split=args.sample_split)
        return r

    renderer = make_renderer(SPP)
    rng = pkg.capi.TileRange(rank, world, 0)
    assert renderer.tiles_in(rng) == len(pkg.tiling.rank_tiles(rank, world, W, H))
    stream = torch.cuda.current_stream().cuda_stream
    # the caller's frame: host memory (pinned), like the buffer csrt::RayTracer hands to Renderer::Draw
    host = torch.empty(H * W * 3, dtype=torch.float32, pin_memory=True)
    host_np = host.numpy().reshape(H, W, 3)
    fg = pkg.tiling.FrameGather(world, rank, W, H, device) if use_gather else None
    kernel_ms_steps = []

    def step(record=False):
        if not use_gather:
            st = renderer.draw_into(host_np)   # blocking; device-to-host copy of the frame included
            if record:
                kernel_ms_steps.append(st["kernel_milliseconds"])
        else:
            renderer.draw_device(fg.packed.data_ptr(), rng, packed=True, stream=stream, blocking=False)
            fg.gather()
            if rank == 0:
                host.copy_(fg.frame, non_blocking=True)

    def sync():
        if use_gather:
            dist.barrier()
        torch.cuda.synchronize()

    # First draw of a fresh renderer: buffer allocation and — scenes outside LDS, library's choice of kernel — the
    # calibration of the kernel configuration.  Reported (`first_draw_ms`), not part of a step.
    sync()
    t_first = time.perf_counter()
    step()
    sync()
    first_draw_ms = 1e3 * (time.perf_counter() - t_first)
    for _ in range(args.warmup):
        step()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step(record=True)
    ev1.record()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # HIP events on the launch stream: inside the library around every draw's launches (N = 1), around the loop (N > 1)
    kernel_ms = float(np.mean(kernel_ms_steps)) if kernel_ms_steps else ev0.elapsed_time(ev1) / max(args.steps, 1)
    kernel_name = renderer.last_kernel()
    choice = renderer.last_choice()   # (kernel, work distribution, pre-pass) the timed steps ran
    gather_ms = None
    if use_gather:
        # the exchange step alone: one more gather of the blocks already rendered
        sync()
        g0 = time.perf_counter()
        fg.gather()
        sync()
        gather_ms = 1e3 * (time.perf_counter() - g0)

    # The same job in the THROUGHPUT mode (independent PCG-hashed stream per (pixel, sample), the samples of a
    # pixel spread over lanes): reported next to `value`, never as `value`.  With the reference stream a pixel's
    # samples are one sequential chain, so a GPU whose tile share is below one pixel per lane cannot go faster
    # than one wavefront's chain; this mode has no such floor.
    throughput = None
    if args.rng == "reference" and not args.no_throughput_mode and primary:
        renderer.set_rng(1, seed=1, sample_split=args.sample_split)
        for _ in range(max(args.warmup, 1)):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        e2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([e2], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        throughput = {"value": W * H * SPP * args.steps / e2 / 1e6, "unit": "Msamples/s", "ms_per_step": 1e3 * e2 / args.steps,
                      "rng": "independent PCG-hashed stream per (pixel, sample); not per-pixel comparable with the reference",
                      "kernel": renderer.last_kernel()}
        renderer.set_rng(0)

    # per-rank kernel time of its share (N > 1): one more blocking draw on every rank, gathered to rank 0
    rank_kernel_ms = None
    if world > 1:
        st = renderer.draw_device(fg.packed.data_ptr(), rng, packed=True, stream=stream, blocking=True)
        mine = torch.tensor([st["kernel_milliseconds"]], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_kernel_ms = [float(v.item()) for v in every]
        kernel_ms = rank_kernel_ms[0]

    # N > 1: the line verifies itself — the communicator's own count of its ranks, and the gathered frame against the frame ONE
    # renderer draws alone (rank 0, outside the timed region), bit for bit
    rccl, frame_check = None, None
    if use_gather:
        rccl = pkg.tiling.communicator_report(device)
        step()
        sync()
        single = renderer.draw()[0] if rank == 0 else None
        frame_check = pkg.tiling.self_check(fg, single)
        sync()

    out = None
    if rank == 0:
        samples = W * H * SPP
        value = samples * args.steps / elapsed / 1e6
        out = {
            "metric": "Msamples/sec (W*H*spp/s)", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "first_draw_ms": first_draw_ms,
            "config": {"workload": f"{name}: {pkg.workloads.DESCRIPTION[name]}"
                                   + ("" if (W, H, SPP) == pkg.workloads.WORKLOADS[name][1] else f" — film overridden: {W}x{H} spp={SPP}"),
                       "baseline_config_index": pkg.workloads.WORKLOADS[name][2],
                       "step": ("blocking mcpt_renderer_draw into a pinned host frame (device-to-host copy included)" if not use_gather else
                                "per rank: asynchronous draw of its tiles + ONE gather; rank 0: scatter + device-to-host copy of the frame"),
                       "rng": ("reference stream (Tea + LCG per pixel)" if args.rng == "reference" else
                               "THROUGHPUT MODE, not per-pixel comparable: independent PCG-hashed stream per (pixel, sample)" if args.rng == "pcg" else
                               "THROUGHPUT MODE, not per-pixel comparable: Owen-scrambled Sobol points per (pixel, sample)"),
                       "kernel": kernel_name,
                       "partition": f"8x8 tiles round-robin over {world} GPU(s)"
                                    + (", one RCCL gather to rank 0" if world > 1 else ""),
                       "film": (f"{W}x{H}: side scaled with sqrt(N) so that every GPU renders 512x512 pixels' worth "
                                "of tiles (weak scaling)") if weak else f"{W}x{H}"},
        }
        if rccl is not None:
            out["rccl"] = rccl
            out["frame_check"] = frame_check
        if rank_kernel_ms is not None:
            out["per_rank_kernel_ms"] = rank_kernel_ms
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
        if throughput:
            out["throughput_mode"] = throughput
        # ---- roofline of the render kernel (everything below is outside the timed region).  N > 1: rank 0's
        # GPU and its share of the tiles.
        count_spp = min(SPP, 16)
        rc = make_renderer(count_spp)
        count_choice = choice if choice[0] != 5 else (1, 1, choice[2])   # (the queued renderer has no counting build: its rays are the stream kernel's)
        rc.set_kernel(count_choice[0]).set_work_distribution(count_choice[1]).set_prepass(count_choice[2])
        if args.rng == "sobol":   # (no counting build draws Sobol points: the rays per sample are counted on PCG-hashed streams)
            rc.set_rng(1, seed=1, sample_split=args.sample_split)
        _, counts = rc.draw(counted=True)
        counting_kernel = rc.last_kernel()
        boxes_per_node_step = 4.0 if "pool-walk" in counting_kernel else 2.0   # a node item of the pool walk tests four children
        scene_info = rc.info()
        rc.close()
        rank_samples = samples if world == 1 else len(pkg.tiling.rank_tiles(0, world, W, H)) * 64 * SPP
        b_per_sample = algorithmic_bytes_per_sample(counts, SPP)
        algorithmic_bytes = b_per_sample * rank_samples
        algorithmic_gbs = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9
        props = torch.cuda.get_device_properties(device)
        n_simd = props.multi_processor_count * 4
        clock_ghz = getattr(props, "clock_rate", 2.4e6) / 1e6   # (fallback only: cycles come from GRBM_GUI_ACTIVE)
        bw = stream_bandwidth(torch, device)
        hbm_peak = max(bw["copy_gbs"], bw["read_gbs"])
        rays = counts["closest_rays"] + counts["shadow_rays"]
        per_sample = {k: counts[k] / counts["samples"] for k in
                      ("closest_rays", "shadow_rays", "node_tests", "prim_tests", "shaded_hits")}
        # rays per second next to samples per second: a film whose camera rays mostly miss cannot flatter this one
        out["grays_per_s"] = rays / counts["samples"] * samples * args.steps / elapsed / 1e9
        out["rays_per_sample"] = rays / counts["samples"]
        walk = {"rays_per_s": rays / counts["samples"] * rank_samples / (kernel_ms * 1e-3),
                "node_phase_lane_util": (counts["node_tests"] / boxes_per_node_step) / (64.0 * max(counts["wave_node_steps"], 1)),
                "counting_kernel": counting_kernel,
                "prim_phase_lane_util": counts["prim_tests"] / (64.0 * max(counts["wave_prim_steps"], 1)),
                # (round 5's advisor) what the counts describe: the counting instantiations walk two queries per vertex (they carry the
                # volume-path code, so they are never merged), the production kernels outside LDS merge a vertex's last shadow query with
                # the next segment's closest query — the same rays, boxes and primitives, half the query rounds — and a node step with
                # few items is worked off by 2 / 4 lanes per item: node_phase_lane_util counts ITEMS per step, not busy lanes
                "note": "counted on the two-queries-per-vertex form; node_phase_lane_util = node items per node step / 64"}
        hbm = {"algorithmic_gbs": algorithmic_gbs, "bytes_per_sample": b_per_sample,
               "stream_peak_gbs": hbm_peak, "stream_copy_gbs": bw["copy_gbs"], "stream_read_gbs": bw["read_gbs"],
               "spec_peak_gbs": HBM_SPEC_GBS, "frac_algorithmic_of_stream_peak": algorithmic_gbs / hbm_peak,
               "frac_algorithmic_of_spec": algorithmic_gbs / HBM_SPEC_GBS}
        roof = {"bound": "hbm", "achieved": algorithmic_gbs, "peak": hbm_peak, "unit": "GB/s",
                "frac": algorithmic_gbs / hbm_peak, "traffic": None,
                "kernel": kernel_name, "kernel_ms": kernel_ms, "per_sample": per_sample, "walk": walk, "hbm": hbm,
                "scene": {"walk_nodes": scene_info["walk_nodes"], "primitives": scene_info["primitives"],
                          "geometry_bytes": scene_info["geometry_bytes"]}}
        outside_lds = "+lds" not in kernel_name
        pmc = None if (args.no_pmc or world > 1 or args.rng != "reference") else pmc_leg(name, (W, H, SPP), choice, memory_side=outside_lds, lds_side=not outside_lds)
        if pmc and pmc["counters"].get("SQ_INSTS_VALU") and pmc["kernel_ns"]:
            c = pmc["counters"]
            pmc_ms = pmc["kernel_ns"] * 1e-6
            cycles = c["GRBM_GUI_ACTIVE"] / N_XCD if c.get("GRBM_GUI_ACTIVE") else pmc_ms * 1e-3 * clock_ghz * 1e9
            issue = c["SQ_INSTS_VALU"] * 4.0 / (n_simd * cycles)
            lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
            valu_peak = n_simd * VALU_LANES_PER_SIMD_GUIDE * GUIDE_CLOCK_GHZ * 1e9 / 1e12   # the guide's peak: 78.6 T lane-ops/s
            peak_16_measured_clock = n_simd * VALU_LANES_PER_SIMD_SQ * cycles / (pmc_ms * 1e-3) / 1e12
            achieved_tlaneops = c["SQ_INSTS_VALU"] * 64.0 * lanes / (pmc_ms * 1e-3) / 1e12   # useful fp32 lane-operations / s
            valu = {"issue_frac": issue, "lane_util": lanes, "useful_frac": issue * lanes,
                    "frac_of_guide_peak": achieved_tlaneops / valu_peak,
                    "measured_clock_ghz": cycles / (pmc_ms * 1e-3) / 1e9,
                    # the two readings of the VALU roof: SQ accounts a wave64 instruction at 4 issue cycles (16 lanes / clk /
                    # SIMD); the CDNA4 guide gives 2 cycles (32 lanes / clk / SIMD).  The truth depends on the instruction
                    # mix (this repository's microbenchmark: plain add / mul / mov 2.7, fma / cmp / cndmask / cvt 4.5 cycles).
                    "frac_at_16_lanes_per_clk": issue * lanes, "frac_at_32_lanes_per_clk": 0.5 * issue * lanes,
                    "issue_frac_at_2_cycles": 0.5 * issue,
                    "valu_insts_per_sample": c["SQ_INSTS_VALU"] / samples,
                    "wait_any_per_wave_cycle": c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                    "peak_tlaneops": valu_peak, "peak_tlaneops_at_16_lanes_measured_clock": peak_16_measured_clock,
                    "achieved_tlaneops": achieved_tlaneops,
                    "pmc_kernel_ms": pmc_ms, "cycles": cycles}
            mix = {k: c[k] for k in CLASS_CYCLES if k in c}
            if mix:
                classified = sum(mix.values())
                other = max(c["SQ_INSTS_VALU"] - classified, 0.0)   # compares, selects, min / max, moves, bit operations, packed, fp64
                est_cycles = sum(v * CLASS_CYCLES[k] for k, v in mix.items()) + other * 4.5
                valu["class_mix"] = dict({k.replace("SQ_INSTS_VALU_", "").lower(): v / c["SQ_INSTS_VALU"] for k, v in mix.items()},
                                         other=other / c["SQ_INSTS_VALU"])
                valu["issue_frac_by_class_cycles"] = est_cycles / (n_simd * cycles)
                valu["frac_by_class_cycles"] = est_cycles / (n_simd * cycles) * lanes
                for k in ("SQ_INSTS_SALU", "SQ_INSTS_VMEM"):
                    if k in c:
                        valu[k.lower() + "_per_valu"] = c[k] / c["SQ_INSTS_VALU"]
            roof["valu"] = valu
            if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and c.get("TCC_REQ_sum"):
                # (round 4 priced the mesh kernels against a "ceiling" from a gather microbenchmark; these are the counters instead)
                l2_requests = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
                roof["memory"] = {
                    "tcp_hit_rate": 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
                    "l2_hit_rate": c["TCC_HIT_sum"] / max(l2_requests, 1.0),
                    "l2_requests_per_sample": c["TCC_REQ_sum"] / samples,
                    "l2_fabric_read_bytes": c["TCC_EA0_RDREQ_sum"] * 128.0,   # (128-byte requests on gfx950; FETCH_SIZE tallies them at 64)
                    "l2_fabric_read_gbs": c["TCC_EA0_RDREQ_sum"] * 128.0 / (pmc_ms * 1e-3) / 1e9,
                    "tcp_pending_stall_per_active_cycle": c["TCP_PENDING_STALL_CYCLES_sum"] / max(c.get("TCP_GATE_EN1_sum", 0.0), 1.0),
                    "note": "vector-cache hits include the 3 further 16-byte reads of a 64-byte node record already fetched"}
            if c.get("SQ_INSTS_LDS") and c.get("SQ_LDS_IDX_ACTIVE"):
                wave_cycles = max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
                roof["lds"] = {
                    "lds_insts_per_sample": c["SQ_INSTS_LDS"] / samples,
                    "lds_insts_per_valu": c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"],
                    # of the cycles the LDS is busy with indexed accesses, the share it spends re-issuing conflicting banks
                    "bank_conflict_per_active_cycle": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"],
                    # the LDS pipeline's busy share of the kernel (IDX_ACTIVE counts per CU-cycle; 256 CUs x cycles)
                    "active_frac_of_cu_cycles": c["SQ_LDS_IDX_ACTIVE"] / max((n_simd / 4.0) * cycles, 1.0),
                    "wave_cycles_waiting_for_lds": c.get("SQ_WAIT_INST_LDS", 0.0) / wave_cycles if "SQ_WAIT_INST_LDS" in c else None,
                    "addr_conflict_per_active_cycle": c.get("SQ_LDS_ADDR_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if "SQ_LDS_ADDR_CONFLICT" in c else None,
                    "unaligned_stall_per_active_cycle": c.get("SQ_LDS_UNALIGNED_STALL", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if "SQ_LDS_UNALIGNED_STALL" in c else None,
                    "salu_cycles_per_wave_cycle": c.get("SQ_INST_CYCLES_SALU", 0.0) / wave_cycles if "SQ_INST_CYCLES_SALU" in c else None,
                }
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                # rocprofv3 reports KiB; FETCH_SIZE counts 64-byte requests where gfx950 moves 128 (guide's
                # gfx950 note): doubled
                traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                roof["traffic"] = traffic
                hbm.update(measured_gbs=traffic / (pmc_ms * 1e-3) / 1e9,
                           measured_frac_of_stream_peak=traffic / (pmc_ms * 1e-3) / 1e9 / hbm_peak,
                           measured_frac_of_spec=traffic / (pmc_ms * 1e-3) / 1e9 / HBM_SPEC_GBS,
                           traffic_over_algorithmic=traffic / algorithmic_bytes,
                           fetch_bytes=2.0 * c["FETCH_SIZE"] * 1024.0, write_bytes=c["WRITE_SIZE"] * 1024.0)
            # the roof the kernel is closest to: VALU issue slots or HBM bytes actually moved
            if issue >= hbm.get("measured_frac_of_stream_peak", 0.0):
                roof.update(bound="valu", achieved=achieved_tlaneops, peak=valu_peak, unit="Tlane-op/s",
                            frac=achieved_tlaneops / valu_peak)
                roof["note"] = ("VALU bound: frac = useful fp32 lane-operations per second (VALU wave-instructions x 64 x lane "
                                "utilisation / kernel time) / the guide's peak (%d SIMDs x 32 lanes / clk x 2.4 GHz).  %.0f %% of the "
                                "issue slots are taken by SQ's 4-cycle accounting (%.0f %% at the guide's 2 cycles per instruction), "
                                "%.0f %% of the lanes of those instructions do work; the same quantity at 16 lanes / clk and the "
                                "measured clock is %.2f (`valu.frac_at_16_lanes_per_clk`); no FMA: the arithmetic contract forbids "
                                "contraction.  HBM moves %s per launch against %s of algorithmic bytes: the walk's data comes from "
                                "LDS / cache."
                                % (n_simd, 100 * issue, 50 * issue, 100 * lanes, issue * lanes,
                                   "%.3g MB" % (roof["traffic"] / 1e6) if roof["traffic"] else "?",
                                   "%.3g GB" % (algorithmic_bytes / 1e9)))
            else:
                roof.update(achieved=hbm["measured_gbs"], frac=hbm["measured_frac_of_stream_peak"])
                roof["note"] = ("memory bound: achieved = HBM bytes moved per launch (PMC) / kernel time, peak = stream "
                                "bandwidth measured in this process; VALU issue %.0f %% x lane utilisation %.0f %%"
                                % (100 * issue, 100 * lanes))
            # WHAT LIMITS the kernel, from the counters (the `bound` above names the nearer of the two roofs the contract knows):
            # issue slots, HBM bytes — or neither: wavefronts that wait (dependent steps of a ray query, a frame's tail of long
            # pixel chains) while issue slots and bandwidth are free
            wait = valu["wait_any_per_wave_cycle"]
            hbm_frac = hbm.get("measured_frac_of_stream_peak", 0.0)
            waves_resident = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / max(n_simd * cycles, 1.0)
            lds_wait = (roof.get("lds") or {}).get("wave_cycles_waiting_for_lds") or 0.0
            lds_conflict = (roof.get("lds") or {}).get("bank_conflict_per_active_cycle") or 0.0
            kind = ("valu-issue" if issue >= 0.6 else "hbm" if hbm_frac >= 0.6 else
                    "lds" if (lds_wait >= 0.5 * max(wait, 1e-9) and lds_conflict >= 0.10) else "latency / occupancy")
            roof["limiter"] = {"kind": kind, "valu_issue_frac": issue, "hbm_frac_of_stream_peak": hbm_frac, "wave_cycles_waiting": wait,
                               "mean_wavefronts_per_simd": waves_resident}
            # `bound` follows the limiter (dragon printed "valu" beside "latency / occupancy"): when neither roof the
            # contract knows is what the kernel waits for, `bound` says so, and `nearest_roof` names the roof achieved / peak / frac price
            roof["nearest_roof"] = roof["bound"]
            roof["bound"] = {"valu-issue": "valu", "hbm": "hbm", "lds": "lds"}.get(kind, "latency")
            roof["pmc"] = {"kernels": pmc["kernels"], "failed": pmc["failed"],
                           "command": "rocprofv3 --kernel-trace --pmc <group> -- python tools/render_scene.py "
                                      f"workload:{name} --film {W} {H} {SPP} --draws 1 --kernel-mode {choice[0]} --work {choice[1]} "
                                      f"--prepass {choice[2]} (one pass per group; counters and durations summed over the draw's kernels)"}
        else:
            roof["note"] = ("no counter pass (rocprofv3 absent, --no-pmc or N > 1): algorithmic bytes (32 B/box test, "
                            "36 B/primitive test, 132 B/shaded hit, 12 B/pixel) over the kernel time against the "
                            "measured stream bandwidth")
            if pmc:
                roof["pmc"] = {"failed": pmc["failed"]}
        out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            recs, cpu_frame, (cw, ch, cspp) = cpu_baseline(pkg, name, W, H, SPP, budget_s=12.0 if primary else 8.0)
            out["cpu_baseline"] = recs.get("reference", recs["port"])
            if "reference" in recs:
                out["cpu_baseline_port"] = recs["port"]
            rg = pkg.capi.Renderer(pkg.workloads.config(name, cw, ch, cspp), device=local_rank)
            rg.set_kernel(kernel_mode)
            if args.rng != "reference":
                rg.set_rng({"pcg": 1, "sobol": 2}[args.rng], seed=1, sample_split=args.sample_split)
            gpu_frame, _ = rg.draw()
            rg.close()
            d = gpu_frame.astype(np.float64) - cpu_frame.astype(np.float64)
            l2 = np.sqrt((d ** 2).sum(axis=2))
            out["parity"] = {"vs": "port", "film": [cw, ch, cspp], "rng": args.rng,
                             "mean_gpu": float(gpu_frame.mean()), "mean_cpu": float(cpu_frame.mean()),
                             "rmse": float(np.sqrt((d ** 2).mean())), "mean_l2": float(l2.mean()),
                             "max_l2": float(l2.max()), "frac_gt_1e-3": float((l2 > 1e-3).mean()),
                             "frac_exact": float((l2 == 0).mean())}
    if args.force_gather and primary:
        step()   # (the throughput-mode loop above left its own frame in the gather buffer)
        sync()
        if rank == 0:
            # the gathered frame must be the plain full-frame draw, bit for bit
            plain, _ = renderer.draw()
            same = bool(np.array_equal(fg.frame.cpu().numpy().reshape(H, W, 3), plain))
            print(json.dumps({"force_gather_frame_equals_plain_draw": same}), file=sys.stderr)
            if not same:
                sys.exit(3)
    renderer.close()
    del host
    return out


def compact(full, detail_path):
    """The ONE line rank 0 prints: the contract's keys, the roofline / cpu_baseline / parity objects reduced to their
    numbers, `also.dragon` the same way — under 2000 bytes, so that a driver that keeps the tail of stdout keeps all of
    it.  Everything measured (per-sample counts, counter readings, notes) goes to `detail_path`."""
    def roof(r):
        v, h = r.get("valu", {}), r.get("hbm", {})
        o = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic") if k in r}
        if o.get("traffic") is not None:
            o["traffic"] = int(o["traffic"])
        o.update(kernel=r["kernel"][:40], kernel_ms=round(r["kernel_ms"], 3),
                 algorithmic_bytes=round(h["bytes_per_sample"], 1), algorithmic_gbs=round(h["algorithmic_gbs"], 1))
        if v:
            o.update(lane_util=round(v["lane_util"], 3), issue_frac_sq=round(v["issue_frac"], 3),
                     frac_at_16_lanes=round(v["frac_at_16_lanes_per_clk"], 3))
        if "limiter" in r:
            o["limiter"] = r["limiter"]["kind"]
            o["waiting"] = round(r["limiter"]["wave_cycles_waiting"], 2)
            o["waves_per_simd"] = round(r["limiter"]["mean_wavefronts_per_simd"], 2)
        if "nearest_roof" in r:
            o["nearest_roof"] = r["nearest_roof"]
        if r.get("lds"):
            o["lds_conflict"] = round(r["lds"]["bank_conflict_per_active_cycle"], 3)
            if r["lds"].get("wave_cycles_waiting_for_lds") is not None:
                o["lds_wait"] = round(r["lds"]["wave_cycles_waiting_for_lds"], 3)
        if "memory" in r:
            o["tcp_hit"] = round(r["memory"]["tcp_hit_rate"], 3)
            o["l2_hit"] = round(r["memory"]["l2_hit_rate"], 3)
        for k in ("achieved", "peak", "frac"):
            o[k] = float("%.4g" % o[k])
        return o

    def one(m):
        o = {k: m[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data", "grays_per_s", "rays_per_sample",
                               "first_draw_ms", "gather_ms", "per_rank_kernel_ms", "rccl") if k in m}
        if m.get("frame_check"):
            o["frame_check"] = {"equals_single_gpu_frame": m["frame_check"]["equals_single_gpu_frame"],
                                "sha256": m["frame_check"]["gathered_frame_sha256"][:16]}
        for k in ("value", "ms_per_step", "grays_per_s", "rays_per_sample", "first_draw_ms"):
            if k in o:
                o[k] = round(o[k], 3)
        c = m["config"]
        o["config"] = {"workload": c["workload"], "baseline_config_index": c["baseline_config_index"],
                       "rng": c["rng"].split(" (")[0]}
        if "roofline" in m:
            o["roofline"] = roof(m["roofline"])
        if "cpu_baseline" in m:
            b = m["cpu_baseline"]
            o["cpu_baseline"] = {"value": round(b["value"], 3), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"],
                                 "sample": b["sample"]}
        if "parity" in m:
            q = m["parity"]
            o["parity"] = {"film": q["film"], "rmse": q["rmse"], "max_l2": q["max_l2"], "frac_exact": q["frac_exact"]}
        if "throughput_mode" in m:
            o["throughput_mode"] = {"value": round(m["throughput_mode"]["value"], 1), "rng": "pcg, not per-pixel comparable"}
        return o

    line = one(full)
    if "also" in full:
        a = one(full["also"]["dragon"])
        for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            a.pop(k, None)
        a["config"] = {"workload": a["config"]["workload"].split(" (")[0] + " (stand-ins for the 4 unshipped OBJ files)"}
        a.pop("throughput_mode", None)
        for k in ("algorithmic_bytes", "algorithmic_gbs", "frac_at_16_lanes", "nearest_roof"):  # (in the detail file; the line stays under 2000 bytes)
            a.get("roofline", {}).pop(k, None)
        if "cpu_baseline" in a:
            a["cpu_baseline"].pop("unit")
            a["cpu_baseline"]["sample"] = a["cpu_baseline"]["sample"].split(" (")[0]
        line["also"] = {"dragon": a}
    line["detail"] = detail_path
    return line


def cpu_tiles_line(pkg, args):
    """`--cpu-tiles` (the CPU test of the N > 1 code path, tests/test_multi_gpu_host.py): the same partition, packed blocks, gather,
    communicator report and frame self-check as a `--gpus N` run, with gloo as the backend and the rank's tiles written by the
    product's HOST build of the kernel body (libmcpt_host.so, mcpt_host_render_tiles) — no GPU, no timing claim."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist.init_process_group("gloo")
    W, H, SPP = args.width or 64, args.height or 64, args.spp or 8
    cfg = pkg.capi.Config.builtin("cornell-box").set_film(W, H, SPP)
    fg = pkg.tiling.FrameGather(world, rank, W, H, torch.device("cpu"))
    rng = pkg.capi.TileRange(rank, world, 0)
    n_tiles = len(pkg.tiling.rank_tiles(rank, world, W, H))
    t0 = time.perf_counter()
    block = np.full((n_tiles, 64, 3), -1.0, dtype=np.float32)
    pkg.capi.host_render(cfg, threads=2, rng=rng, packed=True, out=block)
    mine_ms = 1e3 * (time.perf_counter() - t0)
    packed = np.full((fg.max_tiles, 64, 3), -1.0, dtype=np.float32)   # (poison: padding must never reach the frame)
    packed[:n_tiles] = block
    fg.packed.copy_(torch.from_numpy(packed.reshape(-1)))
    g0 = time.perf_counter()
    fg.gather()
    gather_ms = 1e3 * (time.perf_counter() - g0)
    every = [None] * world
    dist.all_gather_object(every, mine_ms)
    rccl = pkg.tiling.communicator_report(torch.device("cpu"))
    single = pkg.capi.host_render(cfg, threads=2)[0] if rank == 0 else None
    check = pkg.tiling.self_check(fg, single)
    if rank == 0:
        elapsed = max(every) * 1e-3 + gather_ms * 1e-3
        print(json.dumps({"metric": "Msamples/sec (W*H*spp/s)", "value": W * H * SPP / elapsed / 1e6, "unit": "Msamples/s", "n_gpus": world,
                          "steps": 1, "warmup": 0, "ms_per_step": 1e3 * elapsed, "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic; HOST tile writer (CPU test of the N > 1 code path, not a measurement)",
                          "config": {"workload": f"cornell-box {W}x{H} spp={SPP}", "partition": f"8x8 tiles round-robin over {world} rank(s), one gather to rank 0"},
                          "rccl": rccl, "frame_check": check, "per_rank_kernel_ms": every, "gather_ms": gather_ms}))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and not check["equals_single_gpu_frame"]:
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell",
                    choices=["cornell", "dragon", "matpreview-rc", "matpreview-rd", "volumetric"])
    ap.add_argument("--width", type=int, default=0, help="film width (default: the workload's)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--weak", action="store_true",
                    help="N > 1, cornell: grow the film side with sqrt(N) (per-GPU work fixed) instead of "
                         "cutting the workload's own film over the GPUs")
    ap.add_argument("--kernel", choices=["auto", "stream", "lanes", "queued"], default="auto",
                    help="kernel formulation (default: the library's choice by scene class / first-draw calibration)")
    ap.add_argument("--rng", choices=["reference", "pcg", "sobol"], default="reference",
                    help="reference: the reference's per-pixel random stream (the graded mode, frames comparable per "
                         "pixel).  pcg: throughput mode — an independent PCG-hashed stream per (pixel, sample), the "
                         "samples of a pixel spread over lanes; compared with the CPU image by RMSE only.  sobol: the same with "
                         "Owen-scrambled Sobol points (mcpt_renderer_set_rng mode 2)")
    ap.add_argument("--sample-split", type=int, default=0, help="--rng pcg: lanes per pixel (0 = auto)")
    ap.add_argument("--no-throughput-mode", action="store_true",
                    help="skip the second timed loop (same steps, same barriers) in the independent-sample RNG mode that the "
                         "default reference-stream run reports next to `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes")
    ap.add_argument("--no-also", action="store_true",
                    help="N = 1, default workload: skip the `also` block (dragon/scene.xml 1280x720 spp 256, north_star's "
                         "second target, measured in the same process)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record (every counter, note and per-sample count) on stdout instead of the compact "
                         "line; the full record always goes to gpurun_out/bench_detail_<workload>_n<N>.json and stderr")
    ap.add_argument("--force-gather", action="store_true",
                    help="take the multi-GPU code path (RCCL process group, packed tiles, gather, scatter) "
                         "even with one rank: lets a 1-GPU box exercise it")
    ap.add_argument("--cpu-tiles", action="store_true",
                    help="CPU test of the N > 1 code path: gloo + the product's host tile writer, no GPU (tests/test_multi_gpu_host.py)")
    args = ap.parse_args()
    if args.cpu_tiles:
        from _pkg import load_package
        return cpu_tiles_line(load_package(), args)

    import torch
    import torch.distributed as dist
    from _pkg import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_gather = world > 1 or args.force_gather
    if use_gather:
        if "MASTER_ADDR" not in os.environ:   # plain `python bench.py --force-gather`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    name = args.workload
    W, H, SPP = pkg.workloads.WORKLOADS[name][1]
    if args.weak and world > 1:
        if name != "cornell" or args.width or args.height:
            raise SystemExit("--weak is defined for the cornell workload's square film only")
        # the largest square film (side a multiple of the 8-pixel tile) whose per-rank share still fits the
        # 262 144 lanes of one GPU in one round (N = 2: 720, N = 4: 1024, N = 8: 1448)
        side = int(512.0 * (world ** 0.5) / 8.0) * 8
        while -(-((side // 8) ** 2) // world) * 64 > 512 * 512:
            side -= 8
        W = H = side
    film = (args.width or W, args.height or H, args.spp or SPP)
    out = measure(pkg, torch, dist, args, name, film, world, rank, local_rank, device, primary=True)
    default_run = name == "cornell" and film == pkg.workloads.WORKLOADS["cornell"][1] and args.rng == "reference"
    if world == 1 and not use_gather and default_run and not args.no_also:
        # north_star's second target in the same driver-run line
        also = measure(pkg, torch, dist, args, "dragon", pkg.workloads.WORKLOADS["dragon"][1], world, rank, local_rank, device,
                       primary=False)
        out["also"] = {"dragon": also}
    if rank == 0:
        # everything measured -> a side file (and stderr); the compact line -> stdout
        detail_dir = os.path.join(ROOT, "gpurun_out")
        detail_path = None
        try:
            os.makedirs(detail_dir, exist_ok=True)
            detail_path = os.path.join(detail_dir, f"bench_detail_{name}_n{world}.json")
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            detail_path = os.path.relpath(detail_path, ROOT)
        except OSError:
            pass
        if args.full_line:
            print(json.dumps(out))
        else:
            print(json.dumps(out), file=sys.stderr)
            print(json.dumps(compact(out, detail_path), separators=(",", ":")))
    if use_gather:
        dist.destroy_process_group()
    if rank == 0 and out.get("frame_check") and not out["frame_check"]["equals_single_gpu_frame"]:
        print("bench.py: the gathered frame differs from the single-GPU frame", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
