"""The N>1 path on CPU: two gloo processes each produce the pixels of their
own tile set (with the oracle as the pixel source), pack them the way the HIP
kernel packs tiles, gather once to rank 0 and scatter into the frame.  The
assembled frame must equal the single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mcsd_path, width, height, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import torch
    import torch.distributed as dist
    import checkers
    from _pkg import load_package
    pkg = load_package()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg = pkg.tiling.FrameGather(world, rank, width, height, torch.device("cpu"))
    # "render" this rank's tiles: full oracle frame, then keep only own pixels
    full, _ = checkers.Oracle().render(mcsd_path, threads=2)
    tiles = pkg.tiling.rank_tiles(rank, world, width, height)
    x, y, ok = pkg.tiling.tile_pixel_grid(tiles, width, height)
    packed = np.full((fg.max_tiles, 64, 3), -1.0, dtype=np.float32)   # poison padding
    packed[:len(tiles)][ok] = full[y[ok], x[ok]]
    fg.packed.copy_(torch.from_numpy(packed.reshape(-1)))
    frame = fg.gather()
    if rank == 0:
        np.save(out_path, frame.numpy().reshape(height, width, 3))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_of_tiles_rebuilds_the_frame(world, pkg, oracle, mcsd_file, tmp_path):
    import torch.multiprocessing as mp
    width, height = 44, 29  # partial edge tiles in both directions
    scene = pkg.scenes.cornell_box(width, height, 2)
    path = mcsd_file(scene)
    want, _ = oracle.render(path)
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, path, width, height, out), nprocs=world, join=True)
    got = np.load(out)
    assert np.array_equal(got, want)


def _product_worker(rank, world, port, mcsd_path, width, height, out_path):
    """Like _worker, but the rank's packed tile block is written by the PRODUCT's tile writer — the kernel body's
    host build (libmcpt_host.so, mcpt_host_render_tiles), the same enumeration of tiles and the same packed layout as
    the HIP kernel's — not derived from a full frame."""
    sys.path[:0] = [ROOT]
    import torch
    import torch.distributed as dist
    from _pkg import load_package
    pkg = load_package()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg = pkg.tiling.FrameGather(world, rank, width, height, torch.device("cpu"))
    packed = np.full((fg.max_tiles, 64, 3), -1.0, dtype=np.float32)   # poison: padding must never reach the frame
    cfg = pkg.capi.Config.load_mcsd(mcsd_path)
    rng = pkg.capi.TileRange(rank, world, 0)
    n_tiles = len(pkg.tiling.rank_tiles(rank, world, width, height))
    block = np.ascontiguousarray(packed[:n_tiles])
    pkg.capi.host_render(cfg, threads=2, rng=rng, packed=True, out=block)
    packed[:n_tiles] = block
    fg.packed.copy_(torch.from_numpy(packed.reshape(-1)))
    frame = fg.gather()
    if rank == 0:
        np.save(out_path, frame.numpy().reshape(height, width, 3))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_of_the_products_own_packed_tiles(world, pkg, mcsd_file, tmp_path):
    """World-2 / world-3 gloo run fed by the product's packed-tile writer (host build of the kernel body): the
    gathered frame equals the compiled reference's golden frame bit for bit."""
    import torch.multiprocessing as mp
    from golden_cases import cases
    scene = cases(pkg.scenes)["cornell_64_spp8"]
    golden = np.load(os.path.join(ROOT, "tests", "golden", "cornell_64_spp8.npz"))["frame"]
    path = mcsd_file(scene)
    out = str(tmp_path / "frame.npy")
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_product_worker, args=(world, port, path, 64, 64, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), golden)


def test_host_tile_writer_matches_full_frame(pkg):
    """mcpt_host_render_tiles: frame layout and packed layout of three ranks' tile ranges (edge tiles in both
    directions) compose to the full-frame render."""
    cfg = pkg.capi.Config.builtin("cornell-box").set_film(44, 29, 2)
    full, _ = pkg.capi.host_render(cfg)
    composed, from_packed = np.zeros_like(full), np.zeros_like(full)
    for rank in range(3):
        rng = pkg.capi.TileRange(rank, 3, 0)
        pkg.capi.host_render(cfg, rng=rng, out=composed)
        block, _ = pkg.capi.host_render(cfg, rng=rng, packed=True)
        pkg.capi.unpack_tiles(block, rng, 44, 29, from_packed)
    assert np.array_equal(composed, full) and np.array_equal(from_packed, full)


def test_partition_covers_every_pixel_once(pkg):
    for world in (1, 2, 3, 8):
        for (w, h) in ((512, 512), (1280, 720), (21, 13)):
            src, dst = pkg.tiling.scatter_map(world, w, h)
            assert len(dst) == w * h and len(np.unique(dst)) == w * h
            assert len(np.unique(src)) == len(src)
            counts = [len(pkg.tiling.rank_tiles(r, world, w, h)) for r in range(world)]
            assert max(counts) - min(counts) <= 1 and max(counts) == pkg.tiling.max_tiles_per_rank(world, w, h)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_line_of_an_n_rank_run_verifies_itself(world, pkg, tmp_path):
    """`bench.py --gpus N` prints, for N > 1, the communicator's own count of its ranks (`rccl.ranks`, an all-reduce of ones — not
    WORLD_SIZE), per-rank times, the gather time and `frame_check`: the gathered frame's sha256 against the frame one renderer
    draws alone.  Here the same code path (tiling.FrameGather / communicator_report / self_check) runs under torch.distributed.run
    with gloo and the product's host tile writer (`bench.py --cpu-tiles`); the frame is also the compiled reference's golden."""
    import hashlib
    import json
    import subprocess
    port = 33500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--cpu-tiles"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["rccl"]["ranks"] == world and line["rccl"]["backend"] == "gloo"
    assert len(line["per_rank_kernel_ms"]) == world and line["gather_ms"] >= 0.0
    assert line["frame_check"]["equals_single_gpu_frame"] is True
    golden = np.load(os.path.join(ROOT, "tests", "golden", "cornell_64_spp8.npz"))["frame"]
    assert line["frame_check"]["gathered_frame_sha256"] == hashlib.sha256(np.ascontiguousarray(golden).tobytes()).hexdigest()
