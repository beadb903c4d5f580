"""The product's wavefront-cooperative code on the host, 64 lanes in lockstep (tests/emu/wave_shim.h, wave_emu.cpp): the kernel BODY
of csrc/hip/render_kernel_impl.h with csrc/pool_walk.h's cooperative ray query, the uniform / merged path steps of csrc/path_core.h,
the work counter, lanes per path and the workgroup compaction — the same source the GPU runs, 256 fibers per workgroup, cross-lane
operations as collectives of exactly the lanes that reach them, LDS an array of the launch's size.

What is pinned here that the GPU tests cannot pin ("test the compaction / sort stages against a serial host model";
VERDICT round 5, "What's weak" 1-2):
  * the PROTOCOL of the item lists: every frame must equal the compiled reference's golden frame with the lanes of a wavefront run in
    ascending, descending and shuffled order between collectives — nothing may depend on the hardware's instruction-level lockstep
    beyond the places that say so (pool_sync, ballots);
  * nothing is read before the query that reads it wrote it: every wavefront's pool area is filled with a pattern (zeros, ones, a
    signalling NaN) before EVERY ray query, and MemorySanitizer (wave_emu_msan) runs the frame with LDS and the pool areas marked
    uninitialised;
  * every LDS access lies inside the launch's allocation (AddressSanitizer + UBSan, wave_emu_asan; on the GPU an access beyond it
    is dropped silently) — and the sanitizer build is shown to notice an allocation that is too small.
It is also round 6's regression test of EXPERIMENTS R6-1: merged queries in the LDS-resident kernel (kPM), the combination that
rendered cornell wrong on the GPU in round 5, are exact here under every order and pattern — the source is right; what was wrong was
the code gfx950's back end generated for it (see test_gpu_parity.py::test_merged_queries_in_lds_equal_the_golden for the device side).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, EMU)

import wave_emu as W  # noqa: E402
from golden_cases import cases  # noqa: E402


@pytest.fixture(scope="module")
def lockstep():
    return W.WaveEmulator()


@pytest.fixture(scope="module")
def scenes(pkg, tmp_path_factory):
    """golden case -> (mcsd path, golden frame)"""
    made = {}
    folder = tmp_path_factory.mktemp("wave_emu")
    all_cases = cases(pkg.scenes)

    def get(name):
        if name not in made:
            path = str(folder / (name + ".mcsd"))
            pkg.mcsd.dump(all_cases[name], path)
            made[name] = (path, np.load(os.path.join(GOLDEN, name + ".npz"))["frame"])
        return made[name]
    return get


def render(lockstep, scenes, name, features, lds, **kw):
    path, golden = scenes(name)
    frame, info = lockstep.render(path, golden.shape[1], golden.shape[0], features, lds, **kw)
    return frame, golden, info


LEAN = {"two queries per vertex": W.P, "merged queries": W.PM}


def test_the_scheduler_models_a_diverged_block_and_refuses_an_unmarked_one(lockstep):
    lib = lockstep.lib
    assert lib.mcpt_wave_emu_selftest(0) == 0, lib.mcpt_wave_emu_last_error().decode()
    assert lib.mcpt_wave_emu_selftest(1) == 0, "a diverged block with a ballot and no MCPT_WAVE_REGION went through"
    assert b"MCPT_WAVE_REGION" in lib.mcpt_wave_emu_last_error()


@pytest.mark.parametrize("form", LEAN)
@pytest.mark.parametrize("order", [0, 1, 2], ids=["ascending", "descending", "shuffled"])
def test_lds_resident_pool_walk_kernel_equals_the_golden_under_every_lane_order(lockstep, scenes, form, order):
    frame, golden, info = render(lockstep, scenes, "cornell_64_spp8", LEAN[form], True, order=order, seed=11 + order)
    assert info["queries"] > 1000 and info["lds_bytes"] in (40640, 53440)
    assert np.array_equal(frame, golden), f"{(frame != golden).any(axis=2).mean():.3f} of the pixels differ"


@pytest.mark.parametrize("form", LEAN)
@pytest.mark.parametrize("word", [0x00000000, 0xFFFFFFFF, 0x7FA00001], ids=["zeros", "ones", "signalling-nan"])
def test_nothing_is_read_from_a_pool_area_before_the_query_wrote_it(lockstep, scenes, form, word):
    frame, golden, _ = render(lockstep, scenes, "cornell_64_spp8", LEAN[form], True, poison=word, order=2, seed=word & 0xFF)
    assert np.array_equal(frame, golden)


@pytest.mark.parametrize("form", LEAN)
@pytest.mark.parametrize("spread,blocks", [(64, 64), (8, 32), (2, 8)])
def test_lanes_without_a_path_help_and_change_nothing(lockstep, scenes, form, spread, blocks):
    """RenderJob::lane_spread: 1 path per 64 / 8 / 2 lanes (a rank's small tile share) — the others are workers of the queries"""
    frame, golden, info = render(lockstep, scenes, "cornell_64_spp8", LEAN[form], True, lane_spread=spread, max_blocks=blocks, poison=0xDEADBEEF)
    assert info["lane_spread"] == spread
    assert np.array_equal(frame, golden)


@pytest.mark.parametrize("order", [0, 2])
def test_workgroup_compaction_moves_paths_and_changes_nothing(lockstep, scenes, order):
    """the compaction's events (render_kernel_impl.h): __syncthreads of all four wavefronts, paths travelling through the pool areas;
    one workgroup for the whole film, so that every event happens"""
    frame, golden, info = render(lockstep, scenes, "cornell_64_spp8", W.P, True, compact=1, max_blocks=1, order=order, poison=0x7FA00001)
    assert info["blocks"] == 1
    assert np.array_equal(frame, golden)


def test_the_work_counter_hands_out_the_whole_film_to_few_lanes(lockstep, scenes):
    frame, golden, info = render(lockstep, scenes, "cornell_96_spp32", W.PM, True, max_blocks=2, threads=2)
    assert info["blocks"] == 2  # 512 lanes for 9216 pixels
    assert np.array_equal(frame, golden)


BIG = [("terrain_directional", W.EMITTERS | W.PB), ("terrain_directional", W.EMITTERS | W.PBU), ("cornell_64_spp8", W.EMITTERS | W.PB),
       ("rough_conductor_envmap", W.SURFACE | W.PB | W.SLIVERS), ("rough_dielectric_envmap", W.SURFACE | W.PB | W.SLIVERS),
       ("thin_dielectric_sun", W.SURFACE | W.PB | W.SLIVERS), ("plastic_spot", W.SURFACE | W.PB | W.SLIVERS),
       ("bumpy_directional", W.SURFACE | W.PB | W.SLIVERS), ("depth_limited", W.SURFACE | W.PB | W.SLIVERS)]


# (the diffuse + emitters kernels outside LDS — dragon/scene.xml's class — are the ones the tail spread is built into)
SPREAD = [("terrain_directional", W.EMITTERS | W.PB, 1, 0, 0), ("terrain_directional", W.EMITTERS | W.PBU, 1, 2, 2), ("cornell_64_spp8", W.EMITTERS | W.PB, 2, 0, 2),
          ("cornell_64_spp8", W.EMITTERS | W.PB, 1, 2, 1), ("cornell_64_spp8", W.EMITTERS | W.PBU, 1, 4, 2)]


@pytest.mark.parametrize("name,features,blocks,spread,order", SPREAD, ids=[f"{n}-{f:#x}-{b}-{s}" for n, f, b, s, _ in SPREAD])
def test_tail_spread_deals_the_last_paths_out_and_changes_nothing(lockstep, scenes, name, features, blocks, spread, order):
    """RenderJob::tail_spread (render_kernel_impl.h, round 6): once the work counter is dry, the paths a workgroup still holds are dealt
    out over its four wavefronts at five thresholds — barriers of all four wavefronts, path state AND the pending shadow ray of merged
    queries travelling through the pool areas.  One / two workgroups for the whole film, so that the counter runs dry with paths in
    flight and every event happens; 1 path per 1 / 2 / 4 lanes.  (That paths do move: a build that flips a bit of the random state of
    every path an event delivers renders 0.03-4 % of these films' pixels differently — EXPERIMENTS R6-13.)"""
    frame, golden, info = render(lockstep, scenes, name, features, False, order=order, seed=7, poison=0xFFFFFFFF, compact=1, max_blocks=blocks, lane_spread=spread)
    assert info["blocks"] == blocks
    assert np.array_equal(frame, golden)


@pytest.mark.parametrize("blocks,spread,order", [(2, 0, 0), (4, 2, 2), (3, 0, 2), (4, 4, 1)])
def test_path_market_between_workgroups_changes_nothing(lockstep, scenes, blocks, spread, order):
    """RenderJob::market (round 6): the wavefronts of a workgroup that is done take tickets and wait; wavefronts of other workgroups that
    still hold two or more paths give half of them away — records in device memory, release / acquire on the slot's generation word — and
    every wavefront leaves when the count of finished items reaches the job's.  Several workgroups on several host threads (real
    atomics between them); how many paths travel depends on their timing (the GPU side pins a frame in which thousands do:
    test_dragon_full_film_equals_the_oracle), that the frame is the golden and that everybody leaves does not."""
    for name, features in (("cornell_64_spp8", W.EMITTERS | W.PB), ("terrain_directional", W.EMITTERS | W.PBU)):
        frame, golden, info = render(lockstep, scenes, name, features, False, order=order, seed=3, poison=0xFFFFFFFF, compact=2, max_blocks=blocks, lane_spread=spread, threads=blocks)  # (one host thread per workgroup: a waiting workgroup needs the others to run)
        assert info["blocks"] == blocks
        assert np.array_equal(frame, golden)


PREPASS = [("cornell_64_spp8", W.EMITTERS | W.PB, 4), ("cornell_64_spp8", W.EMITTERS | W.PBU, 4), ("terrain_directional", W.EMITTERS | W.PB, 6),
           ("rough_conductor_envmap", W.SURFACE | W.PB | W.SLIVERS, 4), ("rough_dielectric_envmap", W.SURFACE | W.PB | W.SLIVERS, 5), ("depth_limited", W.SURFACE | W.PB | W.SLIVERS, 4),
           ("cornell_64_spp8", W.P, 4)]


@pytest.mark.parametrize("name,features,compact", PREPASS, ids=[f"{n}-{f:#x}-{c}" for n, f, c in PREPASS])
def test_the_next_sample_in_the_same_step_changes_nothing(lockstep, scenes, name, features, compact):
    """path_core.h regenerate_in_step (round 6): with the camera-ray pre-pass (here computed on the host: wave_emu.cpp WithPrepass, option
    bit 2) a sample that ends in path_resolve starts its successor in the same step — merged and unmerged queries, environment maps
    (samples that end by leaving the scene), a depth limit, also with the tail spread and the path market switched on, and the LDS kernel
    with a pre-pass.  That it matters: without the pre-pass these renders take 1.3-2 x the path steps (info['rounds'])."""
    lds = features == W.P
    frame, golden, info = render(lockstep, scenes, name, features, lds, order=2, seed=13, poison=0xFFFFFFFF, compact=compact, max_blocks=2, threads=2)
    plain, _, info0 = render(lockstep, scenes, name, features, lds, order=2, seed=13, compact=compact & 3, max_blocks=2, threads=2)
    assert np.array_equal(frame, golden) and np.array_equal(plain, golden)
    assert info["rounds"] < info0["rounds"], (info["rounds"], info0["rounds"])


def test_path_market_does_not_wait_for_workgroups_that_have_not_started(lockstep, scenes):
    """A wavefront that waits in the market holds its slot until the job's last item is finished: it may only wait when every
    workgroup of the launch has started (the count in RenderJob::market[96]) — otherwise workgroups that are not resident yet (another
    kernel on the device) would wait for the slots of wavefronts that wait for them.  Here: three workgroups on ONE host thread, one
    after the other — a workgroup that waited would wait for ever."""
    frame, golden, info = render(lockstep, scenes, "cornell_64_spp8", W.EMITTERS | W.PB, False, order=2, seed=9, compact=2, max_blocks=3, lane_spread=0, threads=1)
    assert info["blocks"] == 3
    assert np.array_equal(frame, golden)


@pytest.mark.parametrize("name,features", BIG, ids=[f"{n}-{f:#x}" for n, f in BIG])
def test_pool_walk_kernels_outside_lds_equal_the_golden(lockstep, scenes, name, features):
    """32-bit items, the quantised 4-wide hierarchy, the leaf-box test at the primitive, merged (kPB) and unmerged (kPBU) queries,
    directional / environment / spot lights pending with the next segment's query"""
    frame, golden, _ = render(lockstep, scenes, name, features, False, order=2, seed=5, poison=0xFFFFFFFF)
    assert np.array_equal(frame, golden)


G = W.GROUP128
SORTED = [("volumetric_96x54_spp16", W.VOLUME_LEAN | W.ORDERED | G, True), ("volumetric_iso_64x36_spp8", W.VOLUME_LEAN | W.P | G, True),
          ("volumetric_96x54_spp16", W.ALL | W.ORDERED | G, True), ("rough_conductor_envmap", W.SURFACE | W.PBU | W.CONDUCTOR_ONLY | G, False),
          ("rough_dielectric_envmap", W.SURFACE | W.PBU | W.DIELECTRIC_ONLY | G, False), ("plastic_spot", W.SURFACE | W.PBU | G, False),
          ("bumpy_directional", W.SURFACE | W.PBU | G, False)]


@pytest.mark.parametrize("name,features,lds", SORTED, ids=[f"{n}-{f:#x}" for n, f, _ in SORTED])
@pytest.mark.parametrize("order", [0, 2], ids=["ascending", "shuffled"])
def test_class_sorted_kernel_body_equals_the_golden(lockstep, scenes, name, features, lds, order):
    """csrc/hip/sorted_body.h in lockstep, workgroups of 128 lanes: the count sort over the workgroup (ballots, ranks, the per-class
    counts in LDS), the exchange of the path states through LDS in passes with its barriers — and, outside LDS (round 6), through the
    wavefronts' pool areas between two pool walks: volumetric-caustic's production kernel, its pool-walk form, and the surface-material
    meshes' (matpreview's class)."""
    path, golden = scenes(name)
    frame, info = lockstep.render_sorted(path, golden.shape[1], golden.shape[0], features, lds, order=order, seed=3, poison=0xFFFFFFFF, max_blocks=2)
    assert info["blocks"] == 2
    assert np.array_equal(frame, golden)


def test_an_instantiation_that_does_not_cover_the_scene_is_refused(lockstep, scenes):
    with pytest.raises(RuntimeError, match="does not cover"):
        render(lockstep, scenes, "rough_conductor_envmap", W.P, False)


# ---- the same source under the sanitizers (programs: the runtimes want to be the process's first library) ----
def sanitizer_program(name):
    subprocess.run(["make", "-s", "-C", EMU, name], check=True)
    return os.path.join(EMU, name)


def run_program(program, scene, features, frame_path, poison_word=0, shortfall=0, env=None):
    args = [program, scene, str(features), "1", "2", "1", str(poison_word), "0", "0", frame_path] + ([str(shortfall)] if shortfall else [])
    return subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=900)


@pytest.mark.parametrize("form", LEAN)
def test_address_and_undefined_behaviour_sanitizers_find_nothing(scenes, tmp_path, form):
    program = sanitizer_program("wave_emu_asan")
    path, golden = scenes("cornell_64_spp8")
    out = str(tmp_path / "frame.f32")
    done = run_program(program, path, LEAN[form], out, poison_word=0xDEADBEEF)
    assert done.returncode == 0 and "ERROR: AddressSanitizer" not in done.stderr and "runtime error" not in done.stderr, done.stderr[-2000:]
    assert np.array_equal(np.fromfile(out, dtype=np.float32).reshape(golden.shape), golden)


def test_the_address_sanitizer_build_notices_an_lds_allocation_that_is_too_small(scenes, tmp_path):
    """what the GPU would do silently (a dropped write, a read of 0) is an error here: the model of the launch's LDS size is live"""
    program = sanitizer_program("wave_emu_asan")
    path, _ = scenes("cornell_64_spp8")
    done = run_program(program, path, W.PM, str(tmp_path / "frame.f32"), shortfall=1024)
    assert done.returncode != 0 and "heap-buffer-overflow" in done.stderr


@pytest.mark.parametrize("form", LEAN)
def test_memory_sanitizer_with_uninitialised_lds_and_pool_areas_finds_nothing(scenes, tmp_path, form):
    program = sanitizer_program("wave_emu_msan")
    path, golden = scenes("cornell_64_spp8")
    out = str(tmp_path / "frame.f32")
    done = run_program(program, path, LEAN[form], out, poison_word=0x7FA00001, env={"MSAN_OPTIONS": "halt_on_error=1"})
    assert done.returncode == 0 and "MemorySanitizer" not in done.stderr, done.stderr[-2000:]
    assert np.array_equal(np.fromfile(out, dtype=np.float32).reshape(golden.shape), golden)
