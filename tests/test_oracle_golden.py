"""Oracle vs committed golden data produced by the compiled reference
(tests/golden/make_golden.py).  Runs anywhere (no reference sources needed).
Bar: bit-exact — both are the same IEEE arithmetic in the same order."""
import hashlib
import json
import os

import numpy as np
import pytest

from golden_cases import TRACE_PIXELS, cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
CASE_NAMES = sorted(MANIFEST["frames"])


@pytest.fixture(scope="module")
def scenes(pkg):
    return cases(pkg.scenes)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_scene_builders_unchanged(name, pkg, scenes):
    raw = pkg.mcsd.dumps(scenes[name])
    assert hashlib.sha256(raw).hexdigest() == MANIFEST["frames"][name]["mcsd_sha256"], \
        "scene builder changed: regenerate tests/golden with make_golden.py"


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_frame_bit_exact(name, pkg, oracle, scenes, mcsd_file):
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    frame, _ = oracle.render(mcsd_file(scenes[name]))
    assert frame.shape == golden.shape
    assert np.array_equal(frame, golden), \
        f"max abs diff {np.abs(frame - golden).max():.3e}, {(frame != golden).sum()} values differ"
    assert hashlib.sha256(frame.tobytes()).hexdigest() == MANIFEST["frames"][name]["frame_sha256"]


def test_survey_anchor_sha256():
    # cornell 64x64 spp 8, Woop build of the reference
    assert MANIFEST["frames"]["cornell_64_spp8"]["frame_sha256"] == \
        "14fe16d1ff5bbcecf55ad3859b09d6dc02effb5ed4f212b5bf3021038d3b540c"


def test_oracle_thread_count_independent(pkg, oracle, scenes, mcsd_file):
    path = mcsd_file(scenes["cornell_64_spp8"])
    a, _ = oracle.render(path, threads=1)
    b, _ = oracle.render(path, threads=5)
    assert np.array_equal(a, b)


def test_oracle_pixel_ranges_compose(pkg, oracle, scenes, mcsd_file):
    """Tiles are independent: rendering two pixel ranges equals one render."""
    path = mcsd_file(scenes["depth_limited"])
    full, _ = oracle.render(path)
    n = full.shape[0] * full.shape[1]
    a, _ = oracle.render(path, first_pixel=0, n_pixel=1000)
    b, _ = oracle.render(path, first_pixel=1000, n_pixel=n - 1000)
    assert np.array_equal(a + b, full)


@pytest.mark.parametrize("name,i,j", TRACE_PIXELS)
def test_oracle_per_sample_traces(name, i, j, oracle, scenes, mcsd_file):
    """Per-sample radiance and LCG state: catches draw-order bugs directly."""
    g = np.load(os.path.join(GOLDEN, "pixel_traces.npz"))
    rad, state = oracle.trace_pixel(mcsd_file(scenes[name]), i, j)
    assert np.array_equal(state, g[f"{name}:{i}:{j}:state"])
    assert np.array_equal(rad, g[f"{name}:{i}:{j}:radiance"])


def test_known_answers(oracle):
    kat = MANIFEST["kat"]
    for key, want in kat["tea4"].items():
        a, b = (int(x) for x in key.split(","))
        assert oracle.lib.mcpt_oracle_tea4(a, b) == want
    # SURVEY.md §8c values
    assert oracle.lib.mcpt_oracle_tea4(0, 0) == 1576399551
    assert oracle.lib.mcpt_oracle_tea4(3, 0) == 3153161610
    assert oracle.lib.mcpt_oracle_tea4(786429, 0) == 2470013967
    vals, state = oracle.lcg(3153161610, 8)
    assert [float(np.float32(v)) for v in vals] == kat["lcg_from_3153161610"]["values"]
    assert state == kat["lcg_from_3153161610"]["state"]
    assert np.allclose(vals[:3], [0.814443648, 0.246357679, 0.949649513], rtol=0, atol=1e-9)
    assert [float(np.float32(oracle.lib.mcpt_oracle_vdc2(i))) for i in range(40)] == kat["vdc2"]
    assert [float(np.float32(oracle.lib.mcpt_oracle_vdc3(i))) for i in range(40)] == kat["vdc3"]
    assert kat["vdc2"][1:6] == [0.5, 0.25, 0.75, 0.125, 0.625]


def test_kulla_conty_lut(oracle):
    g = np.load(os.path.join(GOLDEN, "kulla_conty_lut.npz"))
    brdf, albedo = oracle.kulla_conty()
    assert np.array_equal(brdf, g["brdf"]) and np.array_equal(albedo, g["albedo"])
    # SURVEY.md §8 a16 checksums
    assert abs(float(brdf.astype(np.float64).sum()) - 11639.5144) < 1e-3
    assert abs(float(albedo.astype(np.float64).sum()) - 70.3473004) < 1e-5
    assert np.float32(brdf[0]) == np.float32(0.875471771)
    assert np.float32(brdf[-1]) == np.float32(0.309845358)


@pytest.mark.parametrize("n", [1, 2, 3, 12, 100, 1000])
def test_lbvh_builder(n, oracle):
    g = np.load(os.path.join(GOLDEN, "lbvh.npz"))
    out = oracle.bvh_build(g[f"n{n}_in_boxes"], g[f"n{n}_in_areas"])
    assert len(out["leaf"]) == 2 * n - 1
    for k, v in out.items():
        assert np.array_equal(v, g[f"n{n}_{k}"]), k
