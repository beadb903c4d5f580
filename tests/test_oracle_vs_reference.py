"""Oracle restatement vs the REAL reference compiled from its own sources
(oracle/_ref).  Skipped where the compiled reference is not available.  This is
what pins the oracle: identical frames, unit outputs and builder tables."""
import numpy as np
import pytest


def _cmp_frames(oracle, reference, path, scene):
    fo, _ = oracle.render(path)
    fr, _ = reference.render(path, scene.camera.width, scene.camera.height)
    assert np.array_equal(fo, fr), \
        f"max abs diff {np.abs(fo - fr).max():.3e}, {(fo != fr).sum()} values differ"


@pytest.mark.parametrize("material", [
    "diffuse", "rough_diffuse_fast", "rough_diffuse_full", "conductor",
    "rough_conductor", "rough_conductor_aniso", "dielectric", "rough_dielectric",
    "thin_dielectric", "plastic", "rough_plastic", "bumpy_diffuse", "masked_diffuse"])
def test_materials(material, pkg, oracle, reference, mcsd_file):
    scene = pkg.scenes.material_preview(material, "mixed", "mesh", 40, 40, 4)
    _cmp_frames(oracle, reference, mcsd_file(scene), scene)


@pytest.mark.parametrize("lighting", ["area", "point", "spot", "directional", "sun",
                                      "envmap", "constant"])
def test_emitters(lighting, pkg, oracle, reference, mcsd_file):
    scene = pkg.scenes.material_preview("rough_plastic", lighting, "sphere", 40, 40, 4)
    _cmp_frames(oracle, reference, mcsd_file(scene), scene)


@pytest.mark.parametrize("shape", ["mesh", "flat_mesh", "sphere", "cube", "disk", "cylinder"])
def test_shapes(shape, pkg, oracle, reference, mcsd_file):
    scene = pkg.scenes.material_preview("rough_dielectric", "mixed", shape, 40, 40, 4)
    _cmp_frames(oracle, reference, mcsd_file(scene), scene)


def test_volpath(pkg, oracle, reference, mcsd_file):
    scene = pkg.scenes.volumetric_caustic(80, 45, 8)
    _cmp_frames(oracle, reference, mcsd_file(scene), scene)
    scene = pkg.scenes.material_preview("dielectric", "mixed", "sphere", 40, 40, 4,
                                        integrator="volpath", medium=True)
    _cmp_frames(oracle, reference, mcsd_file(scene), scene)


def test_cornell_and_terrain(pkg, oracle, reference, mcsd_file):
    for scene in (pkg.scenes.cornell_box(72, 48, 8), pkg.scenes.terrain_scene(40, 60, 40, 2)):
        _cmp_frames(oracle, reference, mcsd_file(scene), scene)


def test_bsdf_units(pkg, oracle, reference, mcsd_file):
    """Sample / Evaluate of every BSDF kind on random (also non-orthonormal,
    also geometrically impossible) records: outputs and LCG state identical."""
    rng = np.random.default_rng(5)

    def unit(v):
        return v / np.linalg.norm(v)

    for material in ("diffuse", "rough_diffuse_full", "rough_conductor_aniso", "conductor",
                     "dielectric", "rough_dielectric", "thin_dielectric", "plastic", "rough_plastic"):
        path = mcsd_file(pkg.scenes.material_preview(material, "constant", "sphere", 8, 8, 1))
        with oracle.open(path) as so, reference.open(path) as sr:
            for _ in range(300):
                n = unit(rng.normal(size=3))
                t = unit(np.cross(n, rng.normal(size=3)) + 0.2 * rng.normal(size=3))
                b = unit(np.cross(n, t))
                rec = np.concatenate([unit(rng.normal(size=3)), unit(rng.normal(size=3)), n, t, b,
                                      rng.random(2), [rng.integers(2)]]).astype(np.float32)
                seed = int(rng.integers(1 << 32))
                for mode in (0, 1):
                    a, sa = so.bsdf(1, mode, rec, seed)
                    c, sc = sr.bsdf(1, mode, rec, seed)
                    assert sa == sc and np.array_equal(a, c, equal_nan=True), (material, mode, rec, a, c)


def test_intersection_units(pkg, oracle, reference, mcsd_file):
    rng = np.random.default_rng(9)
    for shape in ("mesh", "sphere", "disk", "cylinder", "cube"):
        path = mcsd_file(pkg.scenes.material_preview("bumpy_diffuse", "area", shape, 8, 8, 1))
        with oracle.open(path) as so, reference.open(path) as sr:
            for _ in range(600):
                org = rng.normal(size=3) * 1.5 + [0, 0.8, 0]
                d = np.array([0, 0.6, 0]) + rng.normal(size=3) * 0.5 - org
                d /= np.linalg.norm(d)
                a, sa = so.intersect(org, d)
                c, sc = sr.intersect(org, d)
                assert sa == sc and np.array_equal(a, c, equal_nan=True), (shape, org, d, a, c)


def test_lbvh_random(oracle, reference):
    rng = np.random.default_rng(11)
    for n in (1, 2, 5, 37, 500):
        lo = (rng.random((n, 3)) * 4).astype(np.float32)
        hi = lo + (rng.random((n, 3)) * 0.5).astype(np.float32)
        if n == 37:
            hi[:, 2] = lo[:, 2] = 1.0  # planar: zero extent -> NaN Morton input
        areas = rng.random(n).astype(np.float32)
        a = oracle.bvh_build(np.concatenate([lo, hi], 1), areas)
        b = reference.bvh_build(np.concatenate([lo, hi], 1), areas)
        for k in a:
            assert np.array_equal(a[k], b[k]), (n, k)


def test_kulla_conty(oracle, reference):
    a, b = oracle.kulla_conty(), reference.kulla_conty()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_reference_binding_round_trip(pkg, reference, mcsd_file, tmp_path):
    """The binding a reference maintainer would add (integration/mcpt_backend.hpp,
    csrt::ToMcsd): every field of csrt::RendererConfig must reach the C ABI's
    configuration — MCSD -> RendererConfig -> MCSD returns the same bytes."""
    from golden_cases import cases
    for name, scene in cases(pkg.scenes).items():
        path = mcsd_file(scene, name + ".mcsd")
        out = tmp_path / (name + ".round_trip.mcsd")
        reference.binding_round_trip(path, out)
        assert out.read_bytes() == open(path, "rb").read(), name
