"""Seeded random combinations of material x lighting x shape x integrator x medium x
depth limit (the golden cases cover 17 hand-picked ones): on the CPU the product's
kernel body must equal the oracle bit for bit with either walk; on the GPU the HIP
renderer must agree with the oracle and its two walks with each other."""
import os
import sys

import numpy as np
import pytest

N_CPU, N_GPU = 24, 48


def combos(pkg, n, seed):
    S = pkg.scenes
    rng = np.random.default_rng(seed)
    shapes = ["mesh", "flat_mesh", "sphere", "cube", "disk", "cylinder"]
    out = []
    for k in range(n):
        material = str(rng.choice(S.MATERIALS))
        lighting = str(rng.choice(S.LIGHTINGS))
        shape = str(rng.choice(shapes))
        volpath = bool(rng.integers(2))
        medium = volpath and bool(rng.integers(2))
        depth = int(rng.choice([pkg.mcsd.INVALID, 2, 6]))
        w, h = int(rng.integers(17, 41)), int(rng.integers(9, 33))
        name = f"{k}:{material}/{lighting}/{shape}/{'volpath' if volpath else 'path'}{'+medium' if medium else ''}/d{depth}/{w}x{h}"
        out.append((name, S.material_preview(material, lighting, shape, w, h, 3, "volpath" if volpath else "path",
                                             depth, medium)))
    return out


@pytest.fixture(scope="module")
def emulator():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu
    return emu.Emulator()


def test_random_combinations_on_cpu(pkg, oracle, emulator, mcsd_file):
    for name, scene in combos(pkg, N_CPU, seed=2024):
        path = mcsd_file(scene)
        w, h = scene.camera.width, scene.camera.height
        want, _ = oracle.render(path)
        for variant in (-1, emulator.REFERENCE):
            got, _ = emulator.render(path, w, h, variant=variant)
            assert np.array_equal(got, want), (name, variant, float(np.abs(got - want).max()))


@pytest.mark.gpu
def test_random_combinations_on_gpu(pkg, oracle, mcsd_file):
    worst = 0.0
    for name, scene in combos(pkg, N_GPU, seed=777):
        want, _ = oracle.render(mcsd_file(scene))
        r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
        frame, _ = r.draw()
        r.set_walk(True)
        other, _ = r.draw()
        r.close()
        assert np.array_equal(frame, other), name
        d = np.abs(frame.astype(np.float64) - want)
        assert np.isfinite(frame).all(), name
        # bit-exact (csrc/glibc_libm.h); a medium's double exp / log come from the device library and may
        # round differently once in ~2^28 calls: there a flipped decision on a tiny film at spp 3 is bounded
        if scene.media:
            assert d.mean() <= 4e-3 and np.median(d) == 0.0, (name, d.mean(), d.max())
        else:
            assert np.array_equal(frame, want), (name, d.mean(), d.max())
        worst = max(worst, d.mean())
    print("largest mean |difference| over the combinations:", worst)
