"""Edge cases of the render path: empty and one-instance scenes, no lights, 1x1 and
ragged films (not a multiple of the 8x8 tile), spp 1, depth limits 0 / 1, roulette
from the first bounce, a degenerate triangle.  CPU: oracle == compiled reference ==
the product's kernel body, bit for bit.  GPU: the HIP renderer against the oracle."""
import numpy as np
import pytest


def edge_scenes(pkg):
    S, M = pkg.scenes, pkg.mcsd
    out = {}
    s = S.cornell_box(16, 16, 2)
    s.instances = []
    s.emitters = [M.Emitter(type=M.EMIT_CONSTANT, radiance=(0.5, 0.5, 0.5))]
    out["empty_scene_constant_emitter"] = s
    s = S.cornell_box(16, 16, 2)
    s.instances = []
    out["empty_scene_dark"] = s
    s = S.cornell_box(16, 16, 2)
    s.instances = s.instances[-1:]
    out["light_only"] = s
    s = S.cornell_box(16, 16, 2)
    s.instances = s.instances[:-1]
    out["no_lights"] = s
    out["film_1x1"] = S.cornell_box(1, 1, 4)
    out["film_3x5"] = S.cornell_box(3, 5, 2)
    out["film_65x9"] = S.cornell_box(65, 9, 2)
    out["spp_1"] = S.cornell_box(16, 16, 1)
    for name, field, value in (("depth_max_0", "depth_max", 0), ("depth_max_1", "depth_max", 1),
                               ("roulette_from_start", "depth_rr", 0)):
        s = S.cornell_box(16, 16, 2)
        setattr(s.integrator, field, value)
        out[name] = s
    s = S.cornell_box(16, 16, 2)
    s.instances.append(M.Instance(type=M.INST_MESHES, id_bsdf=0,
                                  positions=np.array([[0, 1, 0], [0, 1, 0], [0.2, 1.2, 0]], np.float32),
                                  indices=np.array([[0, 1, 2]], np.uint32)))
    out["degenerate_triangle"] = s
    # a mesh whose texture coordinates are all equal and that has no normals: the tangent rule divides by a
    # zero uv determinant, the shading frame is NaN, the scattered ray is NaN — which "passes" every
    # comparison of the reference's box and triangle tests and is only dropped when an instance's hit is
    # merged (tlas.cpp:27-33).  (Found with the reference's classroom scene, where one such triangle made
    # 88 % of the pixels differ before closest queries dropped NaN distances.)
    s = S.cornell_box(24, 24, 4)
    s.instances.append(M.Instance(type=M.INST_MESHES, id_bsdf=2,
                                  positions=np.array([[-0.6, 0.9, -0.5], [0.6, 0.9, -0.5], [0, 0.9, 0.7]], np.float32),
                                  texcoords=np.array([[0, 1], [0, 1], [0, 1]], np.float32),
                                  indices=np.array([[0, 1, 2]], np.uint32)))
    out["nan_shading_frame"] = s
    # slivers: triangles ~1000 times longer than wide, lying in and just above the floor and across the box.
    # Their computed hit distance is unreliable (cancellation in the edge functions) and so is whether a ray
    # that starts on one "hits" it again just beyond t_min; the reference's answers then depend on its
    # visiting order and on its flat leaf boxes.  The ordered walk marks them (kWalkSliver), reaches them
    # through grown boxes and replays the reference's decision (commit.cpp, traversal.h::test_slot).
    s = S.cornell_box(40, 40, 4)
    rng = np.random.default_rng(12)
    pos, idx = [], []
    for k in range(60):
        a = rng.uniform(-0.9, 0.9, 3)
        d = rng.normal(size=3)
        if k % 3 == 0:
            a[1], d[1] = 0.0, 0.0                      # in the floor plane (y = 0 in this scene)
        elif k % 3 == 1:
            a[1], d[1] = 1e-4, 0.0                     # a hair above it
        d /= np.linalg.norm(d)
        side = np.cross(d, [0.3, 1.0, 0.2])
        side /= np.linalg.norm(side)
        if k % 3 != 2:
            side[1] = 0.0
        b, c = a + 1.9 * d, a + 0.5 * 1.9 * d + 0.002 * side
        idx.append([len(pos), len(pos) + 1, len(pos) + 2])
        pos += [a, b, c]
    s.instances.append(M.Instance(type=M.INST_MESHES, id_bsdf=1, positions=np.asarray(pos, np.float32),
                                  indices=np.asarray(idx, np.uint32)))
    out["slivers"] = s
    # per-vertex tangents and / or bitangents handed over with the mesh (what the reference gets from
    # assimp's CalcTangentSpace; scene.cpp:81-100, transformed by to_world at :271-280): an anisotropic
    # conductor makes the frame's orientation visible
    for name, use_t, use_b in (("supplied_tangents", True, True), ("supplied_tangents_only", True, False),
                               ("supplied_bitangents_only", False, True)):
        s = S.material_preview("rough_conductor_aniso", "mixed", "mesh", 24, 24, 4)
        inst = next(i for i in s.instances if i.type == M.INST_MESHES)
        n = inst.normals.astype(np.float64)
        t = np.cross(n, [0.3, 1.0, 0.2])
        t /= np.linalg.norm(t, axis=1, keepdims=True) + 1e-30
        if use_t:
            inst.tangents = (0.7 * t).astype(np.float32)            # not unit length on purpose
        if use_b:
            inst.bitangents = np.cross(n, t).astype(np.float32)
        inst.to_world = (S.translate_scale(t=(0.1, 0.0, -0.2), s=(1.0, 1.2, 0.9)) @ S.rot_x(20)).astype(np.float32)
        out[name] = s
    return out


NAMES = ["empty_scene_constant_emitter", "empty_scene_dark", "light_only", "no_lights", "film_1x1", "film_3x5",
         "film_65x9", "spp_1", "depth_max_0", "depth_max_1", "roulette_from_start", "degenerate_triangle", "nan_shading_frame", "slivers",
         "supplied_tangents", "supplied_tangents_only", "supplied_bitangents_only"]


@pytest.fixture(scope="module")
def emulator():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu
    return emu.Emulator()


@pytest.mark.parametrize("name", NAMES)
def test_edge_case_on_cpu(name, pkg, oracle, emulator, mcsd_file, request):
    scene = edge_scenes(pkg)[name]
    path = mcsd_file(scene)
    w, h = scene.camera.width, scene.camera.height
    want, _ = oracle.render(path)
    assert want.shape == (h, w, 3) and np.isfinite(want).all()
    for variant in (-1, emulator.REFERENCE):
        got, _ = emulator.render(path, w, h, variant=variant)
        np.testing.assert_array_equal(got, want)
    import checkers
    # (the compiled reference rebuilds its Kulla-Conty table on every render: ~5 s, so a subset)
    if checkers.reference_available() and name in ("empty_scene_constant_emitter", "film_3x5", "depth_max_0",
                                                   "degenerate_triangle", "nan_shading_frame", "slivers", "supplied_tangents",
                                                   "supplied_tangents_only", "supplied_bitangents_only"):
        ref, _ = checkers.Reference().render(path, w, h)
        np.testing.assert_array_equal(ref, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_edge_case_on_gpu(name, pkg, oracle, mcsd_file):
    scene = edge_scenes(pkg)[name]
    want, _ = oracle.render(mcsd_file(scene))
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    frame, st = r.draw()
    r.set_walk(True)
    reference_order, _ = r.draw()
    r.close()
    assert np.array_equal(frame, reference_order)
    assert st["samples"] == scene.camera.width * scene.camera.height * scene.camera.spp
    d = np.abs(frame.astype(np.float64) - want)
    # bit-exact (csrc/glibc_libm.h), NaN pixels cannot occur (per-sample clamp); with a medium see
    # test_random_combinations_on_gpu
    assert np.isfinite(frame).all()
    if scene.media:
        assert d.mean() <= 2e-3 and np.median(d) == 0.0, (name, d.mean(), d.max())
    else:
        assert np.array_equal(frame, want), (name, d.mean(), d.max())
