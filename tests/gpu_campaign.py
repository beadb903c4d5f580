"""Differential campaign on the GPU (not collected by pytest; run by hand: python tests/gpu_campaign.py <seed> ...).
Random scenes (test_random_combinations' generator with more
seeds, plus mesh resolutions on both sides of the 2048-primitive switch to the vote walk, supplied
tangents, object transforms), both walks, plain and counting kernels, against the oracle.  Round-1 result: profiles/r01_gpu_campaign.json.
Round 2: every frame must EQUAL the oracle's (bit-exact device libm), and so must the frames of the other scheduling
choices: stream kernel, work counter, camera-ray pre-pass (profiles/r02_gpu_campaign.json).
Round 3 adds: class sort off, image-order tiles, the stream kernel at 2 / 3 / 4 wavefronts per SIMD (profiles/r03_gpu_campaign.json).
Round 4: the default ray query is the wavefront-cooperative pool walk and every renderer checks itself when it is created; added: the
per-lane walk (pool walk off) and the pool walk forced in the lanes kernel with / without pre-pass; how many renderers fell back to
the reference-order walk at creation; the Sobol mode against its host twin (tests/emu/libmcpt_emu_ld.so) on every fourth scene
(profiles/r04_gpu_campaign.json)."""
import sys, time, importlib, tempfile, os, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('monte-carlo-path-tracing_amd')
from oracle import checkers
from test_random_combinations import combos
orc = checkers.Oracle()
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
import emu
twin = emu.Emulator(low_discrepancy=True)
fell_back, sobol_checked, kernels = 0, 0, {}
S, M = pkg.scenes, pkg.mcsd
tmp = tempfile.mkdtemp()
seeds = [int(s) for s in sys.argv[1:]] or list(range(100, 112))
bad, n, worst = [], 0, 0.0
t0 = time.time()
for seed in seeds:
    rng = np.random.default_rng(seed * 7 + 1)
    for name, scene in combos(pkg, 40, seed=seed):
        for inst in scene.instances:
            if inst.type == M.INST_MESHES and inst.normals is not None and len(inst.normals) and rng.random() < 0.5:
                res = int(rng.choice([6, 24, 40]))
                g = S.uv_sphere_mesh(res, 2 * res, 0.6, (0, 0.6, 0))
                inst.positions, inst.normals, inst.texcoords, inst.indices = g["positions"], g["normals"], g["texcoords"], g["indices"]
                if rng.random() < 0.5:
                    nrm = inst.normals.astype(np.float64)
                    t = np.cross(nrm, rng.normal(size=3)); t /= np.linalg.norm(t, axis=1, keepdims=True) + 1e-30
                    inst.tangents = t.astype(np.float32); inst.bitangents = np.cross(nrm, t).astype(np.float32)
                if rng.random() < 0.5:
                    inst.to_world = (S.translate_scale(t=tuple(rng.normal(size=3) * 0.1), s=tuple(1 + rng.random(3) * 0.3)) @ S.rot_x(float(rng.random() * 40))).astype(np.float32)
        if rng.random() < 0.3:
            # a fan of sliver triangles (aspect ~1000) through the scene: the sliver-aware kernel variants
            pos, idx = [], []
            for k in range(int(rng.integers(5, 40))):
                a = rng.uniform(-1.5, 1.5, 3); a[1] = abs(a[1]) * (0 if k % 4 == 0 else 1)
                d = rng.normal(size=3); d[1] *= 0.2 * (k % 4 != 0); d /= np.linalg.norm(d)
                side = np.cross(d, rng.normal(size=3)); side /= np.linalg.norm(side)
                b, c = a + 3.0 * d, a + 1.5 * d + 0.003 * side
                idx.append([len(pos), len(pos) + 1, len(pos) + 2]); pos += [a, b, c]
            scene.instances.append(M.Instance(type=M.INST_MESHES, id_bsdf=0, positions=np.asarray(pos, np.float32),
                                              indices=np.asarray(idx, np.uint32)))
        path = os.path.join(tmp, 's.mcsd'); M.dump(scene, path)
        want, _ = orc.render(path)
        r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
        fell_back += r.walk()   # (1: the creation's self-check found a differing pixel and chose the reference-order walk)
        a, _ = r.draw()
        kernel = r.last_kernel().split(',')[0].split(' + ')[0]
        kernels[kernel] = kernels.get(kernel, 0) + 1
        c, _ = r.draw(counted=True)
        others = []
        # round 4: the ray query — one walk per lane, and the pool walk in the lanes kernel with and without the pre-pass
        f, _ = r.set_pool_walk(0).draw()
        others.append(bool(np.array_equal(f, a)))
        for prepass in (0, 1):
            f, _ = r.set_pool_walk(1).set_kernel(0).set_prepass(prepass).draw()
            others.append(bool(np.array_equal(f, a)))
        r.set_pool_walk(-1).set_kernel(-1).set_prepass(-1)
        if n % 4 == 0:
            w, h = scene.camera.width, scene.camera.height
            try:
                want_ld, _ = twin.set_rng(True, seed).render(path, w, h)
            finally:
                twin.set_rng(False)
            f, _ = r.set_rng(2, seed=seed, sample_split=1).draw()
            others.append(bool(np.array_equal(f, want_ld)))
            r.set_rng(0)
            sobol_checked += 1
        for kernel, work, prepass in ((0, 0, 0), (1, 1, 1), (0, 1, 1), (1, 0, 0), (4, 1, 1), (4, 0, 0)):
            f, _ = r.set_kernel(kernel).set_work_distribution(work).set_prepass(prepass).draw()
            others.append(bool(np.array_equal(f, a)))
        r.set_kernel(-1).set_work_distribution(-1).set_prepass(-1)
        # round 3: class sort off, image-order tiles (no cost probe), the stream kernel's three register budgets
        f, _ = r.set_class_sort(0).set_tile_order(0).draw()
        others.append(bool(np.array_equal(f, a)))
        r.set_class_sort(-1).set_tile_order(-1)
        for waves in (2, 3, 4):
            f, _ = r.set_kernel(4).set_work_distribution(1).set_prepass(1).set_stream_waves(waves).draw()
            others.append(bool(np.array_equal(f, a)))
        r.set_kernel(-1).set_work_distribution(-1).set_prepass(-1).set_stream_waves(-1)
        r.set_walk(True); b, _ = r.draw(); r.close()
        n += 1
        d = np.abs(a.astype(np.float64) - want)
        ok = np.array_equal(a, b) and np.array_equal(a, c) and np.isfinite(a).all() and np.array_equal(a, want) and all(others)
        worst = max(worst, float(d.mean()))
        if not ok:
            bad.append({'seed': seed, 'name': name, 'walks_equal': bool(np.array_equal(a, b)), 'counted_equal': bool(np.array_equal(a, c)),
                        'finite': bool(np.isfinite(a).all()), 'equals_oracle': bool(np.array_equal(a, want)), 'scheduling_choices_equal': others, 'mean': float(d.mean()), 'median': float(np.median(d)), 'max': float(d.max())})
            print('BAD', bad[-1], flush=True)
    print('seed', seed, 'done; scenes', n, 'bad', len(bad), 'elapsed', round(time.time() - t0), flush=True)
    json.dump({'scenes': n, 'bad': bad, 'worst_mean': worst, 'fell_back_to_the_reference_walk_at_creation': fell_back,
               'sobol_mode_checked_against_its_host_twin': sobol_checked, 'kernel_of_the_default_draw': kernels, 'seeds_done': seeds[:seeds.index(seed) + 1]},
              open(os.path.join(ROOT, 'gpurun_out', 'campaign.json'), 'w'), indent=1)   # (kept current: a run cut short still reports)
json.dump({'scenes': n, 'bad': bad, 'worst_mean': worst, 'fell_back_to_the_reference_walk_at_creation': fell_back,
           'sobol_mode_checked_against_its_host_twin': sobol_checked, 'kernel_of_the_default_draw': kernels}, open(os.path.join(ROOT, 'gpurun_out', 'campaign.json'), 'w'), indent=1)
print('scenes', n, 'bad', len(bad), 'worst mean', worst)
