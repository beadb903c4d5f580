"""GPU box: BASELINE configs at their full film sizes, HIP frame vs the oracle's frame
(about 11 minutes of box time, most of it the CPU oracle).  The matpreview / volumetric
configurations come from scratch/real/*.mcsd (tools/convert_reference_scenes.py)."""
import sys, os, time, json, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from _pkg import load_package
pkg = load_package()
import checkers
from mcpt_amd import capi
orc = checkers.Oracle()
os.makedirs('/tmp/standin', exist_ok=True)
pkg.mcsd.dump(pkg.scenes.blob_field_scene(), '/tmp/standin/blob.mcsd')
CASES = [('config 2: cornell-box 512x512 spp 256', 'builtin:cornell-box', (512, 512, 256)),
         ('config 3 stand-in: blob field 0.8M triangles 1280x720 spp 256', '/tmp/standin/blob.mcsd', (1280, 720, 256)),
         ('config 4: matpreview rough_conductor 1024x1024 spp 512', 'scratch/real/matpreview_rough_conductor.mcsd', (1024, 1024, 512)),
         ('config 4: matpreview rough_dielectric 1024x1024 spp 512', 'scratch/real/matpreview_rough_dielectric.mcsd', (1024, 1024, 512)),
         ('config 5: volumetric-caustic 1280x720 spp 1024', 'scratch/real/volumetric_caustic.mcsd', (1280, 720, 1024))]
out = []
for name, src, film in CASES:
    cfg = (capi.Config.builtin(src[8:]) if src.startswith('builtin:') else capi.Config.load_mcsd(src)).set_film(*film)
    p = tempfile.mktemp(suffix='.mcsd'); cfg.save_mcsd(p)
    r = capi.Renderer(cfg)
    r.draw(); frame, st = r.draw()
    r.close()
    t = time.time(); want, info = orc.render(p); t_cpu = time.time() - t
    d = frame.astype(np.float64) - want
    l2 = np.sqrt((d ** 2).sum(-1))
    n = film[0] * film[1] * film[2]
    rec = {'config': name, 'film': film, 'rmse': float(np.sqrt((d ** 2).mean())), 'mean_l2': float(l2.mean()),
           'median_l2': float(np.median(l2)), 'max_l2': float(l2.max()), 'frac_l2_over_1e-3': float((l2 > 1e-3).mean()),
           'frac_bit_exact': float((l2 == 0).mean()), 'hip_kernel_ms': st['kernel_milliseconds'],
           'hip_msamples_per_s': n / st['kernel_milliseconds'] / 1e3, 'oracle_seconds': info['seconds'],
           'oracle_msamples_per_s': n / info['seconds'] / 1e6, 'oracle_threads': os.cpu_count()}
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/full_size_parity.json', 'w'), indent=1)
