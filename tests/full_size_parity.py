"""GPU box: every BASELINE configuration at its FULL film and spp (monte-carlo-path-tracing_amd/workloads.py: the
reference's own scene files through the product's front end), HIP frame vs the oracle's frame, pixel by pixel
(about 12 minutes of box time, most of it the CPU oracle).  Not collected by pytest: python tests/full_size_parity.py"""
import sys, os, time, json, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from _pkg import load_package
pkg = load_package()
import checkers
from mcpt_amd import capi
orc = checkers.Oracle()
CASES = [('config 2: cornell-box 512x512 spp 256', 'cornell'),
         ('config 3: dragon/scene.xml 1280x720 spp 256 (12 shipped meshes + stand-ins, 845 808 triangles)', 'dragon'),
         ('config 4: matpreview rough_conductor 1024x1024 spp 512', 'matpreview-rc'),
         ('config 4: matpreview rough_dielectric 1024x1024 spp 512', 'matpreview-rd'),
         ('config 5: volumetric-caustic 1280x720 spp 1024', 'volumetric')]
out = []
for name, workload in CASES:
    film = pkg.workloads.WORKLOADS[workload][1]
    cfg = pkg.workloads.config(workload)
    p = tempfile.mktemp(suffix='.mcsd'); cfg.save_mcsd(p)
    r = capi.Renderer(cfg)
    r.draw(); frame, st = r.draw()
    r_name = r.last_kernel()
    r.close()
    t = time.time(); want, info = orc.render(p); t_cpu = time.time() - t
    d = frame.astype(np.float64) - want
    l2 = np.sqrt((d ** 2).sum(-1))
    n = film[0] * film[1] * film[2]
    rec = {'config': name, 'film': film, 'rmse': float(np.sqrt((d ** 2).mean())), 'mean_l2': float(l2.mean()),
           'median_l2': float(np.median(l2)), 'max_l2': float(l2.max()), 'frac_l2_over_1e-3': float((l2 > 1e-3).mean()),
           'frac_bit_exact': float((l2 == 0).mean()), 'hip_kernel_ms': st['kernel_milliseconds'],
           'hip_kernel': r_name, 'hip_msamples_per_s': n / st['kernel_milliseconds'] / 1e3, 'oracle_seconds': info['seconds'],
           'oracle_msamples_per_s': n / info['seconds'] / 1e6, 'oracle_threads': os.cpu_count()}
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/full_size_parity.json', 'w'), indent=1)
