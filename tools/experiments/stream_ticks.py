import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from _pkg import load_package
pkg = load_package()
W = pkg.workloads
rows = []
for name, film in (("cornell", (512, 512, 32)), ("matpreview-rd", (512, 512, 16)), ("dragon", (640, 360, 16)), ("volumetric", (640, 360, 32))):
    cfg = W.config(name, *film)
    r = pkg.capi.Renderer(cfg, device=0)
    for mode, slots in ((0, 0), (1, 0), (2, 512), (2, 1024)):
        try:
            r.set_kernel(mode, slots, 0)
            r.draw()
            _, plain = r.draw()
            _, st = r.draw(counted=True)
        except Exception as e:
            print(name, mode, slots, "error", e); continue
        s = st["samples"]
        tot = max(st["ticks_shade"] + st["ticks_trace"] + st["ticks_wait"], 1)
        row = dict(workload=name, mode=mode, slots=slots, kernel=r.last_kernel(), plain_ms=plain["kernel_milliseconds"],
                   counted_ms=st["kernel_milliseconds"], msamples=s / plain["kernel_milliseconds"] / 1e3,
                   node_util=st["node_tests"] / 2 / 64 / max(st["wave_node_steps"], 1),
                   prim_util=st["prim_tests"] / 64 / max(st["wave_prim_steps"], 1),
                   wave_node_steps_per_sample=st["wave_node_steps"] / s, wave_prim_steps_per_sample=st["wave_prim_steps"] / s,
                   node_tests_per_sample=st["node_tests"] / s, prim_tests_per_sample=st["prim_tests"] / s,
                   rays_per_sample=(st["closest_rays"] + st["shadow_rays"]) / s,
                   shade_frac=st["ticks_shade"] / tot, trace_frac=st["ticks_trace"] / tot, wait_frac=st["ticks_wait"] / tot,
                   rounds=st["rounds"])
        print(json.dumps(row), flush=True)
        rows.append(row)
    r.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/stream_ticks.json", "w"), indent=1)
