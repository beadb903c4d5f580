import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
for name, film in (("dragon", (1280, 720, 64)), ("matpreview-rd", (1024, 1024, 32))):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    r.set_kernel(1).set_work_distribution(1).set_prepass(1)
    r.draw(); _, plain = r.draw()
    _, st = r.draw(counted=True)
    tot = max(st["ticks_shade"] + st["ticks_trace"] + st["ticks_wait"], 1)
    info = r.info()
    print(json.dumps({"workload": name, "film": film, "plain_ms": plain["kernel_milliseconds"], "counted_ms": st["kernel_milliseconds"], "kernel": r.last_kernel(),
                      "rounds_total": st["rounds"], "shade_frac": st["ticks_shade"] / tot, "trace_frac": st["ticks_trace"] / tot, "wait_frac": st["ticks_wait"] / tot,
                      "wave_node_steps": st["wave_node_steps"], "wave_prim_steps": st["wave_prim_steps"], "closest": st["closest_rays"], "shadow": st["shadow_rays"],
                      "samples": st["samples"], "node_tests": st["node_tests"]}))
    r.close()
