import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
for name, film in (("cornell", (512, 512, 128)), ("volumetric", (1280, 720, 64))):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    for mode, refill, prepass in ((0, 0, 0), (1, 0, 0), (4, 0, 0), (4, 8, 0), (4, 0, 1), (4, 40, 0)):
        r.set_kernel(mode, 0, refill).set_work_distribution(1).set_prepass(prepass)
        r.draw()
        best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
        frame, _ = r.draw()
        _, st = r.draw(counted=True)
        tot = max(st["ticks_shade"] + st["ticks_trace"] + st["ticks_wait"], 1)
        print(json.dumps({"workload": name, "mode": mode, "refill": refill, "prepass": prepass, "ms": round(best, 3),
                          "msamples": round(film[0] * film[1] * film[2] / best / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12],
                          "shade": round(st["ticks_shade"] / tot, 2), "trace": round(st["ticks_trace"] / tot, 2), "wait": round(st["ticks_wait"] / tot, 2),
                          "node_util": round(st["node_tests"] / 2 / 64 / max(st["wave_node_steps"], 1), 3),
                          "wave_node_steps_per_sample": round(st["wave_node_steps"] / st["samples"], 3), "kernel": r.last_kernel()[:60]}), flush=True)
    r.close()
