import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
for name, film in (("cornell", (512, 512, 256)),):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    for work in (1,):
        r.set_work_distribution(work)
        r.draw()
        best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
        frame, _ = r.draw()
        print(json.dumps({"workload": name, "film": film, "scatter": os.environ.get("MCPT_SCATTER", "0"), "work": work, "ms": round(best, 3),
                          "msamples": round(film[0] * film[1] * film[2] / best / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12], "kernel": r.last_kernel()[:50]}), flush=True)
    r.close()
