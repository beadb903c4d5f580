"""Register budgets of the stream kernel, alternating in one process: python tools/experiments/waves_ab.py workload:w:h:spp [reps]"""
import sys, os, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
name, w, h, spp = sys.argv[1].split(":")
film = (int(w), int(h), int(spp))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
ms = {2: [], 3: [], 4: []}
for waves in ms:
    r.set_stream_waves(waves); r.draw()
for rep in range(reps):
    for waves in ms:
        r.set_stream_waves(waves)
        ms[waves].append(r.draw()[1]["kernel_milliseconds"])
for waves, v in ms.items():
    print(json.dumps({"workload": name, "film": film, "waves": waves, "min": round(min(v), 2), "median": round(statistics.median(v), 2), "max": round(max(v), 2),
                      "msamples_median": round(film[0] * film[1] * film[2] / statistics.median(v) / 1e3, 1)}), flush=True)
r.close()
