"""Independent-sample mode at 4 / 3 / 2 wavefronts per SIMD of the stream kernel (mesh workloads)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
for name, film in (("dragon", (1280, 720, 256)), ("matpreview-rc", (1024, 1024, 128)), ("matpreview-rd", (1024, 1024, 128))):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    r.set_rng(1, seed=1, sample_split=0)
    for waves in (4, 3, 2, 4, 3, 2):
        r.set_stream_waves(waves)
        r.draw()
        best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
        print(json.dumps({"workload": name, "film": film, "rng": "pcg", "waves": waves, "ms": round(best, 3), "msamples": round(film[0] * film[1] * film[2] / best / 1e3, 1),
                          "kernel": r.last_kernel()[:80]}), flush=True)
    r.close()
