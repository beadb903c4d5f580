"""Round 6: volumetric-caustic's class-sorted kernel with its per-lane walk (the library's choice) against the pool walk inside it
(mcpt_renderer_set_pool_walk(r, 2)), re-measured on the final kernel.  Same process, alternating, frames compared by hash."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
r = pkg.capi.Renderer(pkg.workloads.config("volumetric"), device=0)
ms, sha, kernel = {-1: [], 2: []}, {}, {}
for rnd in range(2):
    for mode in (-1, 2):
        r.set_pool_walk(mode)
        r.draw()
        for _ in range(4):
            frame, st = r.draw()
            ms[mode].append(st["kernel_milliseconds"])
        sha[mode], kernel[mode] = hashlib.sha256(frame.tobytes()).hexdigest()[:12], r.last_kernel()[:110]
rec = {"frames_identical": sha[-1] == sha[2]}
for mode, label in ((-1, "per_lane_walk"), (2, "pool_walk")):
    v = sorted(ms[mode]); rec[label] = {"median_ms": round(v[len(v) // 2], 2), "min_ms": round(v[0], 2), "max_ms": round(v[-1], 2), "kernel": kernel[mode]}
print(json.dumps(rec))
