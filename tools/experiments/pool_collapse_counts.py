#!/usr/bin/env python3
"""Node items and primitive tests per sample of the pool walk, and the frame time, for the hierarchy this process builds
(MCPT_POOL_ORDER=built|sorted; the collapse variants of R4-9 were MCPT_POOL_COLLAPSE=greedy|dp|dpa|ga|dps in a build that is not kept): one JSON line.  EXPERIMENTS R4-9."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
out = {"collapse": os.environ.get("MCPT_POOL_ORDER", os.environ.get("MCPT_POOL_COLLAPSE", "default"))}
for name, film in (("cornell", (512, 512, 64)), ("dragon", (1280, 720, 64)), ("matpreview-rc", (1024, 1024, 64)), ("matpreview-rd", (1024, 1024, 64))):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    _, st = r.draw(counted=True)
    r.draw()
    ms = sorted(r.draw()[1]["kernel_milliseconds"] for _ in range(5))[2]
    out[name] = {"node_items_per_sample": st["node_tests"] / 4 / st["samples"], "prim_tests_per_sample": st["prim_tests"] / st["samples"],
                 "wave_node_steps_per_sample": st.get("wave_node_steps", 0) / st["samples"], "median_kernel_ms": ms, "kernel": r.last_kernel()[:50]}
    r.close()
print(json.dumps(out))
