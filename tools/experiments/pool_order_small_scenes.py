#!/usr/bin/env python3
"""Frame time of three LDS-resident diffuse scenes for the child order this process builds (MCPT_POOL_ORDER=built|sorted; R4-9 ran it as MCPT_POOL_COLLAPSE=greedy|ga): EXPERIMENTS R4-9."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
S = pkg.scenes
out = {"collapse": os.environ.get("MCPT_POOL_ORDER", "default")}
for name, scene in (("terrain 16x16 cells, directional light", S.terrain_scene(16, 512, 512, 64)),
                    ("terrain 20x20 cells", S.terrain_scene(20, 512, 512, 64)),
                    ("cornell 256x256", S.cornell_box(256, 256, 256))):
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    f, _ = r.draw()
    ms = sorted(r.draw()[1]["kernel_milliseconds"] for _ in range(7))
    out[name] = {"median_ms": round(ms[3], 3), "min_ms": round(ms[0], 3), "kernel": r.last_kernel()[:45], "sha": hashlib.sha256(f.tobytes()).hexdigest()[:12]}
    r.close()
print(json.dumps(out))
