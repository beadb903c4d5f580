#!/usr/bin/env python3
"""Does the library's built-in choice of kernel generalise beyond the five BASELINE scenes?  For the reference's OTHER shipped
scenes that load (tools/convert_other_scenes.py -> scratch/other/*.mcsd.gz: classroom, dining-room, box, matpreview rough
plastic / thin dielectric) and the BASELINE meshes, at 640 x 360 spp 64: the default (rule) against every explicit configuration
— lanes kernel with the pool walk, lanes kernel one walk per lane, stream kernel in workgroup rounds, in wavefront rounds — each
timed on full draws (median of 3 after a warm-up draw), plus what mcpt_renderer_calibrate picks.  One JSON document:
profiles/r04_rule_vs_calibrated.json.  Frames compared bit for bit on the way."""
import glob
import gzip
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = [("rule", {}), ("lanes+pool", {"kernel": 0, "pool": 1}), ("lanes", {"kernel": 0, "pool": 0}), ("stream", {"kernel": 1}),
           ("stream-wavefront-rounds", {"kernel": 4})]


def main():
    from _pkg import load_package
    pkg = load_package()
    film = (640, 360, 64)
    scenes = []
    for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "other", "*.mcsd.gz"))):
        scenes.append((os.path.basename(path)[:-8], lambda p=path: pkg.capi.Config.from_mcsd_bytes(gzip.open(p, "rb").read()).set_film(*film)))
    for name in ("dragon", "matpreview-rc", "matpreview-rd"):
        scenes.append((name, lambda n=name: pkg.workloads.config(n, *film)))
    rows = []
    for name, make in scenes:
        row = {"scene": name, "film": film, "configs": {}}
        digests = set()
        for label, settings in CONFIGS:
            r = pkg.capi.Renderer(make(), device=0)
            try:
                if "kernel" in settings:
                    r.set_kernel(settings["kernel"])
                if "pool" in settings:
                    r.set_pool_walk(settings["pool"])
                r.draw()
                ms = []
                for _ in range(3):
                    frame, st = r.draw()
                    ms.append(st["kernel_milliseconds"])
                digests.add(hashlib.sha256(frame.tobytes()).hexdigest()[:16])
                row["configs"][label] = {"ms": round(float(np.median(ms)), 3), "kernel": r.last_kernel()[:120]}
                if label == "rule":
                    row["info"] = {k: r.info()[k] for k in ("primitives", "features", "walk_depth")}
                    r.calibrate()
                    r.draw()
                    row["calibrated_choice"] = r.last_kernel()[:200]
            finally:
                r.close()
        best = min(row["configs"], key=lambda k: row["configs"][k]["ms"])
        row["best"] = best
        row["rule_over_best"] = round(row["configs"]["rule"]["ms"] / row["configs"][best]["ms"], 3)
        row["same_frame"] = len(digests) == 1
        print(json.dumps(row), flush=True)
        rows.append(row)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "rule_vs_calibrated.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
