#!/usr/bin/env python3
"""RMSE of the two throughput modes (PCG-hashed streams / Owen-scrambled Sobol points: mcpt_renderer_set_rng 1 / 2) against a
high-spp frame of the reference stream, at 16 / 64 / 256 spp, on the BASELINE workloads at reduced films.  One JSON line per
workload; EXPERIMENTS R4-7."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
for name, film, truth_spp in (("cornell", (256, 256), 8192), ("dragon", (640, 360), 4096), ("matpreview-rc", (256, 256), 8192),
                              ("matpreview-rd", (256, 256), 8192), ("volumetric", (320, 180), 8192)):
    r = pkg.capi.Renderer(pkg.workloads.config(name, film[0], film[1], truth_spp), device=0)
    truth = r.draw()[0].astype(np.float64)
    r.close()
    row = {"workload": name, "film": film, "truth_spp": truth_spp, "rmse": {}}
    for spp in (16, 64, 256):
        r = pkg.capi.Renderer(pkg.workloads.config(name, film[0], film[1], spp), device=0)
        errs = {}
        for mode, label in ((1, "pcg"), (2, "sobol")):
            e = [float(np.sqrt(((r.set_rng(mode, seed=s).draw()[0].astype(np.float64) - truth) ** 2).mean())) for s in (1, 2, 3)]
            errs[label] = float(np.mean(e))
        r.close()
        errs["ratio"] = errs["sobol"] / errs["pcg"]
        row["rmse"][str(spp)] = errs
    slope = lambda k: float(np.polyfit(np.log([16, 64, 256]), np.log([row["rmse"][str(n)][k] for n in (16, 64, 256)]), 1)[0])
    row["slope"] = {"pcg": slope("pcg"), "sobol": slope("sobol")}
    print(json.dumps(row), flush=True)
