"""A/B of library builds on one box: python tools/experiments/ab_libs.py <workload:w:h:spp,...> <lib-or-'prod'> [<lib> ...] [--mode K] [--reps N]
Each (workload, library) is measured in its own process (MCPT_LIB is read at import), libraries alternating, best of 3 draws
per process; prints one JSON line per measurement with the frame hash (equal hashes = same image)."""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json, hashlib
sys.path.insert(0, %r)
from _pkg import load_package
pkg = load_package()
name, w, h, spp, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
r = pkg.capi.Renderer(pkg.workloads.config(name, w, h, spp), device=0)
if mode >= 0:
    r.set_kernel(mode, 0, 0)
r.draw()
best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
frame, st = r.draw()
print(json.dumps({"workload": name, "film": [w, h, spp], "lib": os.environ.get("MCPT_LIB", "prod"), "kernel": r.last_kernel(), "ms": round(best, 3),
                  "msamples": round(w * h * spp / best / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}))
''' % ROOT
args = [a for a in sys.argv[1:] if not a.startswith("--")]
mode = int(sys.argv[sys.argv.index("--mode") + 1]) if "--mode" in sys.argv else -1
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
if "--mode" in sys.argv: args.remove(sys.argv[sys.argv.index("--mode") + 1])
if "--reps" in sys.argv: args.remove(sys.argv[sys.argv.index("--reps") + 1])
jobs = [j.split(":") for j in args[0].split(",")]
libs = args[1:]
for name, w, h, spp in jobs:
    for rep in range(reps):
        for lib in libs:
            env = dict(os.environ)
            env.pop("MCPT_LIB", None)
            if lib != "prod":
                env["MCPT_LIB"] = os.path.join(ROOT, "monte-carlo-path-tracing_amd", "exp", lib, "libmcpt_hip.so") if "/" not in lib else lib
            p = subprocess.run([sys.executable, "-c", CHILD, name, w, h, spp, str(mode)], env=env, capture_output=True, text=True)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else json.dumps({"workload": name, "lib": lib, "error": p.stderr[-400:]})
            print(line, flush=True)
