import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
groups = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_INSTS_VMEM_RD"],
          ["FETCH_SIZE"], ["WRITE_SIZE"]]
res = {}
for gi, g in enumerate(groups):
    d = f"/tmp/pmcwf{gi}"
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *g, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "tools", "render_scene.py"), "workload:matpreview-rd", "--film", "1024", "1024", "16", "--draws", "1", "--kernel-mode", "3", "--prepass", "1"]
    try:
        subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=200)
    except subprocess.TimeoutExpired:
        print("timeout", g); continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = "shade" if "wavefront_shade" in row["Kernel_Name"] else "trace" if "wavefront_trace" in row["Kernel_Name"] else None
            if k:
                res.setdefault(k, {})
                res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, c in res.items():
    print(k, json.dumps({n: v for n, v in c.items()}))
    if c.get("SQ_WAVE_CYCLES"):
        print("   wait/wave-cycle %.2f  valu-active/wave-cycle %.3f  lane util %.2f  valu insts %.3g  vmem rd %.3g  fetch GB %.1f write GB %.1f" % (
            c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
            c["SQ_INSTS_VALU"], c["SQ_INSTS_VMEM_RD"], 2 * c.get("FETCH_SIZE", 0) / 1e6, c.get("WRITE_SIZE", 0) / 1e6))
