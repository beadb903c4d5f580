#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 6 / README from the bench records of a session directory (bench_<workload>.json)."""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles"
prefix = sys.argv[2] if len(sys.argv) > 2 else "r04_bench_"
for w in ("cornell", "dragon", "matpreview-rc", "matpreview-rd", "volumetric"):
    p = os.path.join(d, f"{prefix}{w}.json")
    if not os.path.exists(p):
        continue
    m = json.load(open(p))
    r, v, h = m["roofline"], m["roofline"].get("valu", {}), m["roofline"]["hbm"]
    cpu, port = m.get("cpu_baseline", {}), m.get("cpu_baseline_port", {})
    print(f"| {w} | **{m['value']:.0f}** | {m['ms_per_step']:.1f} | {m['first_draw_ms']:.1f} | {m.get('grays_per_s', 0):.2f} | {r['kernel'][:80]} | "
          f"{r['bound']}, frac {r['frac']:.3f} | issue {v.get('issue_frac', 0):.2f} × lanes {v.get('lane_util', 0):.2f} ({v.get('frac_at_16_lanes_per_clk', 0):.2f} at 16 lanes/clk) | "
          f"{v.get('valu_insts_per_sample', 0):.0f} | {h.get('measured_gbs', 0):.0f} GB/s, {r['traffic'] / 1e9 if r.get('traffic') else 0:.3g} GB vs {h['bytes_per_sample'] * m['config'].get('samples', 0) / 1e9 if False else h['algorithmic_gbs'] * r['kernel_ms'] / 1e3:.3g} GB algorithmic | "
          f"{v.get('wait_any_per_wave_cycle', 0):.2f} | {cpu.get('value', 0):.2f} / {port.get('value', 0):.2f} ({cpu.get('cores', 0)} cores) | "
          f"{'exact' if m.get('parity', {}).get('frac_exact') == 1.0 else m.get('parity')} | {m.get('throughput_mode', {}).get('value', 0):.0f} |")
