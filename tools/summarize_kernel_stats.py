#!/usr/bin/env python3
"""Per-step table from a rocprofv3 `*_kernel_stats.csv` of a CLI run of S steps:
    python tools/summarize_kernel_stats.py <kernel_stats.csv> <steps> [label]"""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
label = sys.argv[3] if len(sys.argv) > 3 else path
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
launches = sum(int(r["Calls"]) for r in rows) / steps
print(f"{label}: total kernel time per step {total:.3f} ms, {launches:.0f} launches per step")
for r in rows:
    per_step = float(r["TotalDurationNs"]) / steps / 1e3
    if per_step < 0.05:
        continue
    name = r["Name"].split("(")[0].replace("void ", "").replace("nbody::", "")
    print(f"  {name[:60]:60s} calls/step {int(r['Calls']) / steps:5.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  per step {per_step:8.1f} us")
