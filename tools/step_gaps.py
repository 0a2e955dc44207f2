#!/usr/bin/env python3
"""Per-launch time AND inter-launch gaps of a step loop, from a rocprofv3 --kernel-trace CSV:
    python tools/step_gaps.py <dir with *kernel_trace.csv> <label> [last-kernel-substring]
A step ends with every launch of the last kernel (default: accelerate_step_kernel).  Steps of the steady state
(the last 60 %) are averaged: per position in the step, the kernel's duration and the idle time since the previous launch ended."""
import collections, csv, glob, sys

d, label = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "accelerate_step_kernel"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nbody::", "")[:58]
cnt = collections.Counter(name(r) for r in rows)
idx = [i + 1 for i, r in enumerate(rows) if want in name(r)]
idx = idx[int(len(idx) * 0.4):]
steps = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]
shape = collections.Counter(tuple(name(r) for r in s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if tuple(name(r) for r in s) == shape]
k = len(shape)
dur, gap = [0.0] * k, [0.0] * k
wall = 0.0
for si, s in enumerate(steps):
    for j, r in enumerate(s):
        dur[j] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev_end = int(s[j - 1]["End_Timestamp"]) if j else None
        if j:
            gap[j] += (int(r["Start_Timestamp"]) - prev_end) / 1e3
    wall += (int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3
# the gap in front of the first kernel: from the last kernel of the previous step
g0 = [(int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"])) / 1e3 for a, b in zip(steps[:-1], steps[1:])
      if int(b[0]["Start_Timestamp"]) > int(a[-1]["End_Timestamp"])]
ns = len(steps)
print(f"{label}: {ns} steady-state steps of {k} launches ({f.split('/')[-1]})")
print(f"  {'launch':60s} {'kernel us':>10s} {'idle before us':>15s}")
for j in range(k):
    gb = (sum(g0) / len(g0) if g0 else 0.0) if j == 0 else gap[j] / ns
    print(f"  {shape[j]:60s} {dur[j] / ns:10.2f} {gb:15.2f}")
tot_k, tot_g = sum(dur) / ns, sum(gap) / ns + (sum(g0) / len(g0) if g0 else 0.0)
print(f"  sum of kernels {tot_k:.2f} us + idle {tot_g:.2f} us = {tot_k + tot_g:.2f} us per step (device clock)")
