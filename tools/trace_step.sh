#!/bin/bash
# rocprofv3 kernel trace of a few CLI steps; prints the kernel timeline of the last step.  Usage: trace_step.sh <algorithm> <n> [precision]
set -e
ALGO=${1:-octree}; N=${2:-1000000}; PREC=${3:-double}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/trace_${ALGO}_$N
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n $N -s 12 --algorithm $ALGO --workload galaxy --precision $PREC --csv-total > $OUT/out.txt 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].split("::")[-1][:34]
import collections
cnt = collections.Counter(name(r) for r in rows)
first = next(name(r) for r in rows if cnt[name(r)] >= 3 and not name(r).startswith("__amd"))  # first kernel of a step
idx = [i for i, r in enumerate(rows) if name(r) == first]
a, b = idx[-2], idx[-1]
for r in rows[a:b]:
    print(f"{name(r):36s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.2f} us  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size')}")
print("step wall", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, "us")
PY
