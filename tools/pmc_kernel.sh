#!/bin/bash
# Per-dispatch PMC values of one kernel (substring match) in a short CLI run; prints the dispatch with the largest first counter.
# Usage: [PREC=double|float] [WL=galaxy|uniform] [TAG=pmck] pmc_kernel.sh <kernel-substring> <algorithm> <n> "<counters pass 1>" ["<counters pass 2>" ...]
KS=$1; ALGO=$2; N=$3; shift 3
PREC=${PREC:-double}; WL=${WL:-galaxy}; TAG=${TAG:-pmck}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for PASS in "$@"; do
  OUT=$R/gpurun_out/${TAG}_$i; rm -rf $OUT; mkdir -p $OUT
  timeout -k 5 60 rocprofv3 --pmc $PASS --output-format csv -d $OUT -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n $N -s 2 --algorithm $ALGO --workload $WL --precision $PREC --csv-detailed > $OUT/out.txt 2> $OUT/err.txt || { echo "pass $i failed: $PASS"; tail -3 $OUT/err.txt; i=$((i+1)); continue; }
  python3 - $OUT "$KS" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in d.values() for k in v})
print("dispatch", *names)
for k in sorted(d):
    print(k, *[f"{d[k].get(n, 0):.4g}" for n in names])
PY
  i=$((i+1))
done
