#!/bin/bash
# Per-dispatch PMC values of one kernel (substring match) for a python command; prints the LAST dispatch of that kernel.
# Usage: pmc_py.sh <kernel-substring> "<counters>" <outdir> -- python3 script args...
KS=$1; PASS=$2; OUT=$3; shift 4
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 120 rocprofv3 --pmc $PASS --output-format csv -d $OUT -- "$@" > $OUT/out.txt 2> $OUT/err.txt || { echo "failed: $PASS"; tail -3 $OUT/err.txt; exit 0; }
python3 - $OUT "$KS" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in d.values() for k in v})
print("dispatch", *names)
k = sorted(d)[-1]
print(k, *[f"{d[k].get(n, 0):.5g}" for n in names])
PY
