#!/bin/bash
# Config 2's step loop under rocprofv3 --kernel-trace: the CLI (recorded step, replayed), the same loop as plain asynchronous calls,
# and as a graph from Python; per-launch time and inter-launch gaps (tools/step_gaps.py).  Usage: config2_gaps.sh <tag>
set -e
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c2_cli /tmp/c2_eager /tmp/c2_graph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2_cli -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 65536 -s 100 --algorithm all-pairs --precision double --csv-total > $OUT/config2_cli.txt 2>/dev/null
C2_MODE=eager rocprofv3 --kernel-trace --output-format csv -d /tmp/c2_eager -- python3 $R/tools/c2_eager.py > $OUT/config2_eager.txt 2>/dev/null
C2_MODE=graph rocprofv3 --kernel-trace --output-format csv -d /tmp/c2_graph -- python3 $R/tools/c2_eager.py > $OUT/config2_graph.txt 2>/dev/null
cp $(ls /tmp/c2_cli/*/*kernel_stats.csv | head -1) $OUT/config2_kernel_stats.csv
{
  echo "# config 2 (all-pairs 3D double, n = 65536, uniform, 100 steps): per-launch time and inter-launch gaps, rocprofv3 --kernel-trace"
  cat $OUT/config2_cli.txt
  python3 $R/tools/step_gaps.py /tmp/c2_cli "CLI (recorded step replayed, host/drivers.hpp)"
  cat $OUT/config2_graph.txt
  python3 $R/tools/step_gaps.py /tmp/c2_graph "python, recorded step replayed"
  cat $OUT/config2_eager.txt
  python3 $R/tools/step_gaps.py /tmp/c2_eager "python, plain asynchronous calls (what tools/measure_configs.py timed in round 4)"
  echo "# the same three loops WITHOUT the profiler (host clock):"
  $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 65536 -s 100 --algorithm all-pairs --precision double --csv-total
  C2_MODE=graph python3 $R/tools/c2_eager.py
  C2_MODE=eager python3 $R/tools/c2_eager.py
} > $OUT/config2_gaps.txt 2>&1
cat $OUT/config2_gaps.txt
