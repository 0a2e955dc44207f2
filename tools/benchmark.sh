#!/bin/bash
# The reference's benchmark matrix (ci/benchmark:7-16,64,86: galaxy, D=3, double, 200 steps under --csv-total => 10 warm-up
# + 190 timed; every algorithm at N = 100 000, the trees also at N = 1 000 000) through this repository's CLI, as a LOG in
# the shape ci/benchmark writes: identification lines, then per run a "compiler:<name>" line and the CLI's own CSV
# header + row.  Feed the log to tools/scrape_bench_log.py for the one-table CSV (what ci/data.py does for the reference).
# Usage: bash tools/benchmark.sh [steps] > bench.log
set -e
STEPS=${1:-200}
HERE=$(cd "$(dirname "$0")/.." && pwd)
BIN=$HERE/stdpar-nbody_amd/bin/nbody_hip_d3
bash "$HERE/tools/bench_log_header.sh"
CC="hipcc-$(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version: [0-9.]*' | cut -d' ' -f3)-gfx950"
run() { echo "compiler:$CC"; $BIN -n $2 -s $STEPS --precision double --algorithm $1 --workload galaxy --csv-total; }
for algo in all-pairs all-pairs-collapsed octree bvh; do run $algo 100000; done
for algo in octree bvh; do run $algo 1000000; done
