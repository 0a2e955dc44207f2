#!/bin/bash
# The reference's detailed benchmark (ci/benchmark_detailed:9-15,45,61-65: octree + bvh, galaxy, D=3, double, N = 100 000,
# 1000 steps on GPUs, --csv-detailed so that every phase gets its own column) through this repository's CLI, as a log in
# the same shape as tools/benchmark.sh.  Usage: bash tools/benchmark_detailed.sh [steps] [bodies] > bench_detailed.log
set -e
STEPS=${1:-1000}
BODIES=${2:-100000}
HERE=$(cd "$(dirname "$0")/.." && pwd)
BIN=$HERE/stdpar-nbody_amd/bin/nbody_hip_d3
bash "$HERE/tools/bench_log_header.sh"
CC="hipcc-$(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version: [0-9.]*' | cut -d' ' -f3)-gfx950"
for algo in octree bvh; do
  echo "compiler:$CC"
  $BIN -n $BODIES -s $STEPS --precision double --algorithm $algo --workload galaxy --csv-detailed
done
