#!/bin/bash
# Benchmark matrix in the shape of the reference's (ci/benchmark:7-16,64,86: galaxy workload, D=3, double,
# 200 steps under --csv-total => 10 warm-up + 190 timed; N=100000 for every algorithm, N=1000000 for the tree),
# run through this repository's CLI.  One CSV row per run with the host/GPU columns ci/data.py scrapes
# (gpu, driver, cpu, cores, compiler, hostname).  Usage: bash tools/benchmark_matrix.sh [steps] > matrix.csv
set -e
STEPS=${1:-200}
HERE=$(cd "$(dirname "$0")/.." && pwd)
BIN=$HERE/stdpar-nbody_amd/bin/nbody_hip_d3
GPU=$(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 "Marketing Name:.*MI" | sed 's/.*Marketing Name: *//' | tr -d ',' || true)
ARCH=$(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 -o "gfx[0-9a-z]*" || true)
DRV=$(cat /sys/module/amdgpu/version 2>/dev/null || echo unknown)
CPU=$(lscpu | grep -m1 "Model name" | sed 's/.*: *//' | tr -d ',')
CORES=$(nproc)
CC="hipcc $(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version: [0-9.]*' | cut -d' ' -f3)"
echo "algorithm,dim,precision,nsteps,nbodies,total [s],ms/step,gpu,arch,driver,cpu,cores,compiler,hostname"
run() {
  row=$($BIN -n $2 -s $STEPS --precision double --algorithm $1 --workload galaxy --csv-total $3 | tail -1)
  total=$(echo $row | cut -d, -f6); nst=$(echo $row | cut -d, -f4)
  ms=$(python3 -c "print(f'{1e3*$total/max(1,$nst):.3f}')")
  echo "$row,$ms,${GPU:-unknown},${ARCH:-unknown},$DRV,$CPU,$CORES,$CC,$(hostname)"
}
for algo in all-pairs all-pairs-collapsed octree bvh; do run $algo 100000; done
run octree 1000000
run bvh 1000000
run octree 1000000 "--theta 0.3"
run bvh 1000000 "--theta 0.3"
