#!/bin/bash
# rocprofv3 kernel trace + PMC passes for the octree step (N=1e6 galaxy 3D double theta=0.5)
set -e
TAG=${1:-r01}
N=${2:-1000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_octree_$TAG
mkdir -p $OUT
CMD="$R/stdpar-nbody_amd/bin/nbody_hip_d3 -n $N -s 12 --algorithm octree --workload galaxy --precision double --csv-total"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.out 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.out 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.out 2> $OUT/pmc_sq2.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
cat $OUT/trace/*/*kernel_stats.csv
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $OUT/pmc_mem -- $CMD > $OUT/pmc_mem.out 2> $OUT/pmc_mem.err || true
