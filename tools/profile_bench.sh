#!/bin/bash
# Profiles the bench command with rocprofv3 (kernel trace + stats, then separate PMC passes: never combined with a trace
# domain) into gpurun_out/prof_<tag>, then tools/summarize_profile.py turns that into the stamped summaries under
# profiles/<tag>/ that bench.py accepts as evidence (same kernel description string, same K1 source hash).
# Usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
python3 $R/tools/summarize_profile.py $OUT $R/gpurun_out/profiles_$TAG
ls $R/gpurun_out/profiles_$TAG
