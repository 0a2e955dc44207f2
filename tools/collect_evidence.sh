#!/bin/bash
# One GPU-box session that produces everything kept under profiles/<tag>/: the bench line, its rocprofv3 summaries (stamped),
# the per-config timings, the reference-shaped benchmark logs + scraped tables, and PMC passes of K2 and K9.
# Usage (from the repo root on the GPU box): bash tools/collect_evidence.sh r05   (add `short` as a second argument to skip the
# reference-shaped matrix and the octree / small-tree sets: the parts round 5 did not touch)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== bench"; timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
echo "== profile bench"; timeout -k 10 500 bash tools/profile_bench.sh $TAG > $O/profile_bench.txt 2>&1; tail -5 $O/profile_bench.txt
echo "== configs"; timeout -k 10 300 python3 tools/measure_configs.py > $O/configs_all.json 2> $O/configs_all.err; grep -c config $O/configs_all.json
echo "== K1 sizes"; timeout -k 10 200 python3 tools/time_all_pairs.py 10 > $O/k1_times.json 2> $O/k1_times.err
echo "== round 5: stamped PMC summaries of K2 / K9, config 2's gaps and states, K1's hand-off"
timeout -k 10 400 python3 tools/pmc_configs.py $O > $O/pmc_configs.txt 2>&1; cat $O/pmc_configs.txt
timeout -k 10 300 bash tools/config2_gaps.sh $TAG > /dev/null 2>&1; tail -30 $O/config2_gaps.txt
timeout -k 10 200 python3 tools/c2_state_probe.py > $O/config2_state_probe.txt 2>&1
timeout -k 10 100 python3 tools/c2_clock_ramp.py > $O/config2_clock_ramp.txt 2>&1
timeout -k 10 100 bash tools/prepare_kernel_time.sh > $O/prepare_kernel_time.txt 2>&1
timeout -k 10 200 python3 tools/k1_handoff_report.py > $O/k1_handoff_polls.txt 2>&1; cat $O/k1_handoff_polls.txt
timeout -k 10 300 python3 tools/k1_handoff_cost.py > $O/k1_handoff_cost.txt 2>&1; cat $O/k1_handoff_cost.txt
timeout -k 10 300 python3 tools/k1_rule_instances.py > $O/k1_rule_instances.txt 2>&1; cat $O/k1_rule_instances.txt
if [ -f stdpar-nbody_amd/libnbody_hip_var_r3.so ] && [ -f stdpar-nbody_amd/libnbody_hip_var_r4head.so ]; then
  AB_CASES="f32:uniform:262144,f32:galaxy:262144,f32:uniform:100000,f64:uniform:65536,f64:galaxy:262144,f64:galaxy:1048576" timeout -k 10 600 python3 tools/ab_all_pairs.py libnbody_hip_var_r3.so libnbody_hip_var_r4head.so libnbody_hip.so > $O/ab_k1_r3_r4_r5.txt 2>&1; cat $O/ab_k1_r3_r4_r5.txt
fi
timeout -k 10 100 python3 tools/k9_timeline.py > $O/k9_timeline.txt 2>&1
[ -f stdpar-nbody_amd/libnbody_hip_var_pre8.so ] && timeout -k 10 200 python3 tools/ab_k9.py > $O/ab_k9_waves.txt 2>&1
[ -x tools/microbench/cu_map ] && timeout -k 10 60 ./tools/microbench/cu_map > $O/cu_map_microbench.txt 2>&1
STEP_GRAPH_N=64,257,513,1000,1024,2048 timeout -k 10 200 python3 tools/time_step_graph.py > $O/step_graph_small.txt 2>&1; STEP_GRAPH_N=64,257,513,1000,1024,2048 timeout -k 10 200 python3 tools/time_step_graph.py float >> $O/step_graph_small.txt 2>&1
if [ "$2" = "short" ]; then exit 0; fi
echo "== matrix"; timeout -k 10 400 bash tools/benchmark.sh 200 > $O/benchmark.log 2> $O/benchmark.err; python3 tools/scrape_bench_log.py $O/benchmark.log > $O/benchmark.csv; cat $O/benchmark.csv
echo "== detailed"; timeout -k 10 400 bash tools/benchmark_detailed.sh 1000 100000 > $O/benchmark_detailed.log 2> $O/benchmark_detailed.err; python3 tools/scrape_bench_log.py $O/benchmark_detailed.log > $O/benchmark_detailed.csv; cat $O/benchmark_detailed.csv
echo "== PMC K2 / K9"
C1="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
C2="SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
PREC=float WL=uniform TAG=pmc_k2 bash tools/pmc_kernel.sh all_pairs_collapsed_ all-pairs-collapsed 262144 "$C1" "$C2" "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_k2_config3.txt 2>&1
TAG=pmc_k9 bash tools/pmc_kernel.sh bvh_force_sweep_isa bvh 1000000 "$C1" "$C2" "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_k9_config4.txt 2>&1
cat $O/pmc_k2_config3.txt $O/pmc_k9_config4.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k2 -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 262144 -s 12 --precision float --algorithm all-pairs-collapsed --workload uniform --csv-total > $O/trace_k2.txt 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k9 -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 1000000 -s 12 --precision double --algorithm bvh --workload galaxy --csv-total > $O/trace_k9.txt 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_energy -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 262144 -s 3 --precision double --algorithm all-pairs --workload galaxy --csv-detailed --save energy > $O/trace_energy.txt 2>&1
cd $R; timeout -k 10 120 python3 tools/time_energy.py > $O/energy_times.txt 2>&1
{ timeout -k 10 120 python3 tools/time_octree.py 1000000 double; timeout -k 10 120 python3 tools/time_octree.py 1000000 float; timeout -k 10 60 python3 tools/time_octree.py 100000 float; } > $O/octree_times.txt 2>&1
timeout -k 10 200 python3 tools/time_step_graph.py > $O/step_graph.txt 2>&1; timeout -k 10 200 python3 tools/time_step_graph.py float >> $O/step_graph.txt 2>&1
PREC=float TAG=pmc_otf bash tools/pmc_kernel.sh ot_force_isa_f32 octree 1000000 "$C1" "$C2" "FETCH_SIZE" > $O/pmc_octree_walk_f32.txt 2>&1
(for d in 3 2; do for p in double float; do timeout -k 10 120 python3 tools/time_octree.py 1000000 $p galaxy $d || exit 1; done; done) 2>&1 | grep -v amdgpu > $O/octree_times_walks.txt
bash tools/profile_small_trees.sh $TAG > /dev/null 2>&1
cd /tmp
find $O/trace_k2 $O/trace_k9 $O/trace_energy -name "*kernel_stats.csv" | while read f; do echo $f; head -6 $f; done
