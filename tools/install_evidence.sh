#!/bin/bash
# After `gpurun -- bash tools/collect_evidence.sh <tag>`: copy the summaries worth keeping from gpurun_out/ (scratch) into
# profiles/<tag>/ (tracked).  Usage (in the build container, from the repo root): bash tools/install_evidence.sh r02
TAG=${1:-r06}
S=gpurun_out/$TAG; P=gpurun_out/profiles_$TAG; D=profiles/$TAG
mkdir -p $D
for f in bench_n1.json configs_all.json k1_times.json benchmark.log benchmark.csv benchmark_detailed.log benchmark_detailed.csv \
         pmc_k2_config3.txt pmc_k9_config4.txt energy_times.txt octree_times.txt step_graph.txt pmc_octree_walk_f32.txt \
         octree_times_walks.txt small_trees_kernel_stats.txt \
         pmc_k9_config4.json pmc_k2_config3.json k9_config4_kernel_stats.csv k2_config3_kernel_stats.csv config2_gaps.txt config2_kernel_stats.csv \
         config2_state_probe.txt config2_clock_ramp.txt k1_handoff_polls.txt k1_handoff_cost.txt k1_rule_instances.txt ab_k1_r3_r4_r5.txt \
         step_graph_small.txt prepare_kernel_time.txt k9_timeline.txt ab_k9_waves.txt cu_map_microbench.txt; do [ -f $S/$f ] && cp $S/$f $D/$f; done
for f in bench_n1_kernel_stats.csv bench_n1_kernel_stats_by_grid.csv bench_n1_pmc_all_pairs_force.json bench_n1_under_rocprof.json; do cp $P/$f $D/$f; done
cp $(ls -t $(find $S/trace_k2 -name "*kernel_stats.csv") | head -1) $D/config3_collapsed_kernel_stats.csv   # (the newest: gpurun_out/ keeps earlier sessions)
cp $(ls -t $(find $S/trace_k9 -name "*kernel_stats.csv") | head -1) $D/config4_bvh_kernel_stats.csv
cp $(ls -t $(find $S/trace_energy -name "*kernel_stats.csv") | head -1) $D/energies_kernel_stats.csv
git status --short $D
