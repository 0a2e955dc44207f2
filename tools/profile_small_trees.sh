cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_oct1e5 -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 100000 -s 210 --precision double --algorithm octree --workload galaxy --csv-total > $O/trace_oct1e5.txt 2>&1
f=$(find $O/trace_oct1e5 -name "*kernel_stats.csv" | head -1); cp $f $O/oct1e5_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bvh1e5 -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n 100000 -s 210 --precision double --algorithm bvh --workload galaxy --csv-total > $O/trace_bvh1e5.txt 2>&1
f=$(find $O/trace_bvh1e5 -name "*kernel_stats.csv" | head -1); cp $f $O/bvh1e5_kernel_stats.csv
cat $O/trace_oct1e5.txt $O/trace_bvh1e5.txt | tail -4
