# rocprofv3 kernel statistics of the CLI's recorded octree step loop at sizes one block builds (n <= 2048), per step,
# into gpurun_out/<tag>/tiny_trees_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04}
mkdir -p $O
S=210
: > $O/tiny_trees_kernel_stats.txt
for cfg in "octree 1000 float" "octree 2048 float" "octree 1000 double" "bvh 1000 float" "bvh 1000 double"; do
  set -- $cfg
  tag=$1_$2_$3
  rm -rf $O/trace_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- $R/stdpar-nbody_amd/bin/nbody_hip_d3 -n $2 -s $S --precision $3 --algorithm $1 --workload galaxy --csv-total > $O/trace_$tag.txt 2>&1 || exit 1
  f=$(find $O/trace_$tag -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/summarize_kernel_stats.py $f $S "$tag" >> $O/tiny_trees_kernel_stats.txt
  rm -rf $O/trace_$tag
done
cat $O/tiny_trees_kernel_stats.txt
