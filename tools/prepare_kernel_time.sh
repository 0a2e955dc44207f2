cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for N in 8192 65536 262144 1048576; do
  rm -rf /tmp/pt; S=30; [ $N -ge 1000000 ] && S=13
  C2_N=$N C2_STEPS=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $R/tools/c2_eager.py > /dev/null 2>&1
  echo "n=$N"; grep -E "k1_prepare|all_pairs_force_sgpr" $(ls /tmp/pt/*/*kernel_stats.csv) | awk -F'","|",' '{print "   ", substr($1,1,60), "calls", $2, "avg ns", $4}'
done
