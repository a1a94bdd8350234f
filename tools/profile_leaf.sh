# gpurun -- 'bash tools/profile_leaf.sh <tag>': tools/leaf_bench.py (4K) under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>/leaf_kernel_stats.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-leafprof}
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/leaf_bench.py 10 3840 2160 > $O/leaf.json 2> $O/prof.err < /dev/null
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python profiles/summarize_rocpd.py $DB "tools/leaf_bench.py 10 3840 2160 under rocprofv3 --kernel-trace --stats" > $O/leaf_kernel_stats.txt
  grep -i "sao\|full_loop\|pmcore" $O/leaf_kernel_stats.txt | head -12
else
  echo "no rocpd database"; tail -5 $O/prof.err
fi
rm -rf $O/prof
