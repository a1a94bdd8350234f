# Run on the GPU box (gpurun -- 'bash tools/profile_md_bench.sh <tag>'): tools/md_bench.py (4K non-reference B pictures: mode decision + encode pass, one call per picture)
# under rocprofv3 --kernel-trace --stats, summary into gpurun_out/<tag>/md_kernel_stats.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-mdprof}
mkdir -p $O
timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/md_bench.py 3840 2160 7 4 inter 3 > $O/md_bench.json 2> $O/prof.err < /dev/null
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python profiles/summarize_rocpd.py $DB "tools/md_bench.py 3840 2160 7 4 inter 3 (svt_amd_md_encode_picture_inter of 4K non-reference B pictures, decisions checked against the reference's records) under rocprofv3 --kernel-trace --stats" > $O/md_kernel_stats.txt
  head -8 $O/md_kernel_stats.txt
else
  echo "no rocpd database"; tail -5 $O/prof.err
fi
rm -rf $O/prof
