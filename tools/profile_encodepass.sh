# Run on the GPU box (gpurun -- 'bash tools/profile_encodepass.sh <tag>'): tools/encodepass_bench.py under rocprofv3 --kernel-trace --stats,
# summary into gpurun_out/<tag>/encodepass_kernel_stats.txt (copy what should be judged into profiles/).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-prof}
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/encodepass_bench.py 3840 2160 3 > $O/prof_encodepass_bench.txt 2> $O/prof.err < /dev/null
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python profiles/summarize_rocpd.py $DB "tools/encodepass_bench.py 3840 2160 3 under rocprofv3 --kernel-trace --stats" > $O/encodepass_kernel_stats.txt
  head -12 $O/encodepass_kernel_stats.txt
else
  echo "no rocpd database"; tail -5 $O/prof.err
fi
rm -rf $O/prof
