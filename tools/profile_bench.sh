# Run on the GPU box (gpurun -- 'bash tools/profile_bench.sh <tag>'): bench.py under rocprofv3 --kernel-trace --stats, summary into
# gpurun_out/<tag>/kernel_stats.txt (copy what should be judged into profiles/).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-prof}
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --no-cpu-baseline --no-encoder-fps --no-encode-pass --no-pmc > $O/prof_bench.json 2> $O/prof.err < /dev/null
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python profiles/summarize_rocpd.py $DB "bench.py (default steps; the front-half loop roofline is measured on; encoder / encode-pass / PMC legs off) under rocprofv3 --kernel-trace --stats" > $O/kernel_stats.txt
  head -10 $O/kernel_stats.txt
else
  echo "no rocpd database"; tail -5 $O/prof.err
fi
rm -rf $O/prof
