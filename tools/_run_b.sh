set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01_b
timeout 900 python bench.py > gpurun_out/r01_b/bench.json 2> gpurun_out/r01_b/bench.err
tail -1 gpurun_out/r01_b/bench.json | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r01_b/prof -- python bench.py --no-cpu-baseline > gpurun_out/r01_b/prof_bench.json 2> gpurun_out/r01_b/prof.err
DB=$(find gpurun_out/r01_b/prof -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB "round 1 (b): bench.py default (16 pictures/step, 20 steps + 3 warmup), prep + batched ME + batched OIS, 1080p cfg2" > gpurun_out/r01_b/kernel_stats.txt
head -12 gpurun_out/r01_b/kernel_stats.txt
rm -rf gpurun_out/r01_b/prof
