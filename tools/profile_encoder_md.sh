# Run on the GPU box (gpurun -- 'bash tools/profile_encoder_md.sh <tag> [frames]'): the hooked encoder at BASELINE configs[2] with SVT_HOOK_MD=pb under
# rocprofv3 --kernel-trace --stats: how long the mode-decision kernel takes INSIDE the encoder (other pictures' kernels sharing the device).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-prof_md}
N=${2:-160}
mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/md_clip.yuv", "motion", 3840, 2160, 16, 7)
PY
SVT_HOOK_MD=pb SVT_HOOK_PCS_POOL=12 SVT_HOOK_REPORT=$O/report.txt timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n $N -nb 16 -lp 32 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 > $O/app.txt 2> $O/prof.err < /dev/null
grep "Average Speed" $O/app.txt
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python profiles/summarize_rocpd.py $DB "SvtHevcEncApp_hip cfg3 $N pictures, SVT_HOOK_MD=pb SVT_HOOK_PCS_POOL=12 -lp 32 (the run `value` of bench.py is measured on), under rocprofv3 --kernel-trace --stats" > $O/kernel_stats.txt
  head -14 $O/kernel_stats.txt
else
  echo "no rocpd database"; tail -5 $O/prof.err
fi
grep "mode decision" $O/report.txt | cut -c1-260
rm -rf $O/prof /tmp/md_clip.yuv /tmp/md.265
