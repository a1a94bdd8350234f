# gpurun -- 'bash tools/binding_coverage.sh': the hooked reference encoder with every binding switched on, on small clips of the
# BASELINE presets; prints, per binding, how many calls the device answered and how many were left to the reference code.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/coverage
mkdir -p $O
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/cov8.yuv", "motion", 416, 240, 9, 7)
S.write_clip10("/tmp/cov10.yuv", "motion", 416, 240, 5, 7)
PY
export SVT_HOOK_VERBOSE=1 SVT_HOOK_FULLLOOP=1 SVT_HOOK_RECON=1 SVT_HOOK_INTRA=1 SVT_HOOK_INTER=1 SVT_HOOK_QUANT=1 SVT_HOOK_SAO=1
run() { # tag yuv frames args...
  tag=$1; yuv=$2; n=$3; shift 3
  SVT_HOOK_REPORT=$O/$tag.report timeout 200 integration/_build/SvtHevcEncApp_hip -i $yuv -w 416 -h 240 -n $n -q 32 -asm 0 -b /tmp/cov.265 "$@" > /dev/null 2> $O/$tag.err < /dev/null
  echo "== $tag: $*" >> $O/coverage.txt
  cat $O/$tag.report >> $O/coverage.txt
}
: > $O/coverage.txt
run m9_ldp /tmp/cov8.yuv 9 -encMode 9 -pred-struct 0
run m7_ra_sao /tmp/cov8.yuv 9 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1
run m4_ra /tmp/cov8.yuv 9 -encMode 4 -pred-struct 2 -hierarchical-levels 2
run m7_10bit /tmp/cov10.yuv 5 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -bit-depth 10
cat $O/coverage.txt
