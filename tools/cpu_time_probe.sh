# GPU box: wall / user / sys CPU seconds of the reference and the hooked encoder on the same clip (are waiting threads spinning?)
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/c.yuv", "motion", 3840, 2160, 16, 7)
PY
ARGS="-i /tmp/c.yuv -w 3840 -h 2160 -n 96 -nb 16 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1"
for app in oracle/_ref/SvtHevcEncApp_ref integration/_build/SvtHevcEncApp_hip; do
  /usr/bin/time -f "$app wall %e s user %U s sys %S s ctx-switches vol %w invol %c" $app $ARGS -b /tmp/o.265 2>&1 | grep -E "Average Speed|wall"
done
nproc; lscpu | grep -E "Thread|Core|Socket|Model name" 
