# Which lines / branches of svt-hevc_amd/csrc/md_logic.h (the decision code the HIP kernel and the CPU checker share) do the REFERENCE's records reach?
# Builds the checker with gcov instrumentation in a scratch directory, runs the fixture test that compares it with the recorded ModeDecisionLcu calls
# (tests/test_oracle_md_golden.py), and prints the coverage of md_logic.h with every line no record reaches.  CPU only.   usage: bash tools/md_logic_coverage.sh [out.txt]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/profiles/md_logic_coverage.txt}
W=$(mktemp -d /tmp/mdcov.XXXX)
cd $W
for f in $ROOT/oracle/svt_oracle_*.c; do
  gcc -std=gnu99 -O0 -fPIC --coverage -w -I$ROOT/oracle -c $f -o $(basename $f .c).o
done
gcc -shared --coverage -o liboracle_cov.so *.o
SVT_ORACLE_LIB=$W/liboracle_cov.so python -m pytest $ROOT/tests/test_oracle_md_golden.py -q -x -p no:cacheprovider > test.log 2>&1 || { tail -5 test.log; exit 1; }
gcov -b -c svt_oracle_md.c > gcov.log 2>&1 || true
G=$(ls *md_logic.h.gcov | head -1)
python3 - "$G" "$OUT" <<'PY'
import re, sys
g, out = sys.argv[1], sys.argv[2]
lines = open(g).read().splitlines()
exe = miss = 0
missed, br_tot, br_miss = [], 0, 0
cur = None
for l in lines:
    m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", l)
    if m:
        cnt, no, src = m.group(1).strip(), int(m.group(2)), m.group(3)
        cur = (no, src)
        if cnt == "-" or no == 0:
            continue
        if cnt.startswith("#####") or cnt.startswith("====="):
            miss += 1
            missed.append((no, src.rstrip()))
        else:
            exe += 1
    elif l.startswith("branch"):
        br_tot += 1
        if "never executed" in l or "taken 0" in l:
            br_miss += 1
with open(out, "w") as f:
    print("md_logic.h under the reference's recorded ModeDecisionLcu calls (tests/golden/md_*.npz through tests/test_oracle_md_golden.py, gcov -b of the CPU checker):", file=f)
    print("lines executed %d of %d (%.1f%%); branch outcomes taken %d of %d (%.1f%%)" % (exe, exe + miss, 100.0 * exe / max(1, exe + miss), br_tot - br_miss, br_tot,
          100.0 * (br_tot - br_miss) / max(1, br_tot)), file=f)
    print("lines no reference record reaches (%d):" % miss, file=f)
    for no, src in missed:
        print("  %4d: %s" % (no, src[:150]), file=f)
print(open(out).read()[:6000])
PY
rm -rf $W
