# GPU box: round 6's closing measurements -> gpurun_out/r06_final/ (copied to profiles/ by hand)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
# rocprofv3 --kernel-trace --stats of the encoder in the measured configuration (the library's defaults), 160 pictures, -lp 32
python - <<PY
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/md_clip.yuv", "motion", 3840, 2160, 16, 7)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n 160 -nb 16 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -lp 32 > $O/app.txt 2> $O/prof.err < /dev/null
grep "Average Speed" $O/app.txt
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        a = acc[n]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
with open(O + "/kernel_stats.txt", "w") as out:
    print("# SvtHevcEncApp_hip cfg3 160 pictures, library defaults (closed loop of P / B pictures, pool 16, 12 + 4 lanes), -lp 32, under rocprofv3 --kernel-trace --stats", file=out)
    print("# durations in microseconds", file=out)
    print("%-64s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"), file=out)
    for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-64s %8d %14.1f %12.3f %8.2f" % (n[:64], a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot), file=out)
print(open(O + "/kernel_stats.txt").read())
PY
rm -rf $O/tr /tmp/md_clip.yuv /tmp/md.265
python -c "import __graft_entry__ as G; G.smoke()" > $O/smoke.txt 2>&1
tail -1 $O/smoke.txt
