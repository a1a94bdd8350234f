# GPU box (gpurun -- 'bash tools/trace_encoder_md.sh <tag> [frames] [lp]'): kernel timeline of the hooked encoder at BASELINE configs[2] with SVT_HOOK_MD=1 under
# rocprofv3 --kernel-trace: when do the pictures' mode-decision kernels run, how many at a time, on which queues -> gpurun_out/<tag>/md_timeline.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-trace_md}
N=${2:-48}
LP=${3:-32}
mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/md_clip.yuv", "motion", 3840, 2160, 16, 7)
PY
shift 3 2>/dev/null
env SVT_HOOK_MD=${TL_MD:-pb} SVT_HOOK_REPORT=$O/report.txt "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n $N -nb 16 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -lp $LP > $O/app.txt 2> $O/prof.err < /dev/null
grep "Average Speed" $O/app.txt
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
md, other = [], []
for f in glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        rec = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n[:40])
        (md if "k_md_encode_picture" in n else other).append(rec)
md.sort()
with open(O + "/events.txt", "w") as ev:   # every kernel of the run: start end (us since the first) queue name - for offline questions
    allk = sorted(md + other)
    for s_, e_, q_, n_ in allk:
        print("%.1f %.1f %s %s" % ((s_ - allk[0][0]) / 1e3, (e_ - allk[0][0]) / 1e3, q_, n_.replace(" ", "_")), file=ev)
if not md:
    print("no mode-decision kernel in the trace"); sys.exit(0)
t0 = md[0][0]
with open(O + "/md_timeline.txt", "w") as out:
    print("k_md_encode_picture launches: %d; queues used: %s" % (len(md), sorted(set(m[2] for m in md))), file=out)
    for s, e, q, n in md:
        conc = sum(1 for s2, e2, _, _ in md if s2 < e and e2 > s)
        print("%9.2f -> %9.2f ms (%7.2f)  queue %s  overlapping launches %d" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, conc), file=out)
    # concurrency histogram over time (1 ms samples)
    end = max(e for _, e, _, _ in md)
    hist = {}
    t = t0
    while t < end:
        c = sum(1 for s, e, _, _ in md if s <= t < e)
        hist[c] = hist.get(c, 0) + 1
        t += 1000000
    tot = sum(hist.values())
    print("share of wall time with k mode-decision kernels running: " + ", ".join("%d: %.0f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())), file=out)
    durs = sorted((e - s) / 1e6 for s, e, _, _ in md)
    print("duration ms: min %.1f median %.1f max %.1f; sum %.0f over wall %.0f ms" % (durs[0], durs[len(durs) // 2], durs[-1], sum(durs), (end - t0) / 1e6), file=out)
    print("other kernels: %d launches, %.0f ms summed" % (len(other), sum((e - s) for s, e, _, _ in other) / 1e6), file=out)
    byname = {}
    for s, e, _, n in other:
        d = byname.setdefault(n, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e6; d[2] = max(d[2], (e - s) / 1e6)
    for n, d in sorted(byname.items(), key=lambda kv: -kv[1][1])[:12]:
        print("    %-42s %5d launches %9.1f ms summed, longest %.2f ms" % (n, d[0], d[1], d[2]), file=out)
print(open(O + "/md_timeline.txt").read())
PY
grep "mode decision" $O/report.txt | cut -c1-400
rm -rf $O/tr /tmp/md_clip.yuv /tmp/md.265
