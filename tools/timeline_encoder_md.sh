# GPU box: the bindings' own time line (SVT_HOOK_TIMELINE) of the hooked encoder at BASELINE configs[2] with SVT_HOOK_MD=1: per picture the device call, the host work
# around it and the EncodePass phase -> gpurun_out/<tag>/timeline.txt + a summary.  usage: bash tools/timeline_encoder_md.sh <tag> [frames] [lp] [extra env K=V ...]
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-tl_md}
N=${2:-48}
LP=${3:-32}
shift 3 2>/dev/null
mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/md_clip.yuv", "motion", 3840, 2160, 16, 7)
PY
if [ "$TL_MD" = "off" ]; then MDENV="SVT_HOOK_NOTHING=1"; else MDENV="SVT_HOOK_MD=${TL_MD:-1}"; fi   # TL_MD=off: front half only (the host's own EncDec on the time line)
env $MDENV SVT_HOOK_REPORT=$O/report.txt SVT_HOOK_TIMELINE=$O/timeline.txt "$@" timeout 200 integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n $N -nb 16 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -lp $LP > $O/app.txt 2> $O/app.err < /dev/null
grep "Average Speed" $O/app.txt
python - "$O" <<'PY'
import sys
O = sys.argv[1]
ev = [l.split() for l in open(O + "/timeline.txt")]
ev = [(k, int(p), int(a), int(b), float(t0), float(t1)) for k, p, a, b, t0, t1 in ev]
pics = {}
pads = sorted((t0, t1, a) for k, p, a, b, t0, t1 in ev if k == "refpad")
for k, p, a, b, t0, t1 in ev:
    if k in ("refupload", "refpad"):
        continue
    d = pics.setdefault(p, {"tl": a, "slice": b})
    d[k] = (t0, t1)
print("picture tl slice | md start  fill  refs  call(ms) | encodepass first..last (ms) | total residence")
for p in sorted(pics):
    d = pics[p]
    s = d.get("md_fill", d.get("md_host", (0, 0)))[0]
    ep = d.get("encodepass", (0, 0))
    fmt = lambda x: "%6.1f" % (x[1] - x[0]) if x else "   -  "
    print("%4d %2d %2d | %8.1f %s %s %s | %8.1f .. %8.1f (%6.1f) | %7.1f %s" % (p, d["tl"], d["slice"], s, fmt(d.get("md_fill")), fmt(d.get("md_refs")), fmt(d.get("md_call")), ep[0], ep[1],
          ep[1] - ep[0], ep[1] - s, "HOST" if "md_host" in d else ""))
# the base-layer chain of each GOP: start of a host picture's decision -> its last LCU through EncodePass -> start of the NEXT base picture of the chain
base = sorted(p for p in pics if "lcu0" in pics[p] and pics[p]["tl"] == 0)
steps = [pics[b]["lcu0"][0] - pics[a]["lcu0"][0] for a, b in zip(base, base[1:]) if pics[b]["lcu0"][0] > pics[a]["lcu0"][0]]
if steps:
    import statistics as st0
    print("base-layer pictures: first LCU of one -> first LCU of the next of the same GOP chain: median %.1f ms over %d steps (min %.1f, max %.1f)" %
          (st0.median(steps), len(steps), min(steps), max(steps)))
host = sorted(p for p in pics if "md_host" in pics[p] and pics[p]["tl"] == 0)
rows = []
for a, b in zip(host, host[1:]):
    if b - a != 4 or "encodepass" not in pics[a]:
        continue
    s0, e0 = pics[a].get("lcu0", pics[a]["md_host"])[0], pics[a]["encodepass"][1]
    s1 = pics[b].get("lcu0", pics[b]["md_host"])[0]
    if s1 <= s0:
        continue
    # the tail of picture a on the thread that finished its last LCU: SAO of the picture, then the three padding calls (luma first)
    after = [q for q in pads if q[0] >= e0 - 0.5 and q[0] < s1 and q[2] >= 3000]
    sao = pad = rest = None
    if after:
        p0 = after[0][0]
        p1 = max(q[1] for q in pads if q[0] >= p0 and q[0] < p0 + 30.0)
        p1 = max(q[1] for q in pads if q[0] >= p0 and q[0] < p0 + 2.0)
        sao, pad, rest = p0 - e0, p1 - p0, s1 - p1
    rows.append((e0 - s0, s1 - e0, s1 - s0, sao, pad, rest))
if rows:
    import statistics as st
    med = lambda k: st.median(r[k] for r in rows if r[k] is not None) if any(r[k] is not None for r in rows) else float("nan")
    print("base-layer chain (host), %d steps: first LCU -> last LCU %.1f ms (median), last LCU -> next base picture's first LCU %.1f ms, step %.1f ms" % (len(rows), med(0), med(1), med(2)))
    print("   of the gap: last LCU -> padding starts (ApplySaoOffsetsPicture, one thread) %.1f ms, padding %.1f ms, padding done -> next base picture's first LCU "
          "(picture manager, rate control, mode-decision configuration, queues) %.1f ms" % (med(3), med(4), med(5)))
ups = [(t1 - t0, a) for k, p, a, b, t0, t1 in ev if k == "refupload"]
if ups:
    print("reference uploads: %d, ms each: %s" % (len(ups), " ".join("%.1f" % u[0] for u in ups[:40])))
PY
grep "mode decision" $O/report.txt | cut -c1-420
rm -f /tmp/md_clip.yuv /tmp/md.265
