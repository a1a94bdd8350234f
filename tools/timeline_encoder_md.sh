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
env SVT_HOOK_MD=1 SVT_HOOK_REPORT=$O/report.txt SVT_HOOK_TIMELINE=$O/timeline.txt "$@" timeout 200 integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n $N -nb 16 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -lp $LP > $O/app.txt 2> $O/app.err < /dev/null
grep "Average Speed" $O/app.txt
python - "$O" <<'PY'
import sys
O = sys.argv[1]
ev = [l.split() for l in open(O + "/timeline.txt")]
ev = [(k, int(p), int(a), int(b), float(t0), float(t1)) for k, p, a, b, t0, t1 in ev]
pics = {}
for k, p, a, b, t0, t1 in ev:
    if k == "refupload":
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
ups = [(t1 - t0, a) for k, p, a, b, t0, t1 in ev if k == "refupload"]
if ups:
    print("reference uploads: %d, ms each: %s" % (len(ups), " ".join("%.1f" % u[0] for u in ups[:40])))
PY
grep "mode decision" $O/report.txt | cut -c1-420
rm -f /tmp/md_clip.yuv /tmp/md.265
