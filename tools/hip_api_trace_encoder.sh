# GPU box: HIP API calls of the hooked encoder (closed loop on the device) by name - calls, total, longest - to find host calls that block while kernels run.
# usage: bash tools/hip_api_trace_encoder.sh <tag> [frames] [lp] [K=V ...]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-hip_api}
N=${2:-160}
LP=${3:-32}
shift 3 2>/dev/null
mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/md_clip.yuv", "motion", 3840, 2160, 16, 7)
PY
env SVT_HOOK_MD=${TL_MD:-pb} SVT_HOOK_REPORT=$O/report.txt "$@" timeout 400 rocprofv3 --hip-runtime-trace --output-format csv -d $O/tr -o t -- integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n $N -nb 16 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -lp $LP > $O/app.txt 2> $O/prof.err < /dev/null
grep "Average Speed" $O/app.txt
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
by = {}
rows = []
for f in glob.glob(O + "/tr/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Function"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        e = by.setdefault(n, [0, 0.0, 0.0])
        e[0] += 1; e[1] += d; e[2] = max(e[2], d)
        if d > 20.0:
            rows.append((int(r["Start_Timestamp"]), d, n, r.get("Thread_Id", "?")))
print("%-40s %8s %12s %10s" % ("HIP call", "calls", "total ms", "longest ms"))
for n, e in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-40s %8d %12.1f %10.1f" % (n, e[0], e[1], e[2]))
rows.sort()
t0 = rows[0][0] if rows else 0
print("calls longer than 20 ms, in time order (ms since the first of them):")
for s, d, n, t in rows[:120]:
    print("  %9.1f %8.1f ms %-32s thread %s" % ((s - t0) / 1e6, d, n, t))
PY
rm -rf $O/tr /tmp/md_clip.yuv /tmp/md.265
