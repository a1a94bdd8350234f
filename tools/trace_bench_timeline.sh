# GPU box: kernel + memory-copy timeline of a short bench run (do the copies run under the kernels?) -> gpurun_out/<tag>/timeline.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-timeline}
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o t -- python bench.py --no-cpu-baseline --no-encoder-fps --no-pmc --steps 4 --warmup 2 > $O/tr.log 2>&1 < /dev/null
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        tag = "ME" if "k_me" in n else "OIS" if "ois" in n else "PREP" if "prep" in n else "PACK" if "pack" in n else None
        if tag:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
for f in glob.glob(O + "/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", "") + " %.0fMB" % (int(r.get("Bytes", r.get("Size", 0)) or 0) / 1e6)))
ev.sort()
t0 = ev[0][0]
big = [e for e in ev if e[1] - e[0] > 200000]
with open(O + "/timeline.txt", "w") as out:
    for s, e, n in big[-70:]:
        print("%9.3f -> %9.3f ms  (%7.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n), file=out)
print(open(O + "/timeline.txt").read())
PY
rm -rf $O/tr
