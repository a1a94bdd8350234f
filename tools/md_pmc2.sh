# GPU box: PMC passes (counters only) over the mode-decision kernel of tools/md_chain2.py (picture indices $2, default 0: a layer-2 B picture of the 4K fixture; 5 dispatches).
# usage: bash tools/md_pmc2.sh <tag> [pictures] ; env MD_PMC_SETS="1 2 4" selects passes, SVT_PRODUCT_LIB an experimental build
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-pmc2}
PICS=${2:-0}
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$MD_PMC_SETS" ]; then case " $MD_PMC_SETS " in *" $i "*) ;; *) continue;; esac; fi
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o pmc --output-format csv -- python tools/md_chain2.py $PICS > $O/p$i.log 2>&1 < /dev/null
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
meta = {}
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_md_picture" not in k and "k_encode_picture" not in k:
            continue
        k = k.split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
        meta[k] = (r["VGPR_Count"], r.get("Accum_VGPR_Count"), r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
with open(O + "/md_pmc.txt", "w") as out:
    for k in sorted(acc):
        print("%s  VGPR %s AGPR %s SGPR %s LDS %s scratch %s" % ((k,) + meta[k]), file=out)
        a = acc[k]
        for c in sorted(a):
            print("  %-28s %16.0f  per dispatch %14.0f (%d dispatches)" % (c, a[c], a[c] / n[k][c], n[k][c]), file=out)
        if a.get("SQ_WAVE_CYCLES"):
            wc = a["SQ_WAVE_CYCLES"]
            print("  of wave cycles: wait_any %.2f wait_inst_any %.2f active_any %.2f active_valu %.2f active_lds %.2f active_sca %.2f" % tuple(
                a.get(c, 0) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")), file=out)
        if "FETCH_SIZE" in a:
            print("  FETCH_SIZE per dispatch %.1f MB raw (x2 gfx950 correction: %.1f MB)" % (a["FETCH_SIZE"] / n[k]["FETCH_SIZE"] / 1024, a["FETCH_SIZE"] / n[k]["FETCH_SIZE"] / 512), file=out)
        if "WRITE_SIZE" in a:
            print("  WRITE_SIZE per dispatch %.1f MB" % (a["WRITE_SIZE"] / n[k]["WRITE_SIZE"] / 1024), file=out)
print(open(O + "/md_pmc.txt").read())
PY
rm -rf $O/p[0-9]
