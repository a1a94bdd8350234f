# GPU box: SQ counters of k_ois_picture (and k_prep_fused) in a short bench run -> gpurun_out/<tag>/ois_pmc.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-oispmc}
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o pmc --output-format csv -- python bench.py --inner --steps 1 --warmup 1 --batch 16 > $O/p$i.log 2>&1 < /dev/null
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_ois_picture" if "k_ois" in k else "k_prep_fused" if "k_prep" in k else None
        if not k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
with open(O + "/ois_pmc.txt", "w") as out:
    for k in sorted(acc):
        a = acc[k]
        print("%s  VGPR %s SGPR %s LDS %s scratch %s" % ((k,) + meta[k]), file=out)
        w = a.get("SQ_WAVES", 0)
        if w:
            print("  waves %d; per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SMEM %.0f" % ((w,) + tuple(a.get(c, 0) / w for c in
                  ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))), file=out)
        if a.get("SQ_WAVE_CYCLES"):
            wc = a["SQ_WAVE_CYCLES"]
            print("  wave cycles per wave %.0f (x4 clocks); of wave cycles: wait_any %.2f active_any %.2f active_valu %.2f active_lds %.2f active_sca %.2f" % ((wc / w if w else 0,) + tuple(
                a.get(c, 0) / wc for c in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"))), file=out)
        if a.get("SQ_LDS_IDX_ACTIVE"):
            print("  LDS bank conflict cycles / active %.2f" % (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]), file=out)
print(open(O + "/ois_pmc.txt").read())
PY
rm -rf $O/p[0-9]
