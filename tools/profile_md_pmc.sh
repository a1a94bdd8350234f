# Run on the GPU box (gpurun -- 'bash tools/profile_md_pmc.sh <tag>'): the mode-decision + encode-pass kernel of tools/md_bench.py (4K non-reference B pictures recorded from the
# reference on the box) under rocprofv3 PMC passes (counters only, no tracing domains), per-kernel sums into gpurun_out/<tag>/md_pmc.txt.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${1:-pmc}


mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$MD_PMC_ONLY" ]; then case " $MD_PMC_ONLY " in *" $i "*) ;; *) continue;; esac; fi   # MD_PMC_ONLY="1 2 5": only these passes
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o pmc --output-format csv -- python tools/md_bench.py 3840 2160 7 2 inter 5 > $O/p$i.log 2>&1 < /dev/null
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
meta = {}
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_md_encode_picture" not in k:
            continue
        k = "k_md_encode_picture<inter>" if "Lb1" in k or "<true" in k else "k_md_encode_picture<intra>"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = (r["VGPR_Count"], r.get("Accum_VGPR_Count"), r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
with open(O + "/md_pmc.txt", "w") as out:
    for k in sorted(acc):
        print("%s  VGPR %s AGPR %s SGPR %s LDS %s scratch %s (sums over the dispatches of the run: 2 B pictures x 2 calls)" % ((k,) + meta[k]), file=out)
        a = acc[k]
        for c in sorted(a):
            print("  %-26s %16.0f" % (c, a[c]), file=out)
        w = a.get("SQ_WAVES", 0)
        if w:
            print("  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SMEM %.0f" % tuple(a.get(c, 0) / w for c in
                  ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")), file=out)
        if a.get("SQ_WAVE_CYCLES"):
            wc = a["SQ_WAVE_CYCLES"]
            print("  of wave cycles: wait_any %.2f wait_inst_any %.2f active_any %.2f active_valu %.2f active_lds %.2f active_sca %.2f" % tuple(
                a.get(c, 0) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")), file=out)
        if a.get("SQ_LDS_IDX_ACTIVE"):
            print("  LDS bank conflict cycles / active %.2f" % (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]), file=out)
        if a.get("SQC_ICACHE_REQ"):
            print("  instruction cache: hit rate %.3f, misses per wave %.0f, fetch latency %.0f cycles (SQ_IFETCH_LEVEL / SQ_IFETCH)" % (
                a.get("SQC_ICACHE_HITS", 0) / a["SQC_ICACHE_REQ"], a.get("SQC_ICACHE_MISSES", 0) / 992.0, a.get("SQ_IFETCH_LEVEL", 0) / max(1.0, a.get("SQ_IFETCH", 0))), file=out)
        if "FETCH_SIZE" in a:
            print("  FETCH_SIZE %.1f MB raw (x2 gfx950 correction: %.1f MB)  WRITE_SIZE %.1f MB" % (a["FETCH_SIZE"] / 1024, a["FETCH_SIZE"] / 512, a.get("WRITE_SIZE", 0) / 1024), file=out)
print(open(O + "/md_pmc.txt").read())
PY
rm -rf $O/p[0-9]
