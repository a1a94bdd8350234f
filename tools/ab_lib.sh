# same-box A/B of two library builds (svt-hevc_amd/libsvt_old.so.bin = the build before a change) on the encode-pass and mode-decision benches
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-ab}; mkdir -p $O
for v in new old new old; do
  if [ $v = old ]; then cp svt-hevc_amd/libsvt_hevc_amd.so /tmp/keep.so; cp svt-hevc_amd/libsvt_old.so.bin svt-hevc_amd/libsvt_hevc_amd.so; fi
  python tools/md_bench.py 3840 2160 7 4 inter 5 > $O/b_$v.json 2>/dev/null
  python - <<PY
import json, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
d = json.load(open("$O/b_$v.json")); print("$v md B 4K", d["ms_per_picture_incl_host_copies"])
import svtlib as S, encodepass_bench as EPB, ctypes as C
lib = S.load_product(); ctx = C.c_void_p(); assert lib.svt_amd_context_create(0, 640, 384, 1, C.byref(ctx)) == 0
r = EPB.measure_b_picture(lib, ctx)
print("$v ep alone", r["ms_per_picture_alone"], "10% intra", r["intra_units_in_10_percent_of_the_lcus"]["ms_per_picture_alone"], "4 in flight", r["intra_units_in_10_percent_of_the_lcus"]["pictures_per_s_4_in_flight"], "16 in flight", r["pictures_per_s_16_in_flight"])
PY
  if [ $v = old ]; then cp /tmp/keep.so svt-hevc_amd/libsvt_hevc_amd.so; fi
done
