# GPU box: same-box A/B of two library builds in the closed loop (the bound encoder at BASELINE configs[2] with the library's defaults): alternates NEW (the tree's library) and OLD
# (tools/_exp/lib_<tag>.so copied over it) in blocks of 6 encodes, prints the medians.  usage: bash tools/ab_closed_loop.sh <tag> [blocks per side, default 2]
cd ${GRAFT_REPO_ROOT:-.}
tag=$1; n=${2:-2}
cp svt-hevc_amd/libsvt_hevc_amd.so /tmp/lib_new.so
for b in $(seq 1 $n); do
  for v in new old; do
    if [ $v = old ]; then cp tools/_exp/lib_$tag.so svt-hevc_amd/libsvt_hevc_amd.so; else cp /tmp/lib_new.so svt-hevc_amd/libsvt_hevc_amd.so; fi
    python tools/stress_closed_loop.py 6 160 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['fps_min'], d['fps_median'], d['fps_max'], 'differ', d['bitstreams_differing'], 'aborted', d['aborted_or_timed_out'])"
  done
done
cp /tmp/lib_new.so svt-hevc_amd/libsvt_hevc_amd.so
