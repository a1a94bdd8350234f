# GPU box: one iteration of the mode-decision kernel work: chain profile of the 4K "motion" fixture + decision check on the 4K "objects" fixture (both B layers)
# usage: bash tools/md_iter.sh <tag> [lib]
cd ${GRAFT_REPO_ROOT:-.}
[ -n "$2" ] && export SVT_PRODUCT_LIB=$2
timeout 300 python tools/md_chain2.py 0,1 > gpurun_out/$1.json 2> gpurun_out/$1.err
for k in 0 1; do timeout 120 python tools/md_diff.py tools/_fx/md4k_objects.npz $k 2>&1 | tail -4; done > gpurun_out/$1.diff
tail -2 gpurun_out/$1.err; grep -c "differing leaves: 0 of" gpurun_out/$1.diff
