# Run on the GPU box: the hooked encoder at BASELINE configs[2] with SVT_HOOK_MD=pb and SVT_AMD_MD_TIMING=1 (where a device-decided picture's time goes: inputs up, kernel, records down;
# the binding's own split in the report), then the PMC passes of the mode-decision kernel (tools/profile_md_pmc.sh).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-md_timing}; mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "tests")
import svtlib as S
S.write_clip("/tmp/md_clip.yuv", "motion", 3840, 2160, 16, 7)
PY
SVT_HOOK_MD=${MD_MODE:-1} SVT_AMD_MD_TIMING=1 SVT_HOOK_REPORT=$O/report.txt integration/_build/SvtHevcEncApp_hip -i /tmp/md_clip.yuv -w 3840 -h 2160 -n 64 -nb 16 -b /tmp/md.265 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -lp 32 > $O/app.txt 2> $O/stderr.txt < /dev/null
grep "Average Speed" $O/app.txt
grep "mode decision\|picture objects" $O/report.txt | cut -c1-420
grep "svt_amd_md_encode_picture" $O/stderr.txt | awk '{u+=$6; k+=$10; d+=$14; n++} END {printf "calls %d: inputs up %.2f ms, kernel %.2f ms, records down %.2f ms (means)\n", n, u/n, k/n, d/n}'
grep "svt_amd_md_encode_picture" $O/stderr.txt | head -3
if [ -n "$MD_PMC" ]; then bash tools/profile_md_pmc.sh r03_ak_pmc > $O/pmc_log.txt 2>&1; tail -30 $O/pmc_log.txt; fi
