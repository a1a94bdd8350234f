cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_final; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
bash tools/profile_bench.sh r04_final_prof > $O/profile_bench.log 2>&1
cp gpurun_out/r04_final_prof/kernel_stats.txt $O/kernel_stats.txt 2>/dev/null
bash tools/profile_me_pmc.sh r04_final_me_pmc 8 > $O/me_pmc.log 2>&1
cp gpurun_out/r04_final_me_pmc/me_pmc.txt $O/me_pmc.txt 2>/dev/null
( timeout 100 python tools/me_phase_profile.py 16 b_3840x2160_m7; ME_EXP_STAMPS=1 SVT_PRODUCT_LIB=tools/_exp/lib10.so timeout 100 python tools/me_phase_profile.py 16 b_3840x2160_m7 ) > $O/me_phase.txt 2>&1
bash tools/profile_md_bench.sh r04_final_md_prof > $O/profile_md_bench.log 2>&1
cp gpurun_out/r04_final_md_prof/md_kernel_stats.txt $O/md_kernel_stats.txt 2>/dev/null
MD_PMC_ONLY="1 2 6 7" bash tools/profile_md_pmc.sh r04_final_md_pmc > $O/md_pmc.log 2>&1
cp gpurun_out/r04_final_md_pmc/md_pmc.txt $O/md_pmc.txt 2>/dev/null
python -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
ls -la $O
