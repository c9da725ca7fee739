# First GPU-box call of round 3 (about 12 GPU-minutes):  gpurun --timeout 900 -- 'bash tools/dev/r03_first.sh'
# Before it, here (no GPU needed):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dev/gemm_check.hip aot-benchmark_amd/csrc/gemm_conv.hip \
#         aot-benchmark_amd/csrc/gemm_lds.hip -o tools/dev/gemm_check
#   (the same with -DAOT_LEAN_EPI=1 -DAOT_CONV_EPI=1 -o tools/dev/gemm_check_epi)
#   bash tools/dev/build_variant.sh epi1 -DAOT_LEAN_EPI=1; bash tools/dev/build_variant.sh pksum1 -DAOT_ATT_PKSUM=1
#   bash tools/dev/build_variant.sh coop1 -DAOT_COOP_AGPR=1
#   bash tools/dev/build_variant.sh all1 -DAOT_LEAN_EPI=1 -DAOT_CONV_EPI=1 -DAOT_ATT_PKSUM=1 -DAOT_COOP_AGPR=1
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
# 1. the one GPU test that has never run (two-cohort decode path); --runxfail makes a failure visible
timeout 120 python -m pytest tests/test_training_gpu.py -q -s -m gpu -p no:cacheprovider --runxfail -k new_object_group > $O/r03_newgroup.log 2>&1
echo "newgroup rc=$? $(tail -1 $O/r03_newgroup.log)"
# 2. self-checking GEMM harness: shipped kernels + gemm_direct3 (cfg 32/34/38), then the same with the lean tile ends
timeout 300 tools/dev/gemm_check quick > $O/r03_gemm_check.txt 2>&1; echo "gemm_check rc=$?"; grep -c "ERR" $O/r03_gemm_check.txt; tail -3 $O/r03_gemm_check.txt | cut -c1-400
timeout 300 tools/dev/gemm_check_epi quick > $O/r03_gemm_check_epi.txt 2>&1; echo "gemm_check_epi rc=$?"; grep -c "ERR" $O/r03_gemm_check_epi.txt; tail -3 $O/r03_gemm_check_epi.txt | cut -c1-400
# 3. library variants end to end
bash tools/dev/r03_ab.sh base epi1 pksum1 coop1 all1
