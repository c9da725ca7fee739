#!/bin/bash
# round 6, call 19: attn_x6_d32_kernel with its score / value MFMAs in alternating order (AOT_X6_ALT) against the product
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/aot-benchmark_amd/csrc/libaot_hip_alt.so
{
echo "== product"; timeout 300 python tools/dev/mb_attn_x6.py "" 2>&1 | grep -v amdgpu.ids
echo "== AOT_X6_ALT"; timeout 300 python tools/dev/mb_attn_x6.py $L 2>&1 | grep -v amdgpu.ids
echo "== product again"; timeout 300 python tools/dev/mb_attn_x6.py "" quick 2>&1 | grep -v amdgpu.ids
echo "== AOT_X6_ALT again"; timeout 300 python tools/dev/mb_attn_x6.py $L quick 2>&1 | grep -v amdgpu.ids
AOT_HIP_LIB=$L timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "attention_x6 or attention_kernels_reproducible" 2>&1 | tail -3
} > $O/r06_call19.txt 2>&1
cat $O/r06_call19.txt
