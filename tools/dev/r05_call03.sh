#!/bin/bash
# round 5, call 3: the register-staged 64x64 bf16x6 kernel (gemm_x6r_kernel, tile = 65): kernel tests (bit-identical to the LDS-DMA
# 64x64 kernel), time against the shipped kernels at batch 3 / 1; grid sized for 2 instead of 3 workgroups per CU
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "phase_shifted or (conv2d_bf16x6_kernel and 65)" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -12
for b in 3 1; do
  echo "== gemm, batch $b: shipped dispatch / 64x64 LDS-DMA / 64x64 register-staged / 128x128"
  timeout 300 python tools/dev/mb_gemm.py x6,x6n,x6r,x6w "" "" $b 2>&1 | grep -v amdgpu.ids
done
echo "== grid sized for 2 workgroups per CU, batch 3"
timeout 300 python tools/dev/mb_gemm.py x6n,x6r $C/libaot_hip_r2.so "" 3 2>&1 | grep -v amdgpu.ids
} > $O/r05_x6r.txt 2>&1
cat $O/r05_x6r.txt | cut -c1-150
