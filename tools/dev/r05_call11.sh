#!/bin/bash
# round 5, call 11: the 64x64 register-staged kernel with the weight fragments straight from global memory (gemm_x6rd_kernel, tile 66)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "phase_shifted or (conv2d_bf16x6_kernel and 66)" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -12
for b in 3 1; do
  echo "== gemm, batch $b: round-5 dispatch / register-staged 64x64 / + direct weights"
  timeout 300 python tools/dev/mb_gemm.py x6,x6r,x6d "" "" $b 2>&1 | grep -v amdgpu.ids
done
} > $O/r05_x6rd.txt 2>&1
cat $O/r05_x6rd.txt | cut -c1-130
