#!/bin/bash
# round 5, call 6: the register-staged kernel generalised: 64x64 (tile 65) and 128x128 on eight waves (tile 129): kernel tests
# (bit-identical to the LDS-DMA kernels of the same tile), time per shape at batch 3 / 1
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "phase_shifted or (conv2d_bf16x6_kernel and (129 or 65))" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -12
for b in 3 1; do
  echo "== gemm, batch $b: round-5 dispatch / register-staged 64x64 / LDS-DMA 128x128 / register-staged 128x128"
  timeout 300 python tools/dev/mb_gemm.py x6,x6r,x6w,x6s "" "" $b 2>&1 | grep -v amdgpu.ids
done
} > $O/r05_x6r128.txt 2>&1
cat $O/r05_x6r128.txt | cut -c1-150
