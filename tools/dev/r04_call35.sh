#!/bin/bash
# planes-in / planes-out links with the early DMA issue switch (the combined form round 5 starts from)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=$PWD/aot-benchmark_amd/csrc
{
AOT_HIP_LIB=$L/libaot_hip_early.so timeout 100 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6_presplit" 2>&1 | grep -E "passed|failed|Error|assert" | head -4
for v in _early ""; do
  echo "== lib libaot_hip$v.so"
  timeout 100 python tools/dev/mb_gemm.py x6n,x6p,x6pp,x6 $L/libaot_hip$v.so "" 3 2>&1 | grep -v amdgpu.ids
done
} > $O/r04_x6_chain_early.txt 2>&1
grep -E "==|passed|failed|per-frame" $O/r04_x6_chain_early.txt | cut -c1-170
