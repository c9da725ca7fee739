#!/bin/bash
# the 64x64 bf16x6 GEMM kernel: order of the splits and the MFMAs inside a k-step (AOT_X6_MODE 1: both splits first, chains alternating; 2: the next step's A planes split between the MFMA halves)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=$PWD/aot-benchmark_amd/csrc
python tools/dev/mb_gemm.py x6n "" "l3.c1" 3 > /dev/null 2>&1      # warm-up, discarded
{
for v in _m1 _m2; do
  echo "== kernel tests, lib libaot_hip$v.so"
  AOT_HIP_LIB=$L/libaot_hip$v.so timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
done
for b in 3 1; do for v in "" _m1 _m2 ""; do
  echo "== gemm x6n, batch $b, lib libaot_hip$v.so"
  timeout 300 python tools/dev/mb_gemm.py x6n,x6 $L/libaot_hip$v.so "" $b 2>&1 | grep -v amdgpu.ids
done; done
} > $O/r04_x6_step_modes.txt 2>&1
grep -E "==|passed|failed|per-frame" $O/r04_x6_step_modes.txt
