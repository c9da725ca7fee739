#!/bin/bash
# the 64x64 bf16x6 kernels with the DMA of step ss+3 issued in step ss (the slot of step ss is dead once its fragments are in
# registers): two steps of DMA in flight on the same three ring slots.  Kernel tests (both activation forms, the training bf16 form),
# then the frame's GEMM set against the shipped build.
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=$PWD/aot-benchmark_amd/csrc
{
echo "== product library: the pre-split member's tests (relaxed to the tolerance)"
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -6
echo "== early-issue variant: kernel tests"
AOT_HIP_LIB=$L/libaot_hip_early.so timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -6
AOT_HIP_LIB=$L/libaot_hip_early.so timeout 300 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "bf16" 2>&1 | grep -E "passed|failed|Error|assert" | head -6
for b in 3 1; do for v in "" _early ""; do
  echo "== gemm, batch $b, lib libaot_hip$v.so"
  timeout 300 python tools/dev/mb_gemm.py x6n,x6p,x6 $L/libaot_hip$v.so "" $b 2>&1 | grep -v amdgpu.ids
done; done
} > $O/r04_x6_early_issue.txt 2>&1
grep -E "==|passed|failed|per-frame" $O/r04_x6_early_issue.txt
