#!/bin/bash
# the bf16x6 GEMM member on PRE-SPLIT activations: kernel tests (bit-identical to the on-the-fly split), time against the shipped
# kernels; the training bf16 kernel (register allocation changed by the new template member) re-checked
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -12
timeout 300 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "bf16" 2>&1 | grep -E "passed|failed|Error|assert" | head -6
for b in 3 1; do
  echo "== gemm, batch $b"
  timeout 300 python tools/dev/mb_gemm.py x6n,x6p,x6ps,x6 "" "" $b 2>&1 | grep -v amdgpu.ids
done
} > $O/r04_x6_presplit.txt 2>&1
cat $O/r04_x6_presplit.txt | cut -c1-150
