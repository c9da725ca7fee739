#!/bin/bash
# round 6, call 1: the 64-query bf16x6 gated attention kernel (attn_x6_wide64_kernel): kernel tests, then launch times against the
# 32-query kernel over bank sizes and key splits, prefetch depth NVB 2 / 3 / 4 / 5 and the legacy dispatch order (variant libraries)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "gated_attention_x6 or reproducible_under_load" 2>&1 | tail -5
echo "== product library (NVB 3, XCD-major order)"
timeout 600 python tools/dev/mb_gated_x6.py "" 
for v in nvb2 nvb4 nvb5 lin; do
  echo "== variant $v"
  timeout 300 python tools/dev/mb_gated_x6.py aot-benchmark_amd/csrc/libaot_hip_$v.so quick 2>&1 | grep -v "^pack"
done
} > $O/r06_gated64.txt 2>&1
cat $O/r06_gated64.txt
