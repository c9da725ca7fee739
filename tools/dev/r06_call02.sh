#!/bin/bash
# round 6, call 2: the software-pipelined 64-query gated kernel (attn_x6_wide64p_kernel): tests, launch times (NVB 4 / 2, first form)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "gated_attention_x6 or reproducible_under_load" 2>&1 | tail -5
echo "== product library (pipelined, NVB 4)"
timeout 600 python tools/dev/mb_gated_x6.py "" quick
for v in pnvb2; do
  echo "== variant $v"
  timeout 300 python tools/dev/mb_gated_x6.py aot-benchmark_amd/csrc/libaot_hip_$v.so quick 2>&1 | grep -v "^pack" | grep "ns= 9\|ns= 8\|ns= 1 "
done
} > $O/r06_gated64p2.txt 2>&1
cat $O/r06_gated64p2.txt
