#!/bin/bash
# A/B of the schedule / lazy-rescale variants of attn_x6_d32_kernel (variant libraries built by tools/dev/build_variant.sh)
mkdir -p gpurun_out
: > gpurun_out/r03n_attn_x6_variants.txt
for v in xm xn xo xp xq; do
  echo "== $v" >> gpurun_out/r03n_attn_x6_variants.txt
  timeout 120 python tools/dev/mb_attn_x6.py aot-benchmark_amd/csrc/libaot_hip_$v.so quick 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03n_attn_x6_variants.txt
done
cat gpurun_out/r03n_attn_x6_variants.txt
