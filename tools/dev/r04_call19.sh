#!/bin/bash
# A/B of VALU trims in the bf16x6 kernels: packed residual subtraction in the splits (s), packed exponent fma + inline-zero chain start
# (f), two tiles per loop trip without the score-register copy (u)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=aot-benchmark_amd/csrc
python tools/dev/mb_attn_x6.py "" quick > /dev/null 2>&1      # warm-up, discarded
{
for v in "" _s _sf _su _sfu _u ""; do
  echo "== attention, lib libaot_hip$v.so"
  timeout 200 python tools/dev/mb_attn_x6.py $L/libaot_hip$v.so quick 2>&1 | grep -v amdgpu.ids
done
for b in 3 1; do for v in "" _s ""; do
  echo "== gemm x6, batch $b, lib libaot_hip$v.so"
  timeout 300 python tools/dev/mb_gemm.py x6n,x6w,x6 $L/libaot_hip$v.so "" $b 2>&1 | grep -v amdgpu.ids
done; done
} > $O/r04_pk_variants.txt 2>&1
tail -5 $O/r04_pk_variants.txt
