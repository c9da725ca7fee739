#!/bin/bash
# round 5, call 23: gemm_x6rd_kernel with the next step's weight fragments requested at the head of the step (AOT_X6RD_BFIRST)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
for b in 3 1; do
  for v in "" $C/libaot_hip_bfirst.so ""; do
    echo "== batch $b ${v:-shipped}"; timeout 300 python tools/dev/mb_gemm.py x6d "$v" "" $b 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
timeout 300 python tools/dev/mb_gemm.py x6d $C/libaot_hip_bfirst.so "" 3 2>&1 | grep -v amdgpu.ids
} > $O/r05_x6rd_bfirst.txt 2>&1
cat $O/r05_x6rd_bfirst.txt | cut -c1-110
