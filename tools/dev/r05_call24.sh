#!/bin/bash
# round 5, call 24: gemm_x6rd_kernel built for FOUR workgroups per CU (128-register cap, 4-14 registers spilled) against three
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
for b in 3 1; do
  for v in "" $C/libaot_hip_occ4.so ""; do
    echo "== batch $b ${v:-shipped}"; timeout 300 python tools/dev/mb_gemm.py x6d "$v" "" $b 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
timeout 300 python tools/dev/mb_gemm.py x6d $C/libaot_hip_occ4.so "" 3 2>&1 | grep -v amdgpu.ids
} > $O/r05_x6rd_occ4.txt 2>&1
cat $O/r05_x6rd_occ4.txt | cut -c1-110
