#!/bin/bash
# round 5, call 18: s_setprio around the MFMA groups of the bf16x6 attention kernel (AOT_ATTN_X6_PRIO = 1: every group, 2: the value product)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
for v in "" $C/libaot_hip_aprio1.so $C/libaot_hip_aprio2.so ""; do
  echo "== ${v:-shipped}"; timeout 200 python tools/dev/mb_attn_x6.py "$v" quick 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-120
done
} > $O/r05_attn_prio.txt 2>&1
cat $O/r05_attn_prio.txt
