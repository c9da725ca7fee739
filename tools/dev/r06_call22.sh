#!/bin/bash
# round 6, call 22: lgp_aggregate_kernel chunk 8 against 16 (product)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/aot-benchmark_amd/csrc/libaot_hip_cb8.so
{
for i in 1 2; do
echo "== CB 16 (product)"; python tools/dev/mb_local_gated.py "" 2>&1 | grep -v amdgpu.ids
echo "== CB 8"; python tools/dev/mb_local_gated.py $L 2>&1 | grep -v amdgpu.ids
done
} > $O/r06_call22.txt 2>&1
cat $O/r06_call22.txt
