#!/bin/bash
# round 5, call 2: what is a k-step of the 128x128 bf16x6 kernel made of?  Timing probes of gemm_x6pp_kernel (AOT_PP_PROBE; the
# results of the probe builds are WRONG by construction): 1 = no A DMA, 2 = no B DMA, 4 = no split, 8 = no fragment reads, 16 = no MFMAs
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
echo "== base"
timeout 200 python tools/dev/mb_gemm.py x6w,x6z "" "dec c4,dec c8,l2.c1 256,l1.c3" 3 2>&1 | grep -v "amdgpu.ids\|differs"
for v in 1 2 3 7 15 16 19 27; do
  echo "== probe $v"
  timeout 200 python tools/dev/mb_gemm.py x6z $C/libaot_hip_probe$v.so "dec c4,dec c8,l2.c1 256,l1.c3" 3 2>&1 | grep -v "amdgpu.ids\|differs\|^shape"
done
} > $O/r05_x6pp_probes.txt 2>&1
cat $O/r05_x6pp_probes.txt | cut -c1-120
