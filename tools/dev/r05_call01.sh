#!/bin/bash
# round 5, call 1: the phase-shifted 128x128 bf16x6 GEMM (gemm_x6pp_kernel): kernel tests, time against the shipped kernels (and its
# three build variants: DMA issue in the MFMA phase, s_setprio around the MFMA phase, both), split-K on the stride-16 shapes;
# then the bf16x6 cells of the parity matrix
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "phase_shifted or (conv2d_bf16x6_kernel and 256)" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -12
for b in 3 1; do
  echo "== gemm, batch $b: shipped dispatch / 64x64 / 128x128 / 128x128 phase-shifted"
  timeout 300 python tools/dev/mb_gemm.py x6,x6n,x6w,x6z "" "" $b 2>&1 | grep -v amdgpu.ids
done
for v in ppdmac ppprio ppboth; do
  echo "== variant $v, batch 3: 128x128 / phase-shifted"
  timeout 300 python tools/dev/mb_gemm.py x6w,x6z $C/libaot_hip_$v.so "l1.c2,l1.c3,l2.c1 256,l2.c2 3x3 128,l2.ds,l3.c1 512,l3.ds,dec" 3 2>&1 | grep -v amdgpu.ids
done
for b in 3 1; do
  echo "== split-K of the phase-shifted kernel on the stride-16 / stride-8 shapes, batch $b"
  timeout 300 python tools/dev/mb_gemm.py x6,x6z,x6z2,x6z3,x6z4,x6z8 "" "l2.c2,l2.c1 512,l3,lstt,dec ad8,dec c8" $b 2>&1 | grep -v amdgpu.ids
done
} > $O/r05_x6pp.txt 2>&1
cat $O/r05_x6pp.txt | cut -c1-170
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "bf16x6 and (free_running or full_size or engine_vs_reference)" 2>&1 | tail -15 > $O/r05_parity_cells.txt
cat $O/r05_parity_cells.txt
