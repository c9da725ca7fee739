#!/bin/bash
# pre-split member with the reads / DMA pieces interleaved between the MFMAs (+ early DMA issue): tests, then the frame's GEMM set
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=$PWD/aot-benchmark_amd/csrc
{
AOT_HIP_LIB=$L/libaot_hip_ilv.so timeout 60 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6_presplit" 2>&1 | grep -E "passed|failed|Error|assert" | head -4
echo "== lib libaot_hip_ilv.so"
timeout 60 python tools/dev/mb_gemm.py x6n,x6p,x6pp $L/libaot_hip_ilv.so "" 3 2>&1 | grep -v amdgpu.ids
} > $O/r04_x6_interleaved.txt 2>&1
grep -E "==|passed|failed|per-frame|l3.c2 3x3 256|dec c4|lstt 256>256" $O/r04_x6_interleaved.txt | cut -c1-150
