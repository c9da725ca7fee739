#!/bin/bash
# the pre-split member's plane-writing tile end: kernel tests (planes == split of the fp32 result, bit for bit), then planes-in /
# planes-out links against the shipped kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "conv2d_bf16x6_presplit" 2>&1 | grep -E "passed|failed|Error|assert|differs|wrote" | head -8
timeout 200 python tools/dev/mb_gemm.py x6n,x6p,x6pp "" "" 3 2>&1 | grep -v amdgpu.ids
} > $O/r04_x6_plane_output.txt 2>&1
grep -E "passed|failed|Error|assert|per-frame|dec c4|l3.c2 3x3 256|lstt 256>256" $O/r04_x6_plane_output.txt | cut -c1-150
