#!/bin/bash
# round 5, call 10: grid-level key splits of the bf16x6 attention kernel beyond 4 (three resident waves per SIMD: 1696 waves x ns on 1024 SIMDs)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python tools/dev/mb_attn_x6.py "" sweep2 2>&1 | grep -v amdgpu.ids > $O/r05_attn_x6_splits.txt
cut -c1-110 $O/r05_attn_x6_splits.txt
