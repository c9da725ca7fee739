#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/dev/mb_attn_x6.py "" sweep > gpurun_out/r03l_mb_attn_x6.txt 2>&1
tail -30 gpurun_out/r03l_mb_attn_x6.txt
