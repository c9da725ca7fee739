#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "gn_partials or c4_bf16x6 or merged_qkv or graph_replay or (bf16x6 and c2_r50_aotl_70 and throughput) or multi_group or end_to_end_vs_reference_golden or attention_kernels_reproducible or local_ or swin or lane_batched" 2>&1 | tail -8
