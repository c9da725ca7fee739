#!/bin/bash
# round 5, call 15: the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/r05_gpu_suite.txt
cat $O/r05_gpu_suite.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
