#!/bin/bash
# round 5, call 12: split-K on the 64x64 direct-weight kernel (x6kN) against the phase-shifted 128x128 split-K kernel (x6zN) and the
# unsplit kernels on the long-K shapes
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "phase_shifted" 2>&1 | grep -E "passed|failed|Error|assert|differs" | head -12
for b in 3 1; do
  echo "== batch $b"
  timeout 300 python tools/dev/mb_gemm.py x6,x6d,x6k2,x6k3,x6k4,x6k6,x6k8,x6k9 "" "l2.c2,l3.c2,l3.c1 1024,lstt 1024,lstt 512,dec c8,dec ad8,l2.c1 512" $b 2>&1 | grep -v amdgpu.ids
done
} > $O/r05_x6rd_splitk.txt 2>&1
cat $O/r05_x6rd_splitk.txt | cut -c1-190
