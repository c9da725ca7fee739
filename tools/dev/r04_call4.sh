#!/bin/bash
# round 4, call 4: dependent-MFMA gap probe (with / without loads landing), back-to-back / spread variants of the reproducer
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 200 tools/dev/mfma_hazard_probe dep > $O/r04_mfma_dep_probe.txt 2>&1; echo "dep probe rc $?"; cut -c1-250 $O/r04_mfma_dep_probe.txt | grep -v "q0 0 q1 0 q2 0 q3 0 | even 0 odd 0\]   2/SIMD: bad regs \[q0 0 q1 0 q2 0 q3 0 | even 0 odd 0\]   4/SIMD: bad regs \[q0 0 q1 0 q2 0 q3 0" | head -40
timeout 300 tools/dev/x6_hazard 30 > $O/r04_x6_hazard_c.txt 2>&1; echo "x6_hazard rc $?"; cut -c1-200 $O/r04_x6_hazard_c.txt
