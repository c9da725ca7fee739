#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 tools/dev/x6_hazard 12 > $O/r04_x6_hazard_d.txt 2>&1; echo "x6_hazard rc $?"; cut -c1-230 $O/r04_x6_hazard_d.txt | head -150
