#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 tools/dev/x6_hazard 20 > $O/r04_x6_hazard_f.txt 2>&1; echo "x6_hazard rc $?"; grep "differing" $O/r04_x6_hazard_f.txt | cut -c1-150
