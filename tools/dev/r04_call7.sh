#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 tools/dev/x6_hazard 6 > $O/r04_x6_hazard_e.txt 2>&1; echo "x6_hazard rc $?"; grep -B1 -A5 "is wrong" $O/r04_x6_hazard_e.txt | cut -c1-420 | head -150; grep "DBG" $O/r04_x6_hazard_e.txt | grep differing | cut -c1-150
