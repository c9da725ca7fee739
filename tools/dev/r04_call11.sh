#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
d=bisect
timeout 300 tools/dev/x6_hazard_mod tools/dev/$d/KERNEL tools/dev/$d/e0_*.co tools/dev/$d/s_*.co > $O/r04_isa_pk_$d.txt 2>&1; echo "$d rc $?"; grep -v "^        split\|^      launch" $O/r04_isa_pk_$d.txt | cut -c1-160
