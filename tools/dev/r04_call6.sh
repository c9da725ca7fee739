#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for d in bisect bisect8 bisect256; do
  timeout 200 tools/dev/x6_hazard_mod tools/dev/$d/KERNEL tools/dev/$d/e0_*.co tools/dev/$d/e1[2-7]_*.co > $O/r04_isa_$d.txt 2>&1; echo "$d rc $?"; grep -v "^        split\|^      launch" $O/r04_isa_$d.txt | cut -c1-160
done
