#!/bin/bash
# fabric traffic of the bf16x6 attention kernel over the launch mix of one 70-frame clip (two PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pm_$c
  timeout 150 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py aotx6 > $O/r03z_pm_x6_$c.log 2>&1 || echo "pass $c failed"
done
python $R/tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/r03z_attn_x6_traffic.json attn_x6_d32_kernel | tail -12
rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE
