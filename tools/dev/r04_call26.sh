#!/bin/bash
# fabric traffic of DeAOT's bf16x6 gated attention kernel over the bench's launch mix (two PMC passes) -> roofline.traffic of the
# DeAOT sub-runs in the default (bf16x6) bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pm_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py gatedx6 > $O/r04_pm_gatedx6_$c.log 2>&1 || echo "pass $c failed"
done
python $R/tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/r04_gated_attn_x6_traffic.json attn_x6_wide_coop_kernel | tail -12
rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE
