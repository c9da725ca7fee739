cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/pmc_f $O/pmc_w
timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o p -- python $R/scratch/pmc_attn_mix.py > $O/pmc_f.log 2>&1 || echo "fetch pass failed/timeout"
timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o p -- python $R/scratch/pmc_attn_mix.py > $O/pmc_w.log 2>&1 || echo "write pass failed/timeout"
cd $R
python scratch/attn_traffic.py $(find $O/pmc_f -name "*.db" | head -1) $(find $O/pmc_w -name "*.db" | head -1) $O/attn_traffic.json | tail -14
rm -rf $O/pmc_f $O/pmc_w
