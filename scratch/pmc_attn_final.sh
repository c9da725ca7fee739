cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/pmcz1 $O/pmcz2
timeout 60 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/pmcz1 -o p -- python $R/scratch/pmc_attn.py 12 > $O/pmcz1.log 2>&1 || echo "pass1 failed"
cd $R
python scratch/pmc_report.py $(find $O/pmcz1 -name "*.db" | head -1) 2>&1 | grep -E "kernel|attn_" > $O/attn_pmc_final.txt
cat $O/attn_pmc_final.txt
rm -rf $O/pmcz1
