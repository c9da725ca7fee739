cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
L="7:197:1,22:197:1,12:197:1,7:197:3,12:24:1,17:18:1"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf $O/gq$i
  timeout 120 rocprofv3 --kernel-trace --pmc $C -d $O/gq$i -o p -- $R/tools/dev/gemm_check list $L > $O/gq$i.log 2>&1 || echo "pass $i failed"
  python $R/tools/dev/pmc_report.py $(find $O/gq$i -name "*.db" | head -1) > $O/gemm_pmc2_pass$i.txt 2>&1
  rm -rf $O/gq$i
  cat $O/gemm_pmc2_pass$i.txt | cut -c1-330
done
