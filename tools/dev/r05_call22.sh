#!/bin/bash
# round 5, call 22: SQ counters of the final default kernel (gemm_x6rd_kernel, tile 66) beside gemm_x6r_kernel (tile 65), batch 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
i=0
: > $O/r05_x6rd_pmc.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf $O/gp$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/gp$i -o p -- python $R/tools/dev/mb_gemm.py x6r,x6d "" "dec c4,l2.c1 256" 3 > $O/gp$i.log 2>&1 || echo "pass $i failed"
  echo "== pass $i: $C" >> $O/r05_x6rd_pmc.txt
  python $R/tools/dev/pmc_report.py $(find $O/gp$i -name "*.db" | head -1) >> $O/r05_x6rd_pmc.txt 2>&1
  rm -rf $O/gp$i $O/gp$i.log
done
cat $O/r05_x6rd_pmc.txt | grep -v "Functor" | cut -c1-250
