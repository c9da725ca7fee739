#!/bin/bash
# round 5, call 7: SQ / TA counters of the four bf16x6 GEMM kernels on dec c4 3x3 128 and l2.c1 256>128 at batch 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
i=0
: > $O/r05_x6_gemm_pmc.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf $O/gp$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/gp$i -o p -- python $R/tools/dev/mb_gemm.py x6n,x6r,x6w,x6s "" "dec c4,l2.c1 256" 3 > $O/gp$i.log 2>&1 || echo "pass $i failed"
  echo "== pass $i: $C" >> $O/r05_x6_gemm_pmc.txt
  python $R/tools/dev/pmc_report.py $(find $O/gp$i -name "*.db" | head -1) >> $O/r05_x6_gemm_pmc.txt 2>&1
  rm -rf $O/gp$i $O/gp$i.log
done
cat $O/r05_x6_gemm_pmc.txt | cut -c1-250
