#!/bin/bash
# round 6, call 3: counters of the pipelined 64-query gated kernel beside the 32-query one at a 14-frame bank (SQ, cache, fabric bytes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
i=0
: > $O/r06_gated64_pmc.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $O/gp$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/gp$i -o p -- python $R/tools/dev/pmc_gated_x6.py 14 > $O/gp$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/gp$i.log; }
  echo "== pass $i: $C" >> $O/r06_gated64_pmc.txt
  python $R/tools/dev/pmc_report.py $(find $O/gp$i -name "*.db" | head -1) >> $O/r06_gated64_pmc.txt 2>&1
  rm -rf $O/gp$i $O/gp$i.log
done
cat $O/r06_gated64_pmc.txt | cut -c1-260
