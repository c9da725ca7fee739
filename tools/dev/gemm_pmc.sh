# PMC passes over a few GEMM/conv launches (tools/dev/gemm_check list ...): where do the LDS-tiled kernels lose their time?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
# shape 7 = l2.c2 3x3 128 (M 6527, 204 tiles of 64x64), 22 = dec c4 3x3 128 (M 25773, 806 tiles), 13 = l3.c2 3x3 256 (M 1674)
L="7:117:1,7:4:1,7:14:1,7:133:1,22:117:1,22:4:1,13:24:1,13:117:1"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_LOAD_WAVEFRONTS_sum"; do
  i=$((i+1)); rm -rf $O/gp$i
  timeout 120 rocprofv3 --kernel-trace --pmc $C -d $O/gp$i -o p -- $R/tools/dev/gemm_check list $L > $O/gp$i.log 2>&1 || echo "pass $i failed"
  python $R/tools/dev/pmc_report.py $(find $O/gp$i -name "*.db" | head -1) > $O/gemm_pmc_pass$i.txt 2>&1
  rm -rf $O/gp$i
  cat $O/gemm_pmc_pass$i.txt | cut -c1-330
done
