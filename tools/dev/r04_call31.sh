#!/bin/bash
# SQ / LDS counters of the 64x64 bf16x6 GEMM kernel on two shapes: one with a lone workgroup per CU (l3.c2 3x3 256 at batch 1: 108
# tiles) and one that saturates the chip (dec c4 3x3 128 at batch 3) -- what bounds its k-step?  (own passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
: > $O/r04_x6_gemm_pmc.txt
for pass in A B; do
  C=$A; [ $pass = B ] && C=$B
  for b in 1 3; do
    rm -rf $O/pm_g
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/pm_g -o p -- python $R/tools/dev/mb_gemm.py x6n,-2 "" "l3.c2 3x3 256,dec c4,lstt 256>256" $b > $O/r04_pm_gemm_$pass$b.log 2>&1 || echo "pass $pass batch $b failed"
    echo "== pass $pass (batch $b): $C" >> $O/r04_x6_gemm_pmc.txt
    python $R/tools/dev/pmc_report.py $(find $O/pm_g -name "*.db" | head -1) 2>&1 | grep -v "pack\|splitk" >> $O/r04_x6_gemm_pmc.txt
    rm -rf $O/pm_g
  done
done
cut -c1-220 $O/r04_x6_gemm_pmc.txt
