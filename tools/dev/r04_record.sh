#!/bin/bash
# round 4 record pass: fabric traffic of the attention kernels as built (two PMC passes each), SQ / LDS counters of the windowed
# attention kernels, kernel stats of the bench line and of the training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
traffic() {   # mode, kernel substring, output name
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pm_$c
    timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py $1 > $O/r04_pm_$1_$c.log 2>&1 || echo "pass $1 $c failed"
  done
  python $R/tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/$3 $2 | tail -4
  rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE
}
traffic gated attn_fwd_wide_coop_kernel r04_gated_attn_traffic.json
traffic aotx6 attn_x6_d32_kernel r04_attn_x6_traffic.json
traffic aot attn_fwd_d32_pipe_kernel r04_attn_traffic.json
# windowed attention kernels: SQ counters, then LDS counters (own passes)
rm -rf $O/pm_loc
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pm_loc -o p -- python $R/tools/dev/pmc_local.py > $O/r04_pm_local_sq.log 2>&1 || echo "local SQ pass failed"
python $R/tools/dev/pmc_report.py $(find $O/pm_loc -name "*.db" | head -1) > $O/r04_local_pmc.txt 2>&1; cut -c1-230 $O/r04_local_pmc.txt
rm -rf $O/pm_loc
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS -d $O/pm_loc -o p -- python $R/tools/dev/pmc_local.py > $O/r04_pm_local_lds.log 2>&1 || echo "local LDS pass failed"
python $R/tools/dev/pmc_report.py $(find $O/pm_loc -name "*.db" | head -1) > $O/r04_local_pmc_lds.txt 2>&1; cut -c1-230 $O/r04_local_pmc_lds.txt
rm -rf $O/pm_loc
# kernel stats: the bench line (default arithmetic) and the training step
rm -rf $O/prof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python $R/bench.py --gpus 1 --steps 207 --warmup 5 --no-other-configs --no-cpu-baseline --no-x6 --no-jf --repeats 1 > $O/r04_bench_prof.json 2> $O/r04_bench_prof.err
python $R/tools/dev/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/r04_bench_kernel_stats.txt | head -32
rm -rf $O/prof
for prec in f32 bf16; do
  WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 3 > $O/r04_train_prof_$prec.json 2> $O/r04_train_prof_$prec.err
  python $R/tools/dev/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/r04_train_step_kernel_stats_$prec.txt | head -42; tail -n 1 $O/r04_train_prof_$prec.json | cut -c1-300
  rm -rf $O/prof
done
