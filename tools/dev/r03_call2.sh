# GPU call 2 of round 3:  gpurun --timeout 1700 -- 'bash tools/dev/r03_call2.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; L=$R/aot-benchmark_amd/csrc; cd $R
rm -f $O/parity_r03.json
timeout 200 python bench.py --no-cpu-baseline --no-jf --no-roofline --steps 20 > /dev/null 2>&1      # warm-up, discarded
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/r03b_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/r03b_pytest.log)"; grep -E "^(FAILED|ERROR)" $O/r03b_pytest.log | head -20
timeout 400 python bench.py --steps 20 --warmup 5 > $O/r03b_bench20.json 2> $O/r03b_bench20.err; echo "bench20 rc=$?"; cut -c1-400 $O/r03b_bench20.json
timeout 400 python bench.py > $O/r03b_bench207.json 2> $O/r03b_bench207.err; echo "bench207 rc=$?"; python -c "import json; d=json.load(open('$O/r03b_bench207.json')); print(d['value'], d['config']['repeat_fps'], d['config']['single_stream'], d['roofline']['frac'], d['config']['jf_vs_reference']['pixels_outside_near_ties'])"
for m in swinb_deaotl r50_deaotl; do
  timeout 500 python bench.py --model $m > $O/r03b_bench_$m.json 2> $O/r03b_bench_$m.err; echo "$m rc=$?"
  python -c "import json; d=json.load(open('$O/r03b_bench_$m.json')); print('$m', d['value'], d['config']['single_stream'], d['roofline'], d['config']['jf_vs_reference'], d['cpu_baseline'])" | cut -c1-1200
done
# kernel stats: one clip at a time, then the default three clips
cd /tmp
for s in 1 3; do
  rm -rf $O/prof_s$s
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_s$s -o p -- python $R/bench.py --streams $s --steps 138 --repeats 1 --no-cpu-baseline --no-jf --no-roofline > $O/r03b_prof_s$s.log 2>&1
  python $R/tools/dev/prof_summary.py $(find $O/prof_s$s -name "*.db" | head -1) $O/r03b_kernel_stats_s$s.txt | head -24 | cut -c1-130
  rm -rf $O/prof_s$s
done
# attention: fabric traffic (two passes each), dispatch order variants, then the SQ counters of both kernels at M = 14
for v in base order1; do
  lib=$L/libaot_hip.so; [ $v = order1 ] && lib=$L/libaot_hip_order1.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pm_$c
    AOT_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py aot > $O/r03b_pm_${v}_$c.log 2>&1 || echo "pass $v $c failed"
  done
  python $R/tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/r03b_attn_traffic_$v.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/r03b_attn_traffic_$v.json')); print('$v traffic/launch', d['traffic_bytes_per_launch'], d['bytes_per_launch'])"
  rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pm_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py gated > $O/r03b_pm_gated_$c.log 2>&1 || echo "gated pass $c failed"
done
python $R/tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/r03b_gated_attn_traffic.json attn_fwd_wide_coop_kernel > /dev/null 2>&1
python -c "import json; d=json.load(open('$O/r03b_gated_attn_traffic.json')); print('gated traffic/launch', d['traffic_bytes_per_launch'], d['bytes_per_launch'])"
rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE $O/pm_sq
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pm_sq -o p -- python $R/tools/dev/pmc_attn_mix.py m14 > $O/r03b_pm_sq.log 2>&1
python $R/tools/dev/pmc_report.py $(find $O/pm_sq -name "*.db" | head -1) > $O/r03b_attn_pmc.txt 2>&1; cat $O/r03b_attn_pmc.txt | cut -c1-260
rm -rf $O/pm_sq
