# Record pass of round 3:  gpurun --timeout 1800 -- 'bash tools/dev/r03_record.sh'
# whole GPU suite, the bench lines (driver form, whole clips), kernel stats of the default bench and of the bf16x6 family,
# SQ counters of the four flash-attention kernels at M = 14
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/parity_r03.json
timeout 200 python bench.py --no-cpu-baseline --no-jf --no-roofline --no-x6 --steps 20 > /dev/null 2>&1      # warm-up, discarded
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > $O/r03z_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/r03z_pytest.log)"; grep -E "^(FAILED|ERROR)" $O/r03z_pytest.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py --steps 20 --warmup 5 > $O/r03z_bench20.json 2> $O/r03z_bench20.err; echo "bench20 rc=$?"; tail -2 $O/r03z_bench20.err
python -c "import json; d=json.load(open('$O/r03z_bench20.json')); c=d['config']; print('bench20', d['value'], c['repeat_fps'], c['single_stream']['fps'], d['roofline']['frac'], c['jf_vs_reference']['pixels_outside_near_ties'], c['bf16x6_split']['value'], c['bf16x6_split']['jf_vs_reference']['pixels_outside_near_ties'], d['cpu_baseline']['value'])"
timeout 500 python bench.py > $O/r03z_bench207.json 2> $O/r03z_bench207.err; echo "bench207 rc=$?"
python -c "import json; d=json.load(open('$O/r03z_bench207.json')); c=d['config']; print('bench207', d['value'], c['repeat_fps'], c['single_stream']['fps'], d['roofline']['frac'], c['jf_vs_reference']['pixels_outside_near_ties'], c['bf16x6_split']['value'], c['bf16x6_split']['jf_vs_reference']['pixels_outside_near_ties'])"
cd /tmp; rm -rf $O/prof_z
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_z -o p -- python $R/bench.py --steps 138 --repeats 1 --no-cpu-baseline --no-jf --no-roofline --no-x6 > $O/r03z_prof.log 2>&1
python $R/tools/dev/prof_summary.py $(find $O/prof_z -name "*.db" | head -1) $O/r03z_bench_kernel_stats.txt | head -14 | cut -c1-130
rm -rf $O/prof_z
cd /tmp; rm -rf $O/prof_z
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_z -o p -- python $R/bench.py --steps 138 --repeats 1 --mfma bf16x6 --no-cpu-baseline --no-jf --no-roofline > $O/r03z_prof_x6.log 2>&1
python $R/tools/dev/prof_summary.py $(find $O/prof_z -name "*.db" | head -1) $O/r03z_bench_kernel_stats_bf16x6.txt | head -12 | cut -c1-130
rm -rf $O/prof_z $O/pm_sq
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pm_sq -o p -- python $R/tools/dev/pmc_attn_mix.py m14 > $O/r03z_pm_sq.log 2>&1
python $R/tools/dev/pmc_report.py $(find $O/pm_sq -name "*.db" | head -1) > $O/r03z_attn_pmc.txt 2>&1; cut -c1-230 $O/r03z_attn_pmc.txt
rm -rf $O/pm_sq
