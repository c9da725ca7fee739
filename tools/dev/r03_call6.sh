# GPU call 6 of round 3 (gated kernel at two waves per SIMD; DeAOT kernel shares):  gpurun --timeout 900 -- 'bash tools/dev/r03_call6.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; L=$R/aot-benchmark_amd/csrc; cd $R
timeout 200 python bench.py --no-cpu-baseline --no-jf --no-roofline --no-x6 --steps 20 > /dev/null 2>&1      # warm-up, discarded
for v in base occ2; do
  lib=$L/libaot_hip.so; [ $v = occ2 ] && lib=$L/libaot_hip_occ2.so
  AOT_HIP_LIB=$lib timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "gated_attention" > $O/r03f_gated_$v.log 2>&1
  echo "$v gated tests rc=$? $(tail -1 $O/r03f_gated_$v.log)"
  AOT_HIP_LIB=$lib timeout 300 python bench.py --model r50_deaotl --steps 207 --repeats 2 --no-cpu-baseline --no-jf --no-x6 > $O/r03f_deaot_$v.json 2> $O/r03f_deaot_$v.err
  python -c "import json; d=json.load(open('$O/r03f_deaot_$v.json')); print('r50_deaotl $v', d['value'], d['config']['repeat_fps'], d['config']['single_stream']['fps'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
done
cd /tmp; rm -rf $O/prof_d
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_d -o p -- python $R/bench.py --model r50_deaotl --streams 1 --steps 138 --repeats 1 --no-cpu-baseline --no-jf --no-roofline --no-x6 > $O/r03f_prof_d.log 2>&1
python $R/tools/dev/prof_summary.py $(find $O/prof_d -name "*.db" | head -1) $O/r03f_r50_deaotl_kernel_stats_s1.txt | head -16 | cut -c1-130
rm -rf $O/prof_d; cd $R
timeout 400 python bench.py --no-cpu-baseline --no-roofline > $O/r03f_bench.json 2> $O/r03f_bench.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('$O/r03f_bench.json')); c=d['config']; print('f32', d['value'], c['single_stream']['fps'], 'x6', c['bf16x6_split']['value'], c['bf16x6_split']['repeat_fps'], c['bf16x6_split']['jf_vs_reference']['pixels_outside_near_ties'])"
