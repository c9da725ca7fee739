# GPU call 3 of round 3 (bf16x6 family bring-up, long-clip graphs, gated XCD walk):  gpurun --timeout 1000 -- 'bash tools/dev/r03_call3.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; L=$R/aot-benchmark_amd/csrc; cd $R
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "survives_long_clips or graph_replay_bit or encode_ahead" > $O/r03c_graph.log 2>&1
echo "graph tests rc=$? $(tail -1 $O/r03c_graph.log)"; grep -E "^E  " $O/r03c_graph.log | head -6
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bf16x6_kernel" > $O/r03c_x6_kernel.log 2>&1
echo "x6 kernel tests rc=$? $(tail -1 $O/r03c_x6_kernel.log)"; grep -E "^E  " $O/r03c_x6_kernel.log | head -12
timeout 200 python tools/dev/mb_gemm.py -2,x6 > $O/r03c_mb_gemm_b1.txt 2>&1; tail -1 $O/r03c_mb_gemm_b1.txt
timeout 200 python tools/dev/mb_gemm.py -2,x6 "" "" 3 > $O/r03c_mb_gemm_b3.txt 2>&1; tail -1 $O/r03c_mb_gemm_b3.txt
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bf16x6_engine" > $O/r03c_x6_engine.log 2>&1
echo "x6 engine tests rc=$? $(tail -1 $O/r03c_x6_engine.log)"; grep -E "^E  " $O/r03c_x6_engine.log | head -12
timeout 400 python bench.py --no-cpu-baseline --no-roofline > $O/r03c_bench.json 2> $O/r03c_bench.err; echo "bench rc=$?"; tail -3 $O/r03c_bench.err
python -c "import json; d=json.load(open('$O/r03c_bench.json')); c=d['config']; print('f32', d['value'], c['single_stream']['fps'], 'x6', c['bf16x6_split'])" | cut -c1-1500
# gated kernel: XCD-aware walk vs plain, time and fabric traffic
for v in base gxcd1; do
  lib=$L/libaot_hip.so; [ $v = gxcd1 ] && lib=$L/libaot_hip_gxcd1.so
  AOT_HIP_LIB=$lib timeout 300 python bench.py --model r50_deaotl --steps 207 --repeats 2 --no-cpu-baseline --no-jf --no-x6 > $O/r03c_deaot_$v.json 2> $O/r03c_deaot_$v.err
  python -c "import json; d=json.load(open('$O/r03c_deaot_$v.json')); print('r50_deaotl $v', d['value'], d['config']['repeat_fps'], d['config']['single_stream']['fps'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pm_$c
    AOT_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py gated > $O/r03c_pm_${v}_$c.log 2>&1 || echo "pass $v $c failed"
  done
  cd $R
  python tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/r03c_gated_attn_traffic_$v.json attn_fwd_wide_coop_kernel > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/r03c_gated_attn_traffic_$v.json')); print('$v gated traffic/launch', d['traffic_bytes_per_launch'], d['bytes_per_launch'])"
  rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE
done
