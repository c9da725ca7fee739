# GPU call 1 of round 3:  gpurun --timeout 1500 -- 'bash tools/dev/r03_call1.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; L=aot-benchmark_amd/csrc
B="python bench.py --no-cpu-baseline --no-jf --no-roofline"
# 0. throw-away warm-up (the first process on a fresh box reads ~10 % low)
timeout 200 $B --steps 20 > /dev/null 2>&1
# 1. whole GPU suite with the round-3 parity cells
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/r03a_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/r03a_pytest.log)"
# 2. attention variants, per-launch sweep
for v in base pksum1 order1 coop1; do
  lib=$L/libaot_hip_$v.so; [ $v = base ] && lib=$L/libaot_hip.so
  timeout 120 python tools/dev/mb_attn.py $lib 2>&1 | tail -1
done > $O/r03a_mb_attn.txt; cat $O/r03a_mb_attn.txt
# 3. GEMM tile ends
for v in base epi1; do
  lib=$L/libaot_hip_$v.so; [ $v = base ] && lib=$L/libaot_hip.so
  timeout 200 python tools/dev/mb_gemm.py -1,-2 $lib > $O/r03a_mb_gemm_$v.txt 2>&1; echo "$v $(tail -1 $O/r03a_mb_gemm_$v.txt)"
done
timeout 200 tools/dev/gemm_check quick > $O/r03a_gemm_check.txt 2>&1; echo "gemm_check rc=$? ERR=$(grep -c ERR $O/r03a_gemm_check.txt)"; tail -2 $O/r03a_gemm_check.txt | cut -c1-600
# 4. end to end: encode-ahead on / off, variants
for ea in 1 3; do
  timeout 300 $B --steps 207 --repeats 2 --encode-ahead $ea > $O/r03a_bench_ea$ea.json 2> $O/r03a_bench_ea$ea.err
  python -c "import json; d=json.load(open('$O/r03a_bench_ea$ea.json')); print('ahead $ea', d['value'], d['config']['repeat_fps'], d['config']['single_stream']['fps'], d['config']['single_stream']['repeat_fps'])"
done
for v in epi1 pksum1 order1; do
  AOT_HIP_LIB=$PWD/$L/libaot_hip_$v.so timeout 300 $B --steps 207 --repeats 2 > $O/r03a_bench_$v.json 2> $O/r03a_bench_$v.err
  python -c "import json; d=json.load(open('$O/r03a_bench_$v.json')); print('$v', d['value'], d['config']['repeat_fps'], d['config']['single_stream']['fps'], d['config']['single_stream']['repeat_fps'])"
done
for v in base coop1; do
  lib=$PWD/$L/libaot_hip_$v.so; [ $v = base ] && lib=$PWD/$L/libaot_hip.so
  AOT_HIP_LIB=$lib timeout 300 $B --model r50_deaotl --steps 207 --repeats 2 > $O/r03a_deaot_$v.json 2> $O/r03a_deaot_$v.err
  python -c "import json; d=json.load(open('$O/r03a_deaot_$v.json')); print('r50_deaotl $v', d['value'], d['config']['repeat_fps'], d['config']['single_stream']['fps'])"
done
# 5. the driver's form, complete line (J&F leg with the near-tie check, roofline, cpu baseline)
timeout 400 python bench.py --steps 20 --warmup 5 > $O/r03a_bench20.json 2> $O/r03a_bench20.err; echo "bench20 rc=$?"; cat $O/r03a_bench20.json | cut -c1-1500
