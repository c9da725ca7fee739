# GPU call 5 of round 3 (wide bf16x6 tile):  gpurun --timeout 900 -- 'bash tools/dev/r03_call5.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bf16x6_kernel" > $O/r03e_x6_kernel.log 2>&1
echo "x6 kernel tests rc=$? $(tail -1 $O/r03e_x6_kernel.log)"; grep -E "^E  |^FAILED" $O/r03e_x6_kernel.log | head -12
timeout 200 python tools/dev/mb_gemm.py -2,x6n,x6w,x6 > $O/r03e_mb_gemm_b1.txt 2>&1; tail -1 $O/r03e_mb_gemm_b1.txt
timeout 200 python tools/dev/mb_gemm.py -2,x6n,x6w,x6 "" "" 3 > $O/r03e_mb_gemm_b3.txt 2>&1; tail -1 $O/r03e_mb_gemm_b3.txt
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bf16x6_engine" > $O/r03e_x6_engine.log 2>&1
echo "x6 engine tests rc=$? $(tail -1 $O/r03e_x6_engine.log)"; grep -E "^E  " $O/r03e_x6_engine.log | head -12
timeout 400 python bench.py --no-cpu-baseline --no-roofline > $O/r03e_bench.json 2> $O/r03e_bench.err; echo "bench rc=$?"; tail -2 $O/r03e_bench.err
python -c "import json; d=json.load(open('$O/r03e_bench.json')); c=d['config']; print('f32', d['value'], c['single_stream']['fps'], 'x6', c['bf16x6_split']['value'], c['bf16x6_split']['repeat_fps'], c['bf16x6_split']['jf_vs_reference']['pixels_outside_near_ties'])"
