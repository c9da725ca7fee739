# GPU call 4 of round 3 (bf16x6 after the epilogue-load fix; batched Swin encoder):  gpurun --timeout 900 -- 'bash tools/dev/r03_call4.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bf16x6_kernel" > $O/r03d_x6_kernel.log 2>&1
echo "x6 kernel tests rc=$? $(tail -1 $O/r03d_x6_kernel.log)"; grep -E "^E  " $O/r03d_x6_kernel.log | head -12
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bf16x6_engine" > $O/r03d_x6_engine.log 2>&1
echo "x6 engine tests rc=$? $(tail -1 $O/r03d_x6_engine.log)"; grep -E "^E  " $O/r03d_x6_engine.log | head -12
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "free_running and swinb and fuse_probs-3 or swin_encoder" > $O/r03d_swin.log 2>&1
echo "swin tests rc=$? $(tail -1 $O/r03d_swin.log)"; grep -E "^E  " $O/r03d_swin.log | head -12
timeout 200 python tools/dev/mb_gemm.py -2,x6 "" "" 3 > $O/r03d_mb_gemm_b3.txt 2>&1; tail -1 $O/r03d_mb_gemm_b3.txt
timeout 400 python bench.py --no-cpu-baseline --no-roofline > $O/r03d_bench.json 2> $O/r03d_bench.err; echo "bench rc=$?"; tail -2 $O/r03d_bench.err
python -c "import json; d=json.load(open('$O/r03d_bench.json')); c=d['config']; print('f32', d['value'], c['single_stream']['fps'], 'x6', c['bf16x6_split'])" | cut -c1-1200
timeout 400 python bench.py --model swinb_deaotl --no-cpu-baseline --no-roofline --no-x6 > $O/r03d_bench_swinb.json 2> $O/r03d_bench_swinb.err; echo "swinb rc=$?"
python -c "import json; d=json.load(open('$O/r03d_bench_swinb.json')); c=d['config']; print('swinb', d['value'], c['repeat_fps'], c['single_stream'], c['jf_vs_reference']['pixels_outside_near_ties'])"
