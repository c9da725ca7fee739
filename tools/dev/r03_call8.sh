# GPU call 8 of round 3 (split sweeps; DeAOT lines with the new split rule):  gpurun --timeout 900 -- 'bash tools/dev/r03_call8.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 200 python tools/dev/mb_attn.py "" sweep > $O/r03h_mb_attn_sweep.txt 2>&1; tail -1 $O/r03h_mb_attn_sweep.txt | tr '|' '\n'
timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "gated or deaot" > $O/r03h_deaot_tests.log 2>&1; echo "deaot tests rc=$? $(tail -1 $O/r03h_deaot_tests.log)"
for m in r50_deaotl swinb_deaotl; do
  timeout 400 python bench.py --model $m --no-x6 > $O/r03h_bench_$m.json 2> $O/r03h_bench_$m.err; echo "$m rc=$?"
  python -c "import json; d=json.load(open('$O/r03h_bench_$m.json')); c=d['config']; print('$m', d['value'], c['repeat_fps'], c['single_stream']['fps'], d['roofline']['frac'], d['roofline']['avg_launch_us'], c['jf_vs_reference']['pixels_outside_near_ties'], d['cpu_baseline']['value'])"
done
