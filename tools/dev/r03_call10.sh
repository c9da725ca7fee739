# GPU call 10 of round 3 (lgp_scores over window rows; clips per GPU):  gpurun --timeout 900 -- 'bash tools/dev/r03_call10.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 200 python bench.py --no-cpu-baseline --no-jf --no-roofline --no-x6 --steps 20 > /dev/null 2>&1      # warm-up, discarded
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "local_gated or layernorm_groupnorm or end_to_end_vs_reference_golden" > $O/r03i_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/r03i_tests.log)"; grep -E "^E  |^FAILED" $O/r03i_tests.log | head
for m in r50_deaotl swinb_deaotl; do
  timeout 300 python bench.py --model $m --no-x6 --no-cpu-baseline > $O/r03i_bench_$m.json 2> $O/r03i_bench_$m.err; echo "$m rc=$?"
  python -c "import json; d=json.load(open('$O/r03i_bench_$m.json')); c=d['config']; print('$m', d['value'], c['repeat_fps'], c['single_stream']['fps'], d['roofline']['frac'], c['jf_vs_reference']['pixels_outside_near_ties'])"
done
for s in 2 3 4 5; do
  timeout 200 python bench.py --streams $s --steps $((69 * s)) --repeats 2 --no-cpu-baseline --no-jf --no-roofline --no-x6 > $O/r03i_streams$s.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/r03i_streams$s.json')); print('streams $s', d['value'], d['config']['repeat_fps'], d['config']['peak_mem_gib'])"
done
