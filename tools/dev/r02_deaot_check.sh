cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "gated_attention or local_gated or c3 or deaot or bilinear_and_finalize or multi_group or more_than_ten" > $O/deaottest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/deaottest.log
for m in r50_deaotl swinb_deaotl; do timeout 200 python tools/dev/time_model.py $m 40 2>/dev/null | tail -1; done
timeout 300 python bench.py --model swinb_deaotl --no-cpu-baseline --steps 207 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench swinb_deaotl', d['value'], d['config']['single_stream'], d['config']['peak_mem_gib'])"
timeout 300 python bench.py --model r50_deaotl --no-cpu-baseline --steps 207 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench r50_deaotl', d['value'], d['config']['single_stream'])"
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-jf --steps 207 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench r50_aotl', d['value'], d['config']['single_stream'])"
