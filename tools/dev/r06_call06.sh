#!/bin/bash
# round 6, call 6: after the prune + ADVICE changes: kernel tests, engine cells, the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "conv2d or gated or attention or merged_qkv or gn_partials or linear_gn or frame_tail or graph_replay or (bf16x6 and c2_r50_aotl_70 and throughput) or multi_group or end_to_end_vs_reference_golden" 2>&1 | tail -6
echo "== bench (default line, no other configs)"
timeout 1200 python bench.py --steps 20 --warmup 5 --no-other-configs 2> $O/r06_bench_a.err | tail -1 > $O/r06_bench_a.json
tail -12 $O/r06_bench_a.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_a.json').read().strip().splitlines()[-1]); c = d['config']
print('value', d['value'], c.get('repeat_fps'), 'whole', (c.get('whole_clip') or {}).get('fps'), 'single', c.get('single_stream'), 'online', c.get('single_stream_online'))
print('fp32_exact', (c.get('fp32_exact') or {}).get('value'))
print('jf', c.get('jf_vs_reference'))
r = d['roofline']; print('roofline', {k: r.get(k) for k in ('kernel', 'frac', 'achieved', 'avg_launch_us', 'traffic')}); print('gemm', r.get('gemm'))
print('cpu', d.get('cpu_baseline'))
PY
} > $O/r06_call06.txt 2>&1
cat $O/r06_call06.txt
