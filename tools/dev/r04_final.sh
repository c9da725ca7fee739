#!/bin/bash
# round 4 closing pass on the tree as committed: the whole GPU suite, the driver's bench line, rocprofv3 kernel stats of the same bench
# command, the training step under RCCL (both precisions) with its kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -6 > $O/r04z_gpu_suite.txt
cat $O/r04z_gpu_suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04z_bench20.json 2> $O/r04z_bench20.err; tail -c 600 $O/r04z_bench20.json; echo
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04z_bench20.json').read().strip().splitlines()[-1])
c = d['config']
print('value', d['value'], d['dtype'], 'roofline', {k: d['roofline'][k] for k in ('achieved', 'peak', 'frac', 'traffic')})
print('whole_clip', c['whole_clip']['fps'], 'single', c['single_stream']['fps'], 'fp32_exact', c.get('fp32_exact', {}).get('value'))
print('jf', c['jf_vs_reference'])
for k, v in c['other_configs'].items():
    print(k, v.get('fps'), v.get('whole_clip_fps'), v.get('single_stream_fps'), v.get('jf_vs_reference', {}).get('pixels_outside_near_ties'), v.get('roofline', {}).get('frac'))
print('cpu', d['cpu_baseline'])
PY
for prec in f32 bf16; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 6 2>/dev/null | tail -n 1 > $O/r04z_train_ddp_$prec.json; cut -c1-150 $O/r04z_train_ddp_$prec.json
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-whole-clip > $O/r04z_bench_prof.json 2> $O/r04z_bench_prof.err
python $R/tools/dev/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/r04z_bench20_kernel_stats.txt | head -12
rm -rf $O/prof
for prec in f32 bf16; do
  WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 3 > $O/r04z_train_prof_$prec.json 2> $O/r04z_train_prof_$prec.err
  python $R/tools/dev/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/r04z_train_step_kernel_stats_$prec.txt | head -8
  rm -rf $O/prof
done
