#!/bin/bash
# round 4, call 13: the whole GPU suite on the fixed library; the reproducer again (product kernels as reference + scalar-merge variants);
# bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 200 tools/dev/x6_hazard 10 > $O/r04_x6_hazard_g.txt 2>&1; grep "differing" $O/r04_x6_hazard_g.txt | grep -c "differing   0 /"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r04_gpu_suite.txt 2>&1; echo "gpu suite rc $?"; tail -n 6 $O/r04_gpu_suite.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench20b.json 2> $O/r04_bench20b.err; echo "bench rc $?"; tail -n 4 $O/r04_bench20b.err; python -c "
import json; d=json.load(open('$O/r04_bench20b.json')); c=d['config']; print(d['value'], c['single_stream']['fps'], c['whole_clip']['fps'], c['bf16x6_split']['value'], c['jf_vs_reference']['pixels_outside_near_ties'], c['bf16x6_split']['jf_vs_reference']['pixels_outside_near_ties'], d['roofline']['frac'], {k:(v.get('fps'), v.get('jf_vs_reference',{}).get('pixels_outside_near_ties')) for k,v in c['other_configs'].items()})"
