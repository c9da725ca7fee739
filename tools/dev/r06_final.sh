#!/bin/bash
# round 6, closing pass: the whole GPU suite on HEAD, then the driver's command (twice: the boxes of the pool differ by a few %)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 3300 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/r06z_suite.txt
cat $O/r06z_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06z_smoke.txt 2>&1; tail -2 $O/r06z_smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06z_bench20.json 2> $O/r06z_bench20.err
tail -1 $O/r06z_bench20.json | cut -c1-400
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06z2_bench20.json 2> $O/r06z2_bench20.err
tail -1 $O/r06z2_bench20.json | cut -c1-400
