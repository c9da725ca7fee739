# round-2 GPU pass 2: hipGraph replay -- parity test, bench with/without graphs at 1/3/4/6 streams, a clean
# single-stream kernel profile, and the two PMC passes for the attention kernel's fabric traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "graph_replay or concurrent_clips or more_than_ten or free_running_masks" > $O/graphtest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/graphtest.log
Q="--no-cpu-baseline --no-roofline --no-jf --steps 207 --warmup 5"
for cfg in "1 3" "0 3" "1 1" "0 1" "1 4" "1 6" "1 2"; do
  set -- $cfg
  timeout 300 python bench.py $Q --graph $1 --streams $2 > $O/bench_g$1_s$2.log 2>&1
  echo "graph=$1 streams=$2: $(tail -1 $O/bench_g$1_s$2.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps; single", d["config"]["single_stream"])
except Exception as e: print("FAILED", e)')"
done
tail -5 $O/bench_g1_s3.log | cut -c1-600
cd /tmp
rm -rf $O/prof_s1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --no-jf --graph 0 --streams 1 --steps 69 > $O/prof_s1.log 2>&1
cd $R
python tools/dev/prof_summary.py $(find $O/prof_s1 -name "*.db" | head -1) $O/bench_s1_kernel_stats.txt | head -45
rm -rf $O/prof_s1
cd /tmp
rm -rf $O/pmc_f $O/pmc_w
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o p -- python $R/tools/dev/pmc_attn_mix.py > $O/pmc_f.log 2>&1 || echo "fetch pass failed/timeout"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o p -- python $R/tools/dev/pmc_attn_mix.py > $O/pmc_w.log 2>&1 || echo "write pass failed/timeout"
cd $R
python tools/dev/attn_traffic.py $(find $O/pmc_f -name "*.db" | head -1) $(find $O/pmc_w -name "*.db" | head -1) $O/attn_traffic.json | tail -14
rm -rf $O/pmc_f $O/pmc_w
./tools/dev/bf16x6_rate 20000 | tee $O/bf16x6_rate.txt
