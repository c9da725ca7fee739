# round-2 record pass: full GPU test suite, bench in the driver's form, bench with a whole clip per stream, rocprofv3 kernel
# stats of the single-stream bench and of the 3-stream bench, PMC traffic of the attention kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench20.log 2>&1; tail -1 $O/bench20.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench207.log 2>&1; tail -1 $O/bench207.log | cut -c1-300
timeout 400 python bench.py --model swinb_deaotl --no-cpu-baseline > $O/bench_swinb_deaotl.log 2>&1; tail -1 $O/bench_swinb_deaotl.log | cut -c1-200
cd /tmp
for S in 1 3; do   # (kernel stats: launches from the host, so that every kernel carries its name)
  rm -rf $O/prof_s$S
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_s$S -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --no-jf --graph 0 --streams $S --steps $((69*S)) > $O/prof_s$S.log 2>&1
  python $R/tools/dev/prof_summary.py $(find $O/prof_s$S -name "*.db" | head -1) $O/bench_s${S}_kernel_stats.txt | head -24
  rm -rf $O/prof_s$S
done
rm -rf $O/pmc_f $O/pmc_w
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o p -- python $R/tools/dev/pmc_attn_mix.py > $O/pmc_f.log 2>&1 || echo "fetch pass failed/timeout"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o p -- python $R/tools/dev/pmc_attn_mix.py > $O/pmc_w.log 2>&1 || echo "write pass failed/timeout"
cd $R
python tools/dev/attn_traffic.py $(find $O/pmc_f -name "*.db" | head -1) $(find $O/pmc_w -name "*.db" | head -1) $O/attn_traffic.json | grep "traffic_bytes_per_launch"
rm -rf $O/pmc_f $O/pmc_w
