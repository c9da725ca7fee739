cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_f $O/pmc_f $O/pmc_w
rocprofv3 --kernel-trace --stats -d $O/prof_f -o p -- python $R/bench.py --no-cpu-baseline > $O/prof_f_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 69 --warmup 2 > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 69 --warmup 2 > $O/pmc_w.log 2>&1
cd $R
python scratch/prof_summary.py $(find $O/prof_f -name "*.db" | head -1) $O/bench_kernel_stats.txt | head -30
python scratch/attn_traffic.py $(find $O/pmc_f -name "*.db" | head -1) $(find $O/pmc_w -name "*.db" | head -1) $O/attn_traffic.json | tail -12
tail -1 $O/prof_f_bench.log
rm -rf $O/pmc_f $O/pmc_w     # the raw counter DBs are large; the JSON holds the sums
