cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_d
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_d -o p -- python $R/scratch/dev_time2.py r50_deaotl 70 > $O/prof_d.log 2>&1
grep fps $O/prof_d.log
cd $R
python scratch/prof_summary.py $(find $O/prof_d -name "*.db" | head -1) $O/deaot_kernel_stats.txt | head -28
timeout 150 python scratch/dev_time2.py swinb_deaotl 70 2>&1 | grep fps
rm -rf $O/prof_d
