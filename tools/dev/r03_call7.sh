# GPU call 7 of round 3 (gated kernel: split sweep at two waves per SIMD):  gpurun --timeout 600 -- 'bash tools/dev/r03_call7.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python tools/dev/mb_gated.py > $O/r03g_mb_gated.txt 2>&1; cat $O/r03g_mb_gated.txt | cut -c1-330
