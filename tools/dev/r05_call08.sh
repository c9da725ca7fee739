#!/bin/bash
# round 5, call 8: the merged Q|K|V product + the one-kernel frame tail: kernel test, parity cells, bench A/B; the library built
# without the SLP vectoriser (every source but attention.hip) against the shipped one
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
C=aot-benchmark_amd/csrc
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "frame_tail or (bf16x6 and tail) or (bf16x6 and c2_r50_aotl_70 and (free_running or full_size))" 2>&1 | tail -6
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
echo "== bench, no merge / no tail fusion"; AOT_NO_QKV_MERGE=1 AOT_NO_TAIL=1 timeout 600 python bench.py $F --no-jf 2>/dev/null | tail -1 | cut -c1-900
echo "== bench, round-5 host path"; timeout 600 python bench.py $F 2>/dev/null | tail -1 | cut -c1-2600
echo "== bench, round-5 host path, library without the SLP vectoriser"; AOT_HIP_LIB=$PWD/$C/libaot_hip_noslp.so timeout 600 python bench.py $F --no-jf 2>/dev/null | tail -1 | cut -c1-900
echo "== attention x6, shipped"; timeout 200 python tools/dev/mb_attn_x6.py "" quick 2>&1 | grep -v amdgpu.ids | tail -6
echo "== attention x6, no SLP"; timeout 200 python tools/dev/mb_attn_x6.py $C/libaot_hip_noslp.so quick 2>&1 | grep -v amdgpu.ids | tail -6
echo "== gemm set, shipped / no SLP (batch 3)"
timeout 200 python tools/dev/mb_gemm.py x6 "" "" 3 2>&1 | tail -1
timeout 200 python tools/dev/mb_gemm.py x6 $C/libaot_hip_noslp.so "" 3 2>&1 | tail -1
} > $O/r05_call08.txt 2>&1
cat $O/r05_call08.txt | cut -c1-1200
