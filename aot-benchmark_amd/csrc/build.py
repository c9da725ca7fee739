"""Builds libaot_hip.so (gfx950) in-tree next to the sources.  hipcc cross-compiles without a GPU.

    python aot-benchmark_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['gemm_conv.hip', 'gemm_lds.hip', 'attention.hip', 'attention_x6.hip', 'attn_topk.hip', 'local_attn.hip', 'local_gated.hip', 'swin.hip', 'norm_act.hip', 'prepost.hip', 'train_ops.hip', 'train_bwd.hip']
LIB = os.path.join(HERE, 'libaot_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-Wno-unused-result']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + ['common.h', 'conv_params.h', 'build.py', os.path.join('..', '..', 'include', 'aot_hip.h')]
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps)


def build_lib(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc')
    cmd = [hipcc] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    print(LIB)
