"""Builds libaot_hip.so (gfx950) in-tree next to the sources.  hipcc cross-compiles without a GPU.

    python aot-benchmark_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['gemm_conv.hip', 'gemm_lds.hip', 'gemm_x6.hip', 'attention.hip', 'attention_x6.hip', 'attn_topk.hip', 'local_attn.hip', 'local_gated.hip', 'swin.hip', 'norm_act.hip', 'prepost.hip', 'train_ops.hip', 'train_bwd.hip']
LIB = os.path.join(HERE, 'libaot_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']
# Per-file additions.  -fno-slp-vectorize: hipcc's SLP vectoriser pairs scalar fp32 arithmetic into packed (VOP3P) instructions and,
# depending on the register allocation, emits IN-PLACE forms whose op_sel crosses the halves of the overwritten source -- not safe
# on gfx950 (common.h: scalar_fp32(); profiles/r04_hazard.txt).  The streaming / glue / training kernels below gain nothing
# measurable from packed fp32 (they are bound by HBM or launch latency), so they are built without it; the matrix-core kernels
# keep it (their inner loops use explicit packed asm) and are held to the ISA audit of tests/test_host.py like every other file.
# Round 5: EVERY source is built without the SLP vectoriser -- measured neutral on the GEMM set (2378 vs 2344 us), the bf16x6
# attention kernel (243.1 vs 242.0 us at a 14-frame bank) and the bench (743 vs 739 fps): profiles/r05_noslp.txt -- so that no
# compiler-formed packed fp32 instruction is left anywhere; attention.hip writes its packed fp32 arithmetic by hand (explicit
# v_pk_fma / v_pk_mul / v_pk_add, never with crossed halves): tests/test_host.py holds every other source to ZERO in-place packed
# fp32 instructions and attention.hip to those three opcodes.
EXTRA = {s: ['-fno-slp-vectorize'] for s in SOURCES}
# ... and without the packed-fp32 instructions altogether where none is written by hand (ext-vector arithmetic -- float4 products in
# the training kernels -- is otherwise legalised into v_pk_mul_f32 / v_pk_add_f32 in place); the host half of the compile does not
# know the feature and says so on stderr: those lines are dropped below
NO_PK = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
for _s in SOURCES:
    if _s != 'attention.hip':
        EXTRA[_s] = EXTRA[_s] + NO_PK
_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"



def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + ['common.h', 'conv_params.h', 'gemm_tile.h', 'build.py', os.path.join('..', '..', 'include', 'aot_hip.h')]
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps)


def build_lib(force=False, verbose=True, variant=None, defines=()):
    """The product library, or (variant='name', defines=['-DX=1', ...]) a variant next to it for A/B runs: libaot_hip_<name>.so,
    same per-source flags, its own object directory (tools/dev/build_variant.sh)."""
    lib = LIB if variant is None else os.path.join(HERE, 'libaot_hip_%s.so' % variant)
    if variant is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc')
    objdir = os.path.join(HERE, 'build' if variant is None else 'build_' + variant)
    os.makedirs(objdir, exist_ok=True)
    # one object per source (own flags), compiled side by side, then one link
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src[:-4] + '.o')
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + list(defines) + ['-c', os.path.join(HERE, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    objs = []
    for src, obj, pr in procs:
        err = pr.communicate()[1]
        err = ''.join(ln for ln in err.splitlines(True) if _NOISE not in ln)
        if err:
            sys.stderr.write(err)
        if pr.returncode != 0:
            raise subprocess.CalledProcessError(pr.returncode, 'hipcc -c ' + src)
        objs.append(obj)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', lib])
    return lib


def device_asm(outdir, sources=None, defines=()):
    """gfx950 assembly of every source (the flags of the build), for the ISA audits of tests/test_host.py."""
    hipcc = os.environ.get('HIPCC') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc')
    os.makedirs(outdir, exist_ok=True)
    procs, out = [], []
    for src in (sources or SOURCES):
        dst = os.path.join(outdir, src[:-4] + '.s')
        flags = [f for f in FLAGS if f != '-fPIC'] + EXTRA.get(src, []) + list(defines)
        procs.append(subprocess.Popen([hipcc] + flags + ['-S', '--cuda-device-only', os.path.join(HERE, src), '-o', dst],
                                      stderr=subprocess.DEVNULL))
        out.append(dst)
    for pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, 'hipcc -S')
    return out


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    print(LIB)
