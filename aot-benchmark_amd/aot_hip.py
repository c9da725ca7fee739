"""ctypes binding of libaot_hip.so (the C ABI declared in include/aot_hip.h).

Host-side plumbing only: torch supplies device memory and the stream, every op below is one
asynchronous launch of a hand-written gfx950 kernel.  There is NO CPU path: if the library
is missing, or a tensor is not on a ROCm device, the call raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AOT_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libaot_hip.so')   # (AOT_HIP_LIB: A/B runs of variant builds, tools/dev)
_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_long

_SIGS = {
    'aot_conv2d_nhwc_f32': [_P] * 7 + [_L] + [_I] * 20 + [_P],
    'aot_pack_bf16x6_f32': [_P, _P, _I, _I, _I, _I, _P],
    'aot_conv2d_bf16x6_f32': [_P, _P, _I, _P, _P, _P] + [_I] * 18 + [_P],
    'aot_conv2d_bf16x6k_f32': [_P, _P, _I, _P, _P, _P] + [_I] * 18 + [_P, _L, _P],
    'aot_linear_group_bf16x6_f32': [_I, _P, _P, _I, _P, _P, _P] + [_I] * 8 + [_P],
    'aot_linear_bf16x6k_ln_f32': [_P, _P, _I, _P, _P, _P] + [_I] * 9 + [_P, _L, _P, _P, _P, _I, _F, _P],
    'aot_conv2d_c4_bf16x6_f32': [_P, _P, _I, _P, _P] + [_I] * 13 + [_P],
    'aot_pack_bf16_f32': [_P, _P, _I, _I, _I, _I, _P],
    'aot_conv2d_bf16_f32': [_P, _P, _I, _P, _P, _P] + [_I] * 18 + [_P, _P],
    'aot_dwconv2d_nhwc_f32': [_P] * 4 + [_I] * 12 + [_P],
    'aot_maxpool3x3s2_nhwc_f32': [_P, _P] + [_I] * 5 + [_P],
    'aot_nchw_to_nhwc_f32': [_P, _P] + [_I] * 4 + [_P],
    'aot_nhwc_to_nchw_f32': [_P, _P] + [_I] * 4 + [_P],
    'aot_layernorm_f32': [_P] * 6 + [_I] * 7 + [_F, _P],
    'aot_groupnorm_stats_f32': [_P] * 4 + [_I] * 5 + [_F, _I, _P],
    'aot_groupnorm_apply_f32': [_P] * 6 + [_I] * 9 + [_P],
    'aot_gn_act_dwconv5_f32': [_P] * 6 + [_I] * 8 + [_P],
    'aot_linear_gn_bf16x6_f32': [_P, _P, _I, _P, _P, _P] + [_I] * 8 + [_P, _L, _P],
    'aot_gn_act_dwconv5p_f32': [_P, _P, _I, _P, _P, _P, _P] + [_I] * 7 + [_F, _P],
    'aot_layernorm_linear_bf16x6_f32': [_P, _P, _I, _P, _P, _P, _P] + [_I] * 8 + [_F, _P, _L, _P],
    'aot_attn_f32': [_P] * 5 + [_I, _L, _I, _I, _P] + [_I] * 6 + [_F, _I, _P],
    'aot_attn_merge_f32': [_P, _P, _P] + [_I] * 6 + [_P],
    'aot_attn_pack_x6_f32': [_P] * 3 + [_I, _L, _I, _L, _I, _I, _L, _P, _I, _P],
    'aot_attn_x6_f32': [_P] * 4 + [_I, _L, _I, _I, _P] + [_I] * 4 + [_F, _I, _P],
    'aot_attn_pack_x6_part_f32': [_P, _P, _I, _L, _I, _L, _I, _L, _P, _I, _I, _P],
    'aot_gated_attn_x6_f32': [_P] * 6 + [_I, _L, _I, _I, _P] + [_I] * 5 + [_F, _I, _P],
    'aot_preprocess_f32': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P],
    'aot_fuse_probs_f32': [_P] * 5 + [_I] * 5 + [_P],
    'aot_label_resize_f32': [_P, _P] + [_I] * 5 + [_P],
    'aot_attn_topk_f32': [_P] * 5 + [_I] * 8 + [_F, _I, _P],
    'aot_gated_attn_topk_f32': [_P] * 6 + [_I] * 9 + [_F, _I, _P],
    'aot_gated_attn_f32': [_P] * 6 + [_I, _L, _I, _I, _P] + [_I] * 7 + [_F, _I, _P],
    'aot_local_attn_f32': [_P] * 7 + [_I, _L] + [_I] * 9 + [_F, _P],
    'aot_local_gated_f32': [_P] * 8 + [_I, _L] + [_I] * 10 + [_F, _P],
    'aot_swin_window_attn_f32': [_P] * 4 + [_I] * 9 + [_F, _P],
    'aot_patch_merge_f32': [_P, _P] + [_I] * 4 + [_P],
    'aot_idbank_f32': [_P] * 5 + [_I] * 13 + [_P, _P] + [_I] * 3 + [_P],
    'aot_bilinear_nhwc_f32': [_P] * 3 + [_I] * 11 + [_P],
    'aot_gn_bilinear_nhwc_f32': [_P] * 6 + [_I] * 13 + [_P],
    'aot_gn_conv1x1_f32': [_P] * 7 + [_I] * 10 + [_P],
    'aot_logits_finalize_f32': [_P] * 3 + [_I] * 9 + [_P],
    'aot_frame_tail_f32': [_P] * 4 + [_I] * 10 + [_P],
    'aot_add_f32': [_P] * 3 + [_L, _P],
    'aot_copy_rows_f32': [_P, _P, _I, _L, _I, _L, _L, _I, _I, _P, _I, _P],
    # training-side stages (csrc/train_ops.hip)
    'aot_ce_loss_f32': [_P] * 6 + [_I, _I, _L, _L, _P],
    'aot_ce_loss_bwd_f32': [_P] * 6 + [_I, _I, _L, _P],
    'aot_soft_jaccard_f32': [_P] * 5 + [_I, _I, _L, _I, _F, _P],
    'aot_soft_jaccard_bwd_f32': [_P] * 5 + [_I, _I, _L, _F, _P],
    'aot_adamw_step_f32': [_P] * 4 + [_L] + [_F] * 5 + [_I, _F, _P],
    'aot_ema_update_f32': [_P, _P, _L, _F, _P],
    'aot_sumsq_accum_f64': [_P, _L, _P, _P],
    'aot_sumsq_flat_f64': [_P, _L, _P, _I, _P, _P, _P],
    'aot_adamw_flat_f32': [_P] * 4 + [_L, _P, _P, _I, _F, _F, _F, _P, _F, _P],
    # training path, differentiable primitives (csrc/train_bwd.hip)
    'aot_matmul_strided_f32': [_P] * 4 + [_I] * 4 + [_L] * 7 + [_I, _F, _I, _P],
    'aot_im2col_f32': [_P, _P] + [_I] * 11 + [_P],
    'aot_col2im_f32': [_P, _P] + [_I] * 11 + [_P],
    'aot_dwconv2d_bwd_data_f32': [_P] * 3 + [_I] * 11 + [_P],
    'aot_dwconv2d_bwd_weight_f32': [_P] * 3 + [_I] * 11 + [_P],
    'aot_act_f32': [_P, _P, _L, _I, _P],
    'aot_act_bwd_f32': [_P, _P, _P, _L, _I, _P],
    'aot_layernorm_bwd_f32': [_P] * 5 + [_I, _I, _F, _P],
    'aot_groupnorm_bwd_f32': [_P] * 6 + [_I] * 4 + [_P],
    'aot_groupnorm_bwd2_f32': [_P] * 9 + [_I] * 5 + [_P],
    'aot_norm_param_grads_f32': [_P] * 4 + [_L, _I, _P],
    'aot_col_reduce_f32': [_P] * 4 + [_L, _I, _P, _P, _I, _P],
    'aot_transpose_pad_f32': [_P, _P, _L, _I, _L, _L, _L, _I, _P],
    'aot_gather_cols_f32': [_P, _P, _P, _L, _I, _I, _P],
    'aot_copy2d_pad_f32': [_P, _P, _L, _L, _L, _L, _L, _L, _L, _P],
    'aot_softmax_rows_f32': [_P, _P, _L, _I, _P],
    'aot_softmax_rows_bwd_f32': [_P, _P, _P, _L, _I, _P],
    'aot_bilinear_bwd_nhwc_f32': [_P, _P] + [_I] * 7 + [_P],
    'aot_window_gather_f32': [_P, _P] + [_I] * 4 + [_F, _P],
    'aot_window_scatter_f32': [_P, _P] + [_I] * 4 + [_F, _P],
}

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_GELU, ACT_SILU = 0, 1, 2, 3, 4

# Which kernel table aot_conv2d_nhwc_f32 uses when a caller leaves the choice open (cfg = -1): 'latency' = fastest launch on
# its own (one clip at a time), 'throughput' = cheapest in SIMD time when several clips share the GPU (include/aot_hip.h).
# The choice belongs to an ENGINE (build_engine(..., gemm_table=)): every stage of an engine runs inside
# `with use_gemm_table(engine.gemm_table)`, so two engines of one process can use different tables and nothing outlives
# the stage call.  Outside any scope the table is 'latency'.
GEMM_TABLES = {'latency': -1, 'throughput': -2}
# Which matrix-core arithmetic those calls use: 'f32' = v_mfma_f32_32x32x2_f32 (exact fp32 products, the default), 'bf16x6' = the
# fp32-equivalent six-term bf16 split (aot_conv2d_bf16x6_f32; include/aot_hip.h) wherever a layer qualifies.  An engine attribute
# too (build_engine(..., mfma=)), carried by the same scope.
MFMA_MODES = ('f32', 'bf16x6')
X6_TILE = int(os.environ.get('AOT_X6_TILE', '0'))   # 0: kernel of the bf16x6 family chosen by shape; 66 / 129 force one (tests, tuning)
X6K_SCRATCH_FLOATS = 8 << 20   # floats of the per-stream split-K scratch (32 MB: every stride-16 layer of a 480p frame at three lanes fits)
X6_MIN_TILES = 16            # 64x64 output tiles below which a layer stays on the fp32 kernels (their split-K / small-tile forms)


class _Scopes(threading.local):
    """The scope stack is per host thread: engines driven from different threads do not see each other's tables."""

    def __init__(self):
        self.stack = []


_scopes = _Scopes()


class use_gemm_table:
    """Scope in which conv2d / linear calls that leave the kernel choice open use the named dispatch table (and arithmetic)."""

    def __init__(self, name, mfma='f32'):
        if mfma not in MFMA_MODES:
            raise ValueError('mfma must be one of %s' % (MFMA_MODES,))
        self.ent = (GEMM_TABLES[name], mfma == 'bf16x6')

    def __enter__(self):
        _scopes.stack.append(self.ent)
        return self

    def __exit__(self, *exc):
        _scopes.stack.pop()
        return False


def gemm_table():
    """Name of the table (and arithmetic) in force here (graph keys carry it: a graph captured under one never replays under
    another)."""
    if not _scopes.stack:
        return 'latency'
    cfg, x6 = _scopes.stack[-1]
    return ('throughput' if cfg == -2 else 'latency') + ('+bf16x6' if x6 else '')


class AotHipError(RuntimeError):
    pass


def load():
    """Loads the library (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AotHipError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(there is no CPU fallback for the AOT hot path)' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = sig
            fn.restype = _I
        lib.aot_hip_version.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(list(_SIGS) + ['aot_hip_version'])


def version():
    return load().aot_hip_version().decode()


def stream_ptr():
    """Raw handle of torch's current HIP stream on the current device."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


_DEBUG_SYNC = bool(os.environ.get('AOT_HIP_DEBUG_SYNC'))     # development aid: wait for every launch and name the one that faults


def _chk(rc, name):
    if rc != 0:
        raise AotHipError('%s failed with code %d' % (name, rc))
    if _DEBUG_SYNC:
        import sys
        print('[aot_hip] %s' % name, file=sys.stderr, flush=True)
        torch.cuda.synchronize()


def _dev(t):
    if not t.is_cuda:
        raise AotHipError('AOT HIP ops need ROCm device tensors (got %s); there is no CPU fallback' % t.device)
    return t.data_ptr()


def _opt(t):
    return None if t is None else _dev(t)


# ---- thin wrappers: 2-D token-major tensors [M, C] with row stride = t.stride(0) ----------------
def attach_wt(w, cin=None):
    """Gives a packed GEMM weight w [K, ld] its k-contiguous twin [Cout_ld, K] (attribute `_aot_wt`), which lets
    aot_conv2d_nhwc_f32 pick the LDS-direct tile kernel (csrc/gemm_lds.hip) when the shape allows it."""
    if w.shape[0] % 32 == 0 and (cin is None or cin % 32 == 0):
        w._aot_wt = w.t().contiguous()
    return _register_weight(w)


_gemm_weights = None     # weak set of every packed GEMM weight (attach_wt): what pack_bf16x6_all() walks


def _register_weight(w):
    global _gemm_weights
    if _gemm_weights is None:
        import weakref
        _gemm_weights = weakref.WeakSet()
    _gemm_weights.add(w)
    return w


def pack_bf16x6(w):
    """Splits a packed GEMM weight w [K, ld] (K % 32 == 0) into the three bf16 planes of the bf16x6 kernels, once; kept on the
    tensor (`_aot_w6`, int16 [3, K/32, 4, cout_pad, 8]).  Must not happen inside a graph capture: engines in bf16x6 mode call
    pack_bf16x6_all() before their first captured stage."""
    w6 = getattr(w, '_aot_w6', None)
    if w6 is not None:
        return w6
    if torch.cuda.is_current_stream_capturing():
        raise AotHipError('a GEMM weight reached the bf16x6 path unpacked inside a graph capture: call aot_hip.pack_bf16x6_all() '
                          '(or model.prepare(mfma="bf16x6")) first')
    K, ld = w.shape
    if K % 32:
        raise AotHipError('bf16x6 weights need K % 32 == 0')
    cout_pad = (ld + 63) // 64 * 64
    w6 = torch.empty(3, K // 32, 4, cout_pad, 8, dtype=torch.int16, device=w.device)
    _chk(load().aot_pack_bf16x6_f32(_dev(w), _dev(w6), K, ld, w.stride(0), cout_pad, stream_ptr()), 'aot_pack_bf16x6_f32')
    w._aot_w6 = w6
    return w6


def pack_bf16x6_all():
    """Packs every registered GEMM weight that qualifies (on the device, K % 32 == 0) and has no bf16x6 twin yet."""
    for w in list(_gemm_weights or ()):
        if w.is_cuda and w.dim() == 2 and w.shape[0] % 32 == 0 and getattr(w, '_aot_w6', None) is None:
            pack_bf16x6(w)
        taps = getattr(w, '_aot_c4_taps', 0)
        if w.is_cuda and taps and getattr(w, '_aot_w6c4', None) is None:
            pack_bf16x6_c4(w, taps)


def conv2d(x, w, bias, out, H, W, Cin, OH, OW, Cout, KH=1, KW=1, stride=1, pad=0, dil=1, res=None, act=ACT_NONE,
           B=1, res_rows=0, cfg=-1, stream=None):
    """B NHWC images [B*H*W, lda] -> [B*OH*OW, ldc]; res_rows > 0: the residual is one [res_rows, ldr] map shared by the
    images (row m % res_rows)."""
    stack = _scopes.stack
    if cfg == -1 and stack and stack[-1][1] and Cin % 32 == 0 and Cout > 32 and \
            -(-B * OH * OW // 64) * -(-Cout // 64) >= X6_MIN_TILES:
        w6 = getattr(w, '_aot_w6', None)
        if w6 is None:
            w6 = pack_bf16x6(w)
        ks = x6_ksplit(B * OH * OW, Cout, KH * KW * Cin) if X6_TILE == 0 else 1
        if ks != 1:
            return conv2d_x6k(x, w, bias, out, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil, res=res, act=act, B=B,
                              res_rows=res_rows, ksplit=ks, stream=stream)
        _chk(load().aot_conv2d_bf16x6_f32(_dev(x), _dev(w6), w6.shape[3], _opt(bias), _opt(res), _dev(out), B, H, W, Cin, OH,
                                          OW, Cout, KH, KW, stride, pad, dil, x.stride(0), out.stride(0),
                                          res.stride(0) if res is not None else 0, res_rows, act, X6_TILE,
                                          stream if stream is not None else stream_ptr()), 'aot_conv2d_bf16x6_f32')
        return out
    wt = getattr(w, '_aot_wt', None)
    _chk(load().aot_conv2d_nhwc_f32(_dev(x), _dev(w), _opt(wt), _opt(bias), _opt(res), _dev(out), None, 0, B, H, W, Cin, OH,
                                    OW, Cout, KH, KW, stride, pad, dil, x.stride(0), w.stride(0),
                                    wt.stride(0) if wt is not None else 0, out.stride(0),
                                    res.stride(0) if res is not None else 0, res_rows, act,
                                    (stack[-1][0] if stack else -1) if cfg == -1 else cfg,
                                    stream if stream is not None else stream_ptr()), 'aot_conv2d_nhwc_f32')
    return out


def pack_bf16x6_c4(w, taps):
    """The bf16x6 planes of a four-channel KxK weight w [taps * 4, ld] for aot_conv2d_c4_bf16x6_f32: rows zero-padded to eight taps per
    k-step, then aot_pack_bf16x6_f32; kept on the tensor (`_aot_w6c4`).  Not inside a graph capture (pack_bf16x6_all() covers it)."""
    w6 = getattr(w, '_aot_w6c4', None)
    if w6 is not None:
        return w6
    if torch.cuda.is_current_stream_capturing():
        raise AotHipError('a four-channel weight reached the bf16x6 path unpacked inside a graph capture: call aot_hip.pack_bf16x6_all() first')
    kp = -(-taps // 8) * 32
    wp = torch.zeros(kp, w.shape[1], dtype=torch.float32, device=w.device)
    wp[:taps * 4] = w[:taps * 4]
    w._aot_w6c4 = pack_bf16x6(wp)
    return w._aot_w6c4


def conv2d_c4(x, w, bias, out, H, W, OH, OW, Cout, KH, KW, stride, pad, dil=1, act=ACT_NONE, B=1, stream=None):
    """KxK convolution of B four-channel NHWC images [B*H*W, 4] (the ResNet stem).  In a bf16x6 scope with the kernel choice left
    open: aot_conv2d_c4_bf16x6_f32 (one chunk = one filter tap); otherwise the fp32 dispatch of conv2d."""
    stack = _scopes.stack
    if stack and stack[-1][1] and X6_TILE == 0 and Cout > 32 and x.stride(0) == 4 and not os.environ.get('AOT_NO_C4'):
        w6 = pack_bf16x6_c4(w, KH * KW)
        _chk(load().aot_conv2d_c4_bf16x6_f32(_dev(x), _dev(w6), w6.shape[3], _opt(bias), _dev(out), B, H, W, OH, OW, Cout, KH, KW,
                                             stride, pad, dil, out.stride(0), act, stream if stream is not None else stream_ptr()),
             'aot_conv2d_c4_bf16x6_f32')
        return out
    return conv2d(x, w, bias, out, H, W, 4, OH, OW, Cout, KH, KW, stride, pad, dil, act=act, B=B, stream=stream)


_x6k_ws = None


def x6_ksplit(M, Cout, K):
    """Split-K factor of the phase-shifted 128x128 kernel for a layer (1 = do not use it): long-K layers (K >= 2304: the 3x3
    convolutions on 256 channels) whose 128x128 tiles fill less than a third of the chip -- the stride-16 maps, and the stride-8 map
    one clip at a time -- get the largest split that keeps (tiles x slices) within one dispatch round of 256 workgroups with at least
    eight k-steps per slice (profiles/r05_x6pp.txt: l3.c2 3x3 at three lanes 60.6 -> 46.5 us, at one lane 48.6 -> 29.2; dec c8 at one
    lane 49.5 -> 39.5)."""
    nk = K // 32
    if K < 2304:
        # K = 1024 on the stride-16 map one lane at a time (the LSTT's linear2: 108 tiles of 64x64, 32 k-steps): two slices on the
        # 64x64 direct-weight kernel (negative = that kernel), 20.0 -> 17.4 us (profiles/r05_x6rd_splitk.txt)
        if K >= 1024 and nk % 2 == 0 and -(-M // 64) * -(-Cout // 64) <= 128 and 2 * M * Cout <= X6K_SCRATCH_FLOATS:
            return -2
        return 1
    nwide = -(-M // 128) * -(-Cout // 128)
    for ks in (9, 8, 6, 4, 3, 2):
        if nwide * ks <= 256 and nk % ks == 0 and nk // ks >= 8 and ks * M * Cout <= X6K_SCRATCH_FLOATS:
            return ks
    return 1


def conv2d_x6k(x, w, bias, out, H, W, Cin, OH, OW, Cout, KH=1, KW=1, stride=1, pad=0, dil=1, res=None, act=ACT_NONE, B=1,
               res_rows=0, ksplit=1, stream=None):
    """The phase-shifted 128x128 bf16x6 kernel with split-K over the grid (aot_conv2d_bf16x6k_f32); the slabs of the k-slices live in
    a per-stream scratch buffer that every such call on the stream shares (the calls are stream-ordered)."""
    global _x6k_ws
    w6 = getattr(w, '_aot_w6', None)
    if w6 is None:
        w6 = pack_bf16x6(w)
    scratch = None
    if abs(ksplit) > 1:       # (ksplit < 0: |ksplit| slices on the 64x64 register-staged kernel instead of the 128x128 phase-shifted one)
        if _x6k_ws is None:
            from networks.layers.workspace import Workspace
            _x6k_ws = Workspace()
        need = abs(ksplit) * B * OH * OW * Cout
        scratch = _x6k_ws.get('x6k', (max(need, X6K_SCRATCH_FLOATS),), x.device)
    _chk(load().aot_conv2d_bf16x6k_f32(_dev(x), _dev(w6), w6.shape[3], _opt(bias), _opt(res), _dev(out), B, H, W, Cin, OH, OW, Cout,
                                       KH, KW, stride, pad, dil, x.stride(0), out.stride(0), res.stride(0) if res is not None else 0,
                                       res_rows, act, ksplit, _opt(scratch), scratch.numel() if scratch is not None else 0,
                                       stream if stream is not None else stream_ptr()), 'aot_conv2d_bf16x6k_f32')
    return out


def linear_group(xs, ws, biases, outs, act=ACT_NONE, ress=None, res_rows=0, stream=None):
    """outs[g] = act(xs[g] @ ws[g] + biases[g] (+ ress[g])) for up to four linear layers of ONE shape (same M, K, N, leading dimensions).
    In a bf16x6 scope with the kernel choice left open: ONE launch, the problems side by side in the grid (aot_linear_group_bf16x6_f32);
    otherwise one linear() each.  AOT_NO_GROUP: always the latter (A/B runs)."""
    n = len(xs)
    M, K = xs[0].shape
    N = outs[0].shape[1]
    stack = _scopes.stack
    st = stream if stream is not None else stream_ptr()
    same = all(x.shape == xs[0].shape and x.stride(0) == xs[0].stride(0) for x in xs) and \
        all(o.shape == outs[0].shape and o.stride(0) == outs[0].stride(0) for o in outs) and \
        all(w.shape == ws[0].shape for w in ws) and (ress is None or all(r.stride(0) == ress[0].stride(0) for r in ress)) and \
        (all(b is None for b in biases) or all(b is not None for b in biases))
    if 1 < n <= 4 and same and stack and stack[-1][1] and X6_TILE == 0 and K % 32 == 0 and N > 32 and x6_ksplit(M, N, K) == 1 and \
            -(-M // 64) * -(-N // 64) >= X6_MIN_TILES and not os.environ.get('AOT_NO_GROUP'):
        w6s = [getattr(w, '_aot_w6', None) for w in ws]
        w6s = [pack_bf16x6(w) if w6 is None else w6 for w, w6 in zip(ws, w6s)]
        arr = ctypes.c_void_p * n
        pb = arr(*[_dev(b) for b in biases]) if biases[0] is not None else None
        pr = arr(*[_dev(r) for r in ress]) if ress is not None else None
        _chk(load().aot_linear_group_bf16x6_f32(n, arr(*[_dev(x) for x in xs]), arr(*[_dev(w6) for w6 in w6s]), w6s[0].shape[3], pb, pr,
                                                arr(*[_dev(o) for o in outs]), M, K, N, xs[0].stride(0), outs[0].stride(0),
                                                ress[0].stride(0) if ress is not None else 0, res_rows, act, st),
             'aot_linear_group_bf16x6_f32')
        return outs
    for g in range(n):
        linear(xs[g], ws[g], biases[g], outs[g], res=ress[g] if ress is not None else None, act=act, res_rows=res_rows, stream=st)
    return outs


def linear_ln_out(x, w, bias, out, gamma, beta, ln_out, eps=1e-5, res=None, res_rows=0, stream=None):
    """out = x @ w + bias (+ res) and ln_out = LayerNorm(out) * gamma + beta.  Where the layer runs split-K in a bf16x6 scope with 256
    output channels, the reduce launch writes both (aot_linear_bf16x6k_ln_f32; bit-identical to the separate LayerNorm launch);
    otherwise linear() + layernorm().  AOT_NO_LNO_FUSE: always the latter (A/B runs)."""
    global _x6k_ws
    M, K = x.shape
    N = out.shape[1]
    stack = _scopes.stack
    st = stream if stream is not None else stream_ptr()
    if stack and stack[-1][1] and X6_TILE == 0 and N == 256 and K % 32 == 0 and -(-M // 64) * 4 >= X6_MIN_TILES and \
            ln_out.stride(0) % 4 == 0 and ln_out.data_ptr() % 16 == 0 and gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0 and \
            not os.environ.get('AOT_NO_LNO_FUSE'):
        ks = x6_ksplit(M, N, K)
        if ks != 1:
            w6 = getattr(w, '_aot_w6', None)
            if w6 is None:
                w6 = pack_bf16x6(w)
            if _x6k_ws is None:
                from networks.layers.workspace import Workspace
                _x6k_ws = Workspace()
            scratch = _x6k_ws.get('x6k', (max(abs(ks) * M * N, X6K_SCRATCH_FLOATS),), x.device)
            _chk(load().aot_linear_bf16x6k_ln_f32(_dev(x), _dev(w6), w6.shape[3], _opt(bias), _opt(res), _dev(out), M, K, N, x.stride(0),
                                                  out.stride(0), res.stride(0) if res is not None else 0, res_rows, ACT_NONE, ks,
                                                  _dev(scratch), scratch.numel(), _dev(gamma), _dev(beta), _dev(ln_out), ln_out.stride(0),
                                                  eps, st), 'aot_linear_bf16x6k_ln_f32')
            return out
    linear(x, w, bias, out, res=res, res_rows=res_rows, stream=st)
    layernorm(out, gamma, beta, ln_out, eps=eps, stream=st)
    return out


def groupnorm_apply(x, stats, gamma, beta, out, groups, act=ACT_NONE, B=1, add=None, add_rows=0, stream=None):
    """The apply half of groupnorm() on finished statistics (stats [B][G][2] doubles)."""
    M = x.shape[0] // B
    _chk(load().aot_groupnorm_apply_f32(_dev(x), _dev(stats), _dev(gamma), _dev(beta), _dev(out), _opt(add), B, M, x.shape[1], groups,
                                        x.stride(0), out.stride(0), add.stride(0) if add is not None else 0, add_rows, act,
                                        stream if stream is not None else stream_ptr()), 'aot_groupnorm_apply_f32')
    return out


def pack_bf16(w_kn, stream=None):
    """w [K, N] fp32 (K % 32 == 0, unit column stride) -> one bf16 plane in the tile order of the matrix-core kernels, rounded to
    nearest even (aot_pack_bf16_f32): int16 [K/32, 4, cout_pad, 8]."""
    K, N = w_kn.shape
    if K % 32 or w_kn.stride(1) != 1:
        raise AotHipError('pack_bf16: w [K, N] with K %% 32 == 0 and unit column stride (got %s)' % (tuple(w_kn.shape),))
    cout_pad = (N + 63) // 64 * 64
    wq = torch.empty(K // 32, 4, cout_pad, 8, dtype=torch.int16, device=w_kn.device)
    _chk(load().aot_pack_bf16_f32(_dev(w_kn), _dev(wq), K, N, w_kn.stride(0), cout_pad, stream if stream is not None else stream_ptr()),
         'aot_pack_bf16_f32')
    return wq


def gemm_bf16_packed(a, wq, N, bias=None, out=None, stream=None, ks=1):
    """a [M, K] fp32 @ the packed weight of pack_bf16 (N real columns) (+ bias) -> fp32 [M, N].  ks > 1: split-K (K / 32 divisible
    by ks) through a scratch slab -- the weight gradients of the training path, whose K is the row count."""
    M, K = a.shape
    if a.stride(1) != 1 or wq.shape[0] * 32 != K:
        raise AotHipError('gemm_bf16_packed: a [M, K] against a weight packed for K = %d' % (wq.shape[0] * 32))
    if ks > 1 and (K // 32) % ks:
        raise AotHipError('gemm_bf16_packed: K / 32 = %d is not divisible by the split %d' % (K // 32, ks))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    scratch = torch.empty(ks * M * N, dtype=torch.float32, device=a.device) if ks > 1 else None
    _chk(load().aot_conv2d_bf16_f32(_dev(a), _dev(wq), wq.shape[2], _opt(bias), None, _dev(out), 1, 1, M, K, 1, M, N, 1, 1, 1, 0, 1,
                                    a.stride(0), out.stride(0), 0, 0, ACT_NONE, ks, _opt(scratch),
                                    stream if stream is not None else stream_ptr()),
         'aot_conv2d_bf16_f32')
    return out


def gemm_bf16(a, w_kn, bias=None, out=None, stream=None):
    """Training path, precision 'bf16': out [M, N] = a [M, K] @ w_kn [K, N] (+ bias) with both operands rounded to bf16 (round to
    nearest even) and fp32 accumulation (aot_pack_bf16_f32 + aot_conv2d_bf16_f32).  K % 32 == 0; a / out row-major fp32."""
    M, K = a.shape
    N = w_kn.shape[1]
    if K % 32 or w_kn.shape[0] != K or a.stride(1) != 1 or w_kn.stride(1) != 1:
        raise AotHipError('gemm_bf16: a [M, K] @ w [K, N] with K %% 32 == 0, unit inner strides (got %s @ %s)' % (tuple(a.shape), tuple(w_kn.shape)))
    cout_pad = (N + 63) // 64 * 64
    wq = torch.empty(K // 32, 4, cout_pad, 8, dtype=torch.int16, device=a.device)
    st = stream if stream is not None else stream_ptr()
    _chk(load().aot_pack_bf16_f32(_dev(w_kn), _dev(wq), K, N, w_kn.stride(0), cout_pad, st), 'aot_pack_bf16_f32')
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _chk(load().aot_conv2d_bf16_f32(_dev(a), _dev(wq), cout_pad, _opt(bias), None, _dev(out), 1, 1, M, K, 1, M, N, 1, 1, 1, 0, 1,
                                    a.stride(0), out.stride(0), 0, 0, ACT_NONE, 1, None, st), 'aot_conv2d_bf16_f32')
    return out


def conv2d_cfg(x, w, bias, out, H, W, Cin, OH, OW, Cout, KH=1, KW=1, stride=1, pad=0, dil=1, res=None, act=ACT_NONE,
               cfg=-1, wt=None, scratch=None, B=1, res_rows=0, stream=None):
    """Tuning / test form: explicit kernel configuration, explicit k-contiguous weight and split-K scratch."""
    _chk(load().aot_conv2d_nhwc_f32(_dev(x), _opt(w), _opt(wt), _opt(bias), _opt(res), _dev(out), _opt(scratch),
                                    scratch.numel() if scratch is not None else 0, B, H, W, Cin, OH, OW, Cout, KH, KW,
                                    stride, pad, dil, x.stride(0), w.stride(0) if w is not None else 0,
                                    wt.stride(0) if wt is not None else 0,
                                    out.stride(0), res.stride(0) if res is not None else 0, res_rows, act, cfg,
                                    stream if stream is not None else stream_ptr()), 'aot_conv2d_nhwc_f32')
    return out


def linear(x, w, bias, out, res=None, act=ACT_NONE, res_rows=0, stream=None):
    """out[M, N] = act(x[M, K] @ w[K, N] + bias (+ res))."""
    M, K = x.shape
    return conv2d(x, w, bias, out, 1, M, K, 1, M, out.shape[1], res=res, act=act, res_rows=res_rows, stream=stream)


def dwconv2d(x, w, bias, out, H, W, C, OH, OW, K, stride=1, pad=0, dil=1, act=ACT_NONE, B=1, stream=None):
    _chk(load().aot_dwconv2d_nhwc_f32(_dev(x), _dev(w), _opt(bias), _dev(out), B, H, W, C, OH, OW, K, K, stride, pad, dil,
                                      act, stream if stream is not None else stream_ptr()), 'aot_dwconv2d_nhwc_f32')
    return out


def maxpool3x3s2(x, out, H, W, C, OH, OW, stream=None):
    _chk(load().aot_maxpool3x3s2_nhwc_f32(_dev(x), _dev(out), H, W, C, OH, OW,
                                          stream if stream is not None else stream_ptr()), 'aot_maxpool3x3s2_nhwc_f32')
    return out


def nchw_to_nhwc(x, out, C, H, W, Cpad, stream=None):
    _chk(load().aot_nchw_to_nhwc_f32(_dev(x), _dev(out), C, H, W, Cpad, stream if stream is not None else stream_ptr()),
         'aot_nchw_to_nhwc_f32')
    return out


def nhwc_to_nchw(x, out, C, H, W, stream=None):
    _chk(load().aot_nhwc_to_nchw_f32(_dev(x), _dev(out), C, H, W, x.stride(0),
                                     stream if stream is not None else stream_ptr()), 'aot_nhwc_to_nchw_f32')
    return out


def layernorm(x, gamma, beta, out, add=None, out2=None, eps=1e-5, add_rows=0, stream=None):
    M, C = x.shape
    _chk(load().aot_layernorm_f32(_dev(x), _dev(gamma), _dev(beta), _dev(out), _opt(add), _opt(out2), M, C, x.stride(0),
                                  out.stride(0), add.stride(0) if add is not None else 0,
                                  out2.stride(0) if out2 is not None else 0, add_rows, eps,
                                  stream if stream is not None else stream_ptr()), 'aot_layernorm_f32')
    return out


def gn_buffers(ws, device, B, groups, nsplit):
    """(scratch, stats, ticket) for aot_groupnorm_stats_f32 out of a Workspace; the ticket words start at zero and the
    kernel leaves them zero."""
    scratch = ws.get('gn_scratch', (B * groups * nsplit * 2,), device, torch.float64)
    stats = ws.get('gn_stats', (B * groups * 2,), device, torch.float64)
    ticket = ws.get_zeroed('gn_ticket', (B * groups,), device, torch.int32)
    return scratch, stats, ticket


def groupnorm_stats(x, groups, bufs, B=1, eps=1e-5, nsplit=32, stream=None):
    """x [B*M, C] -> stats [B][G][2] (mean, rstd), one launch."""
    scratch, stats, ticket = bufs
    M = x.shape[0] // B
    _chk(load().aot_groupnorm_stats_f32(_dev(x), _dev(scratch), _dev(stats), _dev(ticket), B, M, x.shape[1], groups,
                                        x.stride(0), eps, nsplit, stream if stream is not None else stream_ptr()),
         'aot_groupnorm_stats_f32')
    return stats


def groupnorm(x, gamma, beta, out, groups, bufs, act=ACT_NONE, eps=1e-5, nsplit=32, B=1, add=None, add_rows=0,
              stream=None):
    """out = act(GroupNorm(x)) (+ add[row % add_rows]); x [B*M, C], statistics per (lane, group).  Two launches."""
    s = stream if stream is not None else stream_ptr()
    stats = groupnorm_stats(x, groups, bufs, B=B, eps=eps, nsplit=nsplit, stream=s)
    M = x.shape[0] // B
    _chk(load().aot_groupnorm_apply_f32(_dev(x), _dev(stats), _dev(gamma), _dev(beta), _dev(out), _opt(add), B, M,
                                        x.shape[1], groups, x.stride(0), out.stride(0),
                                        add.stride(0) if add is not None else 0, add_rows, act, s),
         'aot_groupnorm_apply_f32')
    return out


def gn_act_dwconv5(x, gamma, beta, w, out, groups, bufs, h, wd, act=ACT_GELU, eps=1e-5, nsplit=32, B=1, stream=None):
    """GNActDWConv2d (basic.py:15-35) in two launches: statistics, then GroupNorm-apply + activation + 5x5 depthwise conv."""
    s = stream if stream is not None else stream_ptr()
    stats = groupnorm_stats(x, groups, bufs, B=B, eps=eps, nsplit=nsplit, stream=s)
    _chk(load().aot_gn_act_dwconv5_f32(_dev(x), _dev(stats), _dev(gamma), _dev(beta), _dev(w), _dev(out), B, h, wd, x.shape[1],
                                       groups, x.stride(0), out.stride(0), act, s), 'aot_gn_act_dwconv5_f32')
    return out


def attention(q, k, v, out, T, H, scale_div, part=None, nsplit=1, T_dev=None, B=1, kv_brows=0, stream=None):
    """q/out [B*Nq, .]; lane b attends to rows [b*kv_brows, b*kv_brows + T) of k/v."""
    s = stream if stream is not None else stream_ptr()
    lib = load()
    nq = q.shape[0] // B
    _chk(lib.aot_attn_f32(_dev(q), _dev(k), _dev(v), _dev(out), _opt(part), B, kv_brows, nq, T, _opt(T_dev), H, 32,
                          q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale_div, nsplit, s), 'aot_attn_f32')
    if nsplit > 1:
        _chk(lib.aot_attn_merge_f32(_dev(part), None, _dev(out), q.shape[0], H, H * 32, 0, out.stride(0), nsplit, s),
             'aot_attn_merge_f32')
    return out


def x6_bank(B, rows, C, device):
    """Packed (pre-split, tile-major) K / V bank of the bf16x6 attention kernel for B lanes of up to `rows` memorised rows:
    (planes, cap_rows).  Zero-filled: rows past the bank length are multiplied by weights that are exactly 0."""
    cap = (rows + 31) // 32 * 32
    return torch.zeros(B * cap * C * 6, dtype=torch.int16, device=device), cap


def attention_pack_x6(k, v, bank, rows, B=1, src_brows=None, slot=0, slot_dev=None, stream=None):
    """Splits rows [b*src_brows, + rows) of k / v (token-major fp32, C columns) into the three bf16 planes of lane b's packed
    bank at rows slot*rows .. (aot_attn_pack_x6_f32; slot_dev: device int32 overriding `slot`, for graph replay)."""
    kv, cap = bank
    _chk(load().aot_attn_pack_x6_f32(_dev(k), _dev(v), _dev(kv), B, rows, k.shape[1], rows if src_brows is None else src_brows,
                                     k.stride(0), v.stride(0), cap, _opt(slot_dev), slot,
                                     stream if stream is not None else stream_ptr()), 'aot_attn_pack_x6_f32')
    return bank


def attention_x6(q, bank, out, T, H, scale_div, part=None, nsplit=1, T_dev=None, B=1, stream=None):
    """aot_hip.attention on a packed bank: the bf16x6 member (fp32-equivalent arithmetic on the bf16 matrix cores)."""
    kv, cap = bank
    s = stream if stream is not None else stream_ptr()
    lib = load()
    nq = q.shape[0] // B
    _chk(lib.aot_attn_x6_f32(_dev(q), _dev(kv), _dev(out), _opt(part), B, cap, nq, T, _opt(T_dev), H, 32, q.stride(0),
                             out.stride(0), scale_div, nsplit, s), 'aot_attn_x6_f32')
    if nsplit > 1:
        _chk(lib.aot_attn_merge_f32(_dev(part), None, _dev(out), q.shape[0], H, H * 32, 0, out.stride(0), nsplit, s),
             'aot_attn_merge_f32')
    return out


def x6_gated_bank(B, rows, dqk, dv, device):
    """Packed bank of the gated (DeAOT) bf16x6 attention kernel: (K planes, V planes, cap_rows), zero-filled."""
    cap = (rows + 31) // 32 * 32
    return (torch.zeros(B * cap * dqk * 3, dtype=torch.int16, device=device),
            torch.zeros(B * cap * dv * 3, dtype=torch.int16, device=device), cap)


def gated_pack_x6(k, v, bank, rows, B=1, src_brows=None, slot=0, slot_dev=None, stream=None):
    """Rows [b*src_brows, + rows) of k [., 128] / v [., 1024] (either may be None) into lane b's packed gated bank at rows
    slot*rows .. (aot_attn_pack_x6_part_f32, K-style / V-style)."""
    kp, vp, cap = bank
    s = stream if stream is not None else stream_ptr()
    for x, planes, tr in ((k, kp, 0), (v, vp, 1)):
        if x is not None:
            _chk(load().aot_attn_pack_x6_part_f32(_dev(x), _dev(planes), B, rows, x.shape[1], rows if src_brows is None else src_brows,
                                                  x.stride(0), cap, _opt(slot_dev), slot, tr, s), 'aot_attn_pack_x6_part_f32')
    return bank


def gated_attention_x6(q, bank, gate, out, T, scale_div, part=None, nsplit=1, T_dev=None, B=1, stream=None):
    """aot_hip.gated_attention on a packed bank: the bf16x6 member (dqk = 128, dv = 1024)."""
    kp, vp, cap = bank
    s = stream if stream is not None else stream_ptr()
    lib = load()
    dv = out.shape[1]
    nq = q.shape[0] // B
    _chk(lib.aot_gated_attn_x6_f32(_dev(q), _dev(kp), _dev(vp), _opt(gate), _dev(out), _opt(part), B, cap, nq, T, _opt(T_dev),
                                   q.shape[1], dv, q.stride(0), gate.stride(0) if gate is not None else 0, out.stride(0),
                                   scale_div, nsplit, s), 'aot_gated_attn_x6_f32')
    if nsplit > 1:
        _chk(lib.aot_attn_merge_f32(_dev(part), _opt(gate), _dev(out), q.shape[0], dv // 256, dv,
                                    gate.stride(0) if gate is not None else 0, out.stride(0), nsplit, s), 'aot_attn_merge_f32')
    return out


def attention_topk(q, k, v, out, T, H, scale_div, top_k, scores, stream=None):
    """Top-k sparse attention (MultiheadAttention top_k > 0): scores is scratch of H*Nq*((T+3)&~3) floats."""
    _chk(load().aot_attn_topk_f32(_dev(q), _dev(k), _dev(v), _dev(out), _dev(scores), q.shape[0], T, H, 32, q.stride(0),
                                  k.stride(0), v.stride(0), out.stride(0), scale_div, top_k,
                                  stream if stream is not None else stream_ptr()), 'aot_attn_topk_f32')
    return out


def gated_attention_topk(q, k, v, gate, out, T, scale_div, top_k, scores, stream=None):
    """Top-k sparse gated attention (GatedPropagation top_k > 0): q [Nq,128], k [T,128], v [T,dv], gate/out [Nq,dv];
    scores is scratch of Nq*((T+3)&~3) floats."""
    _chk(load().aot_gated_attn_topk_f32(_dev(q), _dev(k), _dev(v), _opt(gate), _dev(out), _dev(scores), q.shape[0], T,
                                        q.shape[1], out.shape[1], q.stride(0), k.stride(0), v.stride(0),
                                        gate.stride(0) if gate is not None else 0, out.stride(0), scale_div, top_k,
                                        stream if stream is not None else stream_ptr()), 'aot_gated_attn_topk_f32')
    return out


def gated_attention(q, k, v, gate, out, T, scale_div, part=None, nsplit=1, T_dev=None, B=1, kv_brows=0, stream=None):
    """DeAOT single-head attention with a wide value: q [B*Nq,128], k [.,128], v [.,dv], gate/out [B*Nq,dv]."""
    s = stream if stream is not None else stream_ptr()
    lib = load()
    dv = out.shape[1]
    nq = q.shape[0] // B
    _chk(lib.aot_gated_attn_f32(_dev(q), _dev(k), _dev(v), _opt(gate), _dev(out), _opt(part), B, kv_brows, nq, T,
                                _opt(T_dev), q.shape[1], dv, q.stride(0), k.stride(0), v.stride(0),
                                gate.stride(0) if gate is not None else 0, out.stride(0), scale_div, nsplit, s),
         'aot_gated_attn_f32')
    if nsplit > 1:
        _chk(lib.aot_attn_merge_f32(_dev(part), _opt(gate), _dev(out), q.shape[0], dv // 256, dv,
                                    gate.stride(0) if gate is not None else 0, out.stride(0), nsplit, s),
             'aot_attn_merge_f32')
    return out


def pack_local_tables(relk_weight, relk_bias, relv, heads, max_dis=7):
    """relative_emb_k.weight [H*W2, d(,1,1)], .bias [H*W2], relative_emb_v [H, d, W2] -> the scalar-load
    layouts of aot_local_attn_f32: ([H,WS,d,16], [H,WS,16], [H,WS,d,16])."""
    ws = 2 * max_dis + 1
    d = relv.shape[1]
    # key table pre-multiplied by sqrt(d): the kernel keeps only q/sqrt(d) in registers
    wk = (relk_weight.detach().double().reshape(heads, ws, ws, d) * (float(d) ** 0.5)).float().permute(0, 1, 3, 2)  # [H, dy, c, dx]
    rv = relv.detach().float().reshape(heads, d, ws, ws).permute(0, 2, 1, 3)            # [H, dy, c, dx]
    bk = relk_bias.detach().float().reshape(heads, ws, ws)
    pad = lambda t: torch.nn.functional.pad(t, (0, 16 - ws)).contiguous()
    return pad(wk), pad(bk), pad(rv)


def local_attention(q, k, v, relk_w, relk_b, relv_t, out, h, w, H, scale_div, max_dis=7, B=1, kv_brows=0, stream=None):
    _chk(load().aot_local_attn_f32(_dev(q), _dev(k), _dev(v), _dev(relk_w), _dev(relk_b), _dev(relv_t), _dev(out), B,
                                   kv_brows if kv_brows else h * w, h, w, H, 32, max_dis, q.stride(0), k.stride(0),
                                   v.stride(0), out.stride(0), scale_div,
                                   stream if stream is not None else stream_ptr()), 'aot_local_attn_f32')
    return out


def local_gated(q, k, v, gate, relk_t, relk_b, prob, out, h, w, scale_div, max_dis=7, B=1, kv_brows=0, stream=None):
    _chk(load().aot_local_gated_f32(_dev(q), _dev(k), _dev(v), _opt(gate), _dev(relk_t), _dev(relk_b), _dev(prob),
                                    _dev(out), B, kv_brows if kv_brows else h * w, h, w, q.shape[1], out.shape[1], max_dis,
                                    q.stride(0), k.stride(0), v.stride(0), gate.stride(0) if gate is not None else 0,
                                    out.stride(0), scale_div, stream if stream is not None else stream_ptr()),
         'aot_local_gated_f32')
    return out


def swin_window_attention(qkv, qkv_bias, table, out, H, W, C, nH, shift, scale, B=1, stream=None):
    """qkv / out hold B images of H x W tokens stacked along the rows: one launch for the batch."""
    _chk(load().aot_swin_window_attn_f32(_dev(qkv), _dev(qkv_bias), _dev(table), _dev(out), B, H, W, C, nH, 7, shift,
                                         qkv.stride(0), out.stride(0), scale,
                                         stream if stream is not None else stream_ptr()), 'aot_swin_window_attn_f32')
    return out


def patch_merge(x, out, H, W, C, stream=None):
    _chk(load().aot_patch_merge_f32(_dev(x), _dev(out), H, W, C, x.stride(0),
                                    stream if stream is not None else stream_ptr()), 'aot_patch_merge_f32')
    return out


def idbank(mask, table, bias, out, H, W, OH, OW, K, stride, pad, C, nlabel, sumtab=None, G=1, group_size=0, group0=0,
           fuse=None, stream=None):
    """mask [H, W] label ids -> id embedding rows [G*OH*OW, C]; lane g = object group group0+g (its labels -> 1..group_size).
    fuse = [(add_i, out_i), ...] (<= 4): out_i = id_emb + add_i in the same launch (V + id_emb of every LSTT layer)."""
    n = len(fuse) if fuse else 0
    if n:
        adds = (ctypes.c_void_p * n)(*[_dev(a) for a, _ in fuse])
        outs = (ctypes.c_void_p * n)(*[_dev(o) for _, o in fuse])
        ldadd, ldfout = fuse[0][0].stride(0), fuse[0][1].stride(0)
        if any(a.stride(0) != ldadd or o.stride(0) != ldfout for a, o in fuse):
            raise AotHipError('fused id-bank outputs must share their row strides')
    else:
        adds = outs = None
        ldadd = ldfout = 0
    _chk(load().aot_idbank_f32(_dev(mask), _dev(table), _opt(sumtab), _opt(bias), _opt(out), G, group_size, group0, H, W, OH, OW, K,
                               stride, pad, C, nlabel, out.stride(0) if out is not None else C, adds, outs, n, ldadd, ldfout,
                               stream if stream is not None else stream_ptr()), 'aot_idbank_f32')
    return out


def bilinear(x, out, IH, IW, OH, OW, C, align_corners, add=None, B=1, add_shared=False, stream=None):
    _chk(load().aot_bilinear_nhwc_f32(_dev(x), _opt(add), _dev(out), B, IH, IW, OH, OW, C, x.stride(0),
                                      add.stride(0) if add is not None else 0, out.stride(0), int(align_corners),
                                      int(bool(add_shared)), stream if stream is not None else stream_ptr()),
         'aot_bilinear_nhwc_f32')
    return out


def gn_bilinear(x, stats, gamma, beta, out, IH, IW, OH, OW, C, groups, align_corners, act=ACT_NONE, add=None, B=1, add_shared=False,
                stream=None):
    """out = bilinear(act(GroupNorm(x))) (+ add) with finished statistics (aot_gn_bilinear_nhwc_f32): groupnorm_apply + bilinear in
    one launch, bit-identical to the pair."""
    _chk(load().aot_gn_bilinear_nhwc_f32(_dev(x), _dev(stats), _dev(gamma), _dev(beta), _opt(add), _dev(out), B, IH, IW, OH, OW, C, groups,
                                         x.stride(0), add.stride(0) if add is not None else 0, out.stride(0), int(align_corners),
                                         int(bool(add_shared)), act, stream if stream is not None else stream_ptr()),
         'aot_gn_bilinear_nhwc_f32')
    return out


def gn_conv1x1(x, stats, gamma, beta, w, bias, out, groups, cout, gn_act=ACT_NONE, act=ACT_NONE, B=1, stream=None):
    """out[:, :cout] = act(gn_act(GroupNorm(x)) @ w + bias) for cout <= 32 with finished statistics (aot_gn_conv1x1_f32): groupnorm_apply
    + the 1x1 convolution in one launch, bit-identical to the pair (fp32 matrix cores, cfg 3)."""
    M = x.shape[0] // B
    _chk(load().aot_gn_conv1x1_f32(_dev(x), _dev(stats), _dev(gamma), _dev(beta), _dev(w), _opt(bias), _dev(out), B, M, x.shape[1], cout,
                                   groups, x.stride(0), w.stride(0), out.stride(0), gn_act, act,
                                   stream if stream is not None else stream_ptr()), 'aot_gn_conv1x1_f32')
    return out


def logits_finalize(logits, out4, out, IH, IW, C, OH, OW, obj_total, align_corners, G=1, stream=None):
    """logits [G*IH*IW, ld]: out4 [G, C, IH, IW] planar masked copy, out [C, OH, OW] (G = 1) or the soft aggregation of the
    G object groups [1 + G*(C-1), OH, OW]; obj_total = objects in the frame (group g holds ids g*(C-1)+1 ..)."""
    _chk(load().aot_logits_finalize_f32(_dev(logits), _opt(out4), _opt(out), G, IH, IW, C, logits.stride(0), OH, OW,
                                        obj_total, int(align_corners), stream if stream is not None else stream_ptr()),
         'aot_logits_finalize_f32')


def x6_gn_fusable(M, K, Cout, B=1):
    """True where linear_gn_x6 + gn_act_dwconv5_part may replace linear + groupnorm_stats + gn_act_dwconv5: a bf16x6 scope with the
    kernel choice left open, one lane, 32-channel groups on a layer the family takes anyway."""
    stack = _scopes.stack
    return bool(stack and stack[-1][1] and X6_TILE == 0 and B == 1 and K % 32 == 0 and Cout % 32 == 0 and
                -(-M // 64) * -(-Cout // 64) >= X6_MIN_TILES)


def linear_gn_x6(x, w, bias, out, part, res=None, act=ACT_NONE, res_rows=0, stream=None):
    """out[M, N] = act(x @ w + bias (+ res)) on the bf16x6 family, and the GroupNorm partial sums of out (32-channel groups) into
    `part` [2 * ceil(M / 64) * N / 32 * 2] floats from the same tile end (aot_linear_gn_bf16x6_f32).  Returns P, the partial rows."""
    M, K = x.shape
    N = out.shape[1]
    w6 = getattr(w, '_aot_w6', None)
    if w6 is None:
        w6 = pack_bf16x6(w)
    _chk(load().aot_linear_gn_bf16x6_f32(_dev(x), _dev(w6), w6.shape[3], _opt(bias), _opt(res), _dev(out), M, K, N, x.stride(0),
                                         out.stride(0), res.stride(0) if res is not None else 0, res_rows, act, _dev(part), part.numel(),
                                         stream if stream is not None else stream_ptr()), 'aot_linear_gn_bf16x6_f32')
    return 2 * (-(-M // 64))


def fold_layernorm(w, bias, gamma, beta):
    """The affine half of LayerNorm folded into the linear layer behind it: (n * gamma + beta) W + b = n (diag(gamma) W) + (beta W + b)
    for the row-normalised n.  w [K, N] packed k-major; returns (W' registered for the bf16x6 packer, b'); beta W in double.  W' carries
    its column sums (`_aot_colsum`, summed in double): the kernel's mean correction, (x - mean) W' = (x - c) W' - (mean - c) colsum."""
    wf = _register_weight((gamma.detach().float()[:, None] * w).contiguous())
    wf._aot_colsum = wf.double().sum(0).float().contiguous()
    bf = beta.detach().double() @ w.double()
    if bias is not None:
        bf = bf + bias.double()
    return wf, bf.float().contiguous()


def x6_ln_fusable(M, K, Cout):
    """True where layernorm_linear_x6 may replace layernorm + linear: a bf16x6 scope with the kernel choice left open, on a layer the
    family takes anyway (AOT_NO_LN_FUSE: the two-launch form, for A/B runs)."""
    stack = _scopes.stack
    return bool(stack and stack[-1][1] and X6_TILE == 0 and K % 32 == 0 and K <= 2048 and Cout > 32 and
                -(-M // 64) * -(-Cout // 64) >= X6_MIN_TILES and not os.environ.get('AOT_NO_LN_FUSE'))


def layernorm_linear_x6(x, wf, bf, out, eps=1e-5, res=None, act=ACT_NONE, res_rows=0, gn_part=None, stream=None):
    """out[M, N] = act(LayerNorm(x) @ w + b (+ res)) in one launch (aot_layernorm_linear_bf16x6_f32); (wf, bf) = fold_layernorm(w, b,
    gamma, beta).  gn_part: also the GroupNorm partials of out, as linear_gn_x6 -- returns their row count P then, else out."""
    M, K = x.shape
    N = out.shape[1]
    w6 = getattr(wf, '_aot_w6', None)
    if w6 is None:
        w6 = pack_bf16x6(wf)
    _chk(load().aot_layernorm_linear_bf16x6_f32(_dev(x), _dev(w6), w6.shape[3], _opt(bf), _dev(wf._aot_colsum), _opt(res), _dev(out), M, K, N, x.stride(0),
                                                out.stride(0), res.stride(0) if res is not None else 0, res_rows, act, eps,
                                                _opt(gn_part), gn_part.numel() if gn_part is not None else 0,
                                                stream if stream is not None else stream_ptr()), 'aot_layernorm_linear_bf16x6_f32')
    return out if gn_part is None else 2 * (-(-M // 64))


def gn_act_dwconv5_part(x, gamma, beta, w, out, groups, part, P, H, W, act=ACT_GELU, eps=1e-5, stream=None):
    """gn_act_dwconv5 with the statistics taken from the producing GEMM's partial sums (linear_gn_x6; one lane)."""
    _chk(load().aot_gn_act_dwconv5p_f32(_dev(x), _dev(part), P, _dev(gamma), _dev(beta), _dev(w), _dev(out), H, W, x.shape[1], groups,
                                        x.stride(0), out.stride(0), act, eps, stream if stream is not None else stream_ptr()),
         'aot_gn_act_dwconv5p_f32')
    return out


def frame_tail(logits, out4, label_out, label_in, IH, IW, C, obj_total, align_corners, stream=None):
    """logits [IH*IW, ld] (one object group) -> label_out [1,1,OH,OW] (argmax of the softmax of the resized, masked logits), label_in
    [1,1,LH,LW] | None (nearest resize of label_out: the memory update's mask), out4 [1,C,IH,IW] | None: aot_frame_tail_f32."""
    OH, OW = label_out.shape[-2:]
    LH, LW = label_in.shape[-2:] if label_in is not None else (0, 0)
    _chk(load().aot_frame_tail_f32(_dev(logits), _opt(out4), _dev(label_out), _opt(label_in), IH, IW, C, logits.stride(0), OH, OW,
                                   LH, LW, obj_total, int(align_corners), stream if stream is not None else stream_ptr()),
         'aot_frame_tail_f32')
    return label_out, label_in


def add(a, b, out, stream=None):
    _chk(load().aot_add_f32(_dev(a), _dev(b), _dev(out), a.numel(), stream if stream is not None else stream_ptr()),
         'aot_add_f32')
    return out


def copy_rows(src, dst, rows, B=1, src_brows=None, dst_brows=0, slot=0, slot_dev=None, stream=None):
    """dst[b*dst_brows + slot*rows + r] = src[b*src_brows + r] for r < rows (token-major 2-D tensors, src.shape[1] columns);
    src_brows = 0 broadcasts one block to every lane; slot_dev: device int32 overriding `slot` (graph replay)."""
    _chk(load().aot_copy_rows_f32(_dev(src), _dev(dst), B, rows, src.shape[1], rows if src_brows is None else src_brows,
                                  dst_brows, src.stride(0), dst.stride(0), _opt(slot_dev), slot,
                                  stream if stream is not None else stream_ptr()), 'aot_copy_rows_f32')
    return dst


_MEAN3 = (ctypes.c_double * 3)(0.485, 0.456, 0.406)
_STD3 = (ctypes.c_double * 3)(0.229, 0.224, 0.225)


def preprocess(img, out_h, out_w, flip=False, out=None, stream=None):
    """img [H, W, 3] uint8 or float32 (values 0..255) on the device -> normalised engine input [1, 3, out_h, out_w]
    (MultiRestrictSize's cubic resize + flip and MultiToTensor, video_transforms.py:655-711)."""
    if img.dim() != 3 or img.shape[2] != 3 or img.stride(2) != 1 or img.stride(1) != 3:
        raise AotHipError('preprocess expects an interleaved [H, W, 3] image')
    if img.dtype not in (torch.uint8, torch.float32):
        raise AotHipError('preprocess expects uint8 or float32')
    if out is None:
        out = torch.empty(1, 3, out_h, out_w, dtype=torch.float32, device=img.device)
    _chk(load().aot_preprocess_f32(_dev(img), int(img.dtype == torch.uint8), img.shape[0], img.shape[1], img.stride(0),
                                   _dev(out), out_h, out_w, int(bool(flip)), _MEAN3, _STD3,
                                   stream if stream is not None else stream_ptr()), 'aot_preprocess_f32')
    return out


def fuse_probs(logits, flips, new_label=None, want_aug_labels=True, want_prob=False, stream=None):
    """logits [A, nc, H, W] (one row per augmentation, decoded at the original size), flips: list of A bools.
    Returns (fused_label [1,1,H,W], aug_labels [A,1,H,W] | None, fused_prob [1,nc,H,W] | None) -- evaluator.py:325-372."""
    A, nc, H, W = logits.shape
    dev = logits.device
    fused = torch.empty(1, 1, H, W, dtype=torch.float32, device=dev)
    augl = torch.empty(A, 1, H, W, dtype=torch.float32, device=dev) if want_aug_labels else None
    prob = torch.empty(1, nc, H, W, dtype=torch.float32, device=dev) if want_prob else None
    mask = sum(1 << i for i, f in enumerate(flips) if f)
    _chk(load().aot_fuse_probs_f32(_dev(logits.contiguous()), _opt(new_label), _dev(fused), _opt(augl), _opt(prob), A, nc, H, W,
                                   mask, stream if stream is not None else stream_ptr()), 'aot_fuse_probs_f32')
    return fused, augl, prob


def label_resize(label, out_h, out_w, flip=False, stream=None):
    """label [..., H, W] float -> [1, 1, out_h, out_w]: flip_tensor(label, 3) then F.interpolate(mode='nearest')."""
    H, W = label.shape[-2:]
    out = torch.empty(1, 1, out_h, out_w, dtype=torch.float32, device=label.device)
    _chk(load().aot_label_resize_f32(_dev(label.contiguous()), _dev(out), H, W, out_h, out_w, int(bool(flip)),
                                     stream if stream is not None else stream_ptr()), 'aot_label_resize_f32')
    return out


# ---- training-side stages (csrc/train_ops.hip; SURVEY 8f4 first slice) ------------------------------------------------
def ce_loss(logits, labels, top_k=0, stream=None):
    """logits [B,C,H,W] (contiguous), labels [B,H,W] fp32 ids (255 = ignore) -> (loss [B], saved state for ce_loss_bwd).
    top_k > 0: mean of the top_k largest per-pixel losses of each sample; 0: mean over the valid pixels."""
    B, C = logits.shape[:2]
    P = logits[0, 0].numel()
    dev = logits.device
    loss_px = torch.empty(B, P, dtype=torch.float32, device=dev)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    thr = torch.empty(2 * B, dtype=torch.int32, device=dev) if top_k > 0 else None      # [order key | share of the ties] per sample
    cnt = torch.empty(B, dtype=torch.float32, device=dev) if top_k <= 0 else None
    _chk(load().aot_ce_loss_f32(_dev(logits), _dev(labels), _dev(loss_px), _dev(loss), _opt(thr), _opt(cnt), B, C, P, int(top_k),
                                stream if stream is not None else stream_ptr()), 'aot_ce_loss_f32')
    return loss, (loss_px, thr, cnt)


def ce_loss_bwd(logits, labels, saved, gscale, stream=None):
    """gscale [B] = upstream gradient / k (or / number of valid pixels) -> grad of the logits."""
    loss_px, thr, _ = saved
    B, C = logits.shape[:2]
    grad = torch.empty_like(logits)
    _chk(load().aot_ce_loss_bwd_f32(_dev(logits), _dev(labels), _dev(loss_px), _opt(thr), _dev(gscale), _dev(grad), B, C,
                                    logits[0, 0].numel(), stream if stream is not None else stream_ptr()), 'aot_ce_loss_bwd_f32')
    return grad


def soft_jaccard(logits, labels, eps=1e-6, nchunk=64, stream=None):
    B, C = logits.shape[:2]
    P = logits[0, 0].numel()
    dev = logits.device
    part = torch.empty(B * nchunk * 16 * 3, dtype=torch.float64, device=dev)
    sums = torch.empty(B, 16, 3, dtype=torch.float64, device=dev)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    _chk(load().aot_soft_jaccard_f32(_dev(logits), _dev(labels), _dev(part), _dev(sums), _dev(loss), B, C, P, nchunk, eps,
                                     stream if stream is not None else stream_ptr()), 'aot_soft_jaccard_f32')
    return loss, sums


def soft_jaccard_bwd(logits, labels, sums, gout, eps=1e-6, stream=None):
    B, C = logits.shape[:2]
    grad = torch.empty_like(logits)
    _chk(load().aot_soft_jaccard_bwd_f32(_dev(logits), _dev(labels), _dev(sums), _dev(gout), _dev(grad), B, C,
                                         logits[0, 0].numel(), eps, stream if stream is not None else stream_ptr()),
         'aot_soft_jaccard_bwd_f32')
    return grad


def _flat_f32(*tensors):
    """The optimiser / EMA / norm kernels walk raw memory: fp32 and contiguous, or they would corrupt the tensor silently."""
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise AotHipError('expected a contiguous float32 tensor, got %s, contiguous=%s' % (t.dtype, t.is_contiguous()))


def adamw_step(p, g, m, v, lr, weight_decay, beta1, beta2, eps, step, gscale=1.0, stream=None):
    _flat_f32(p, g, m, v)
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise AotHipError('adamw_step: parameter, gradient and moments differ in size')
    _chk(load().aot_adamw_step_f32(_dev(p), _dev(g), _dev(m), _dev(v), p.numel(), lr, weight_decay, beta1, beta2, eps, int(step),
                                   gscale, stream if stream is not None else stream_ptr()), 'aot_adamw_step_f32')


def ema_update(shadow, param, one_minus_decay, stream=None):
    _flat_f32(shadow, param)
    if shadow.numel() != param.numel():
        raise AotHipError('ema_update: shadow and parameter differ in size')
    _chk(load().aot_ema_update_f32(_dev(shadow), _dev(param), shadow.numel(), one_minus_decay,
                                   stream if stream is not None else stream_ptr()), 'aot_ema_update_f32')


def sumsq_flat(x, scratch, out, stream=None):
    """out (one fp64 element) = sum(x^2) over a flat fp32 buffer in one launch; scratch = (partials fp64 [nblk], ticket int32 [1],
    zeroed once)."""
    _flat_f32(x)
    part, ticket = scratch
    if out.dtype != torch.float64 or part.dtype != torch.float64 or ticket.dtype != torch.int32:
        raise AotHipError('sumsq_flat: fp64 partials / accumulator and an int32 ticket')
    nblk = min(part.numel(), max(1, (x.numel() + 4095) // 4096))
    _chk(load().aot_sumsq_flat_f64(_dev(x), x.numel(), _dev(part), nblk, _dev(ticket), _dev(out),
                                   stream if stream is not None else stream_ptr()), 'aot_sumsq_flat_f64')


def adamw_flat(p, g, m, v, seg_off, hyp, beta1, beta2, eps, sumsq=None, max_norm=0.0, stream=None):
    """torch.optim.AdamW over flat buffers: seg_off int64 [nseg + 1] (device), hyp fp32 [nseg, 4] (device) = lr (< 0: skip), weight
    decay, 1 - beta1^step, sqrt(1 - beta2^step) per tensor; sumsq (device fp64) + max_norm: the gradient clip, applied in-kernel."""
    _flat_f32(p, g, m, v, hyp)
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise AotHipError('adamw_flat: parameter, gradient and moment buffers differ in size')
    nseg = hyp.shape[0]
    if seg_off.dtype != torch.int64 or seg_off.numel() != nseg + 1 or hyp.shape[1] != 4 or not seg_off.is_cuda:
        raise AotHipError('adamw_flat: seg_off int64 [nseg + 1], hyp fp32 [nseg, 4], both on the device')
    if sumsq is not None and sumsq.dtype != torch.float64:
        raise AotHipError('adamw_flat: the sum of squares must be float64')
    _chk(load().aot_adamw_flat_f32(_dev(p), _dev(g), _dev(m), _dev(v), p.numel(), _dev(seg_off), _dev(hyp), nseg, beta1, beta2, eps,
                                   _dev(sumsq) if sumsq is not None else None, float(max_norm),
                                   stream if stream is not None else stream_ptr()), 'aot_adamw_flat_f32')


def sumsq_accum(x, out, stream=None):
    """out (one fp64 element on the device) += sum(x^2)."""
    _flat_f32(x)
    if out.dtype != torch.float64:
        raise AotHipError('sumsq_accum: the accumulator must be float64')
    _chk(load().aot_sumsq_accum_f64(_dev(x), x.numel(), _dev(out), stream if stream is not None else stream_ptr()),
         'aot_sumsq_accum_f64')
