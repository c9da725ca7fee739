"""Differentiable primitives of the training path: torch.autograd.Function wrappers over the C ABI
(csrc/train_bwd.hip and the forward kernels of the inference path).  The reference builds its autograd graph out of
ATen ops (networks/engines/aot_engine.py:33-108 under loss.backward(), trainer.py:460-519); here every node of the
graph is one of the few primitives below, each with a hand-written backward:

    matmul      C = alpha * A . B (+ bias) on arbitrarily strided 3-D views (QK^T, PV, the relative-position products and -- on
                transposed views -- every one of their gradients; small or odd-shaped linears)
    linear      nn.Linear / 1x1 convs / the matmul half of KxK convs: forward, dgrad and wgrad on the LDS-direct fp32 GEMM kernels
                of the inference path (aot_conv2d_nhwc_f32)
    im2col      KxK convolutions = im2col + matmul (adjoint: col2im)
    dwconv2d    depthwise KxK
    act, layernorm, groupnorm, softmax_rows, bilinear, window_gather / window_scatter, to_nchw, maxpool3x3s2 (forward only)

All tensors are fp32 on a ROCm device, activations token-major / NHWC ([B*H*W, C]).  torch itself is used for views,
concatenation and elementwise adds / multiplies of the graph (plumbing); there is no CPU path."""
import torch
from torch.autograd import Function

import aot_hip
from aot_hip import _chk, _dev, _opt, load, stream_ptr

ACT = {'none': aot_hip.ACT_NONE, 'relu': aot_hip.ACT_RELU, 'relu6': aot_hip.ACT_RELU6, 'gelu': aot_hip.ACT_GELU,
       'silu': aot_hip.ACT_SILU}


# ---- arithmetic of the conv / linear products ------------------------------------------------------------------------------
# 'f32' (default): exact fp32 products on v_mfma_f32_32x32x2_f32.  'bf16': forward and dgrad on v_mfma_f32_32x32x16_bf16 -- both
# operands rounded to bf16 (round to nearest even: v_cvt_pk_bf16_f32 for the activations, the weight pre-rounded by
# aot_pack_bf16_f32), fp32 accumulation, fp32 outputs, fp32 master weights: what `--amp` (trainer.py:460-487, fp16 autocast +
# GradScaler there) / BASELINE config 5 ask for, without a loss scale (bf16 keeps fp32's exponent range).  The weight gradient
# (a long reduction into few tiles: split-K) and the attention products stay fp32.  A Function records the mode at forward time
# and its backward uses the same one, wherever backward() is called from.
_PRECISION = ['f32']


class matmul_precision:
    def __init__(self, mode):
        if mode not in ('f32', 'bf16'):
            raise ValueError("matmul_precision: 'f32' or 'bf16'")
        self.mode = mode

    def __enter__(self):
        self.prev = _PRECISION[0]
        _PRECISION[0] = self.mode
        return self

    def __exit__(self, *exc):
        _PRECISION[0] = self.prev
        return False


# Within one optimiser step a weight is used once per frame and sample group (forward) and once more per use in backward (dgrad):
# its k-major copy and its bf16 planes are made once per step.  The cache lives only inside `with weight_cache():` (TrainStep opens
# one per step): the optimiser's kernels update parameters behind torch's version counters, so nothing may outlive a step.
_WCACHE = [None]


class weight_cache:
    def __enter__(self):
        self.prev = _WCACHE[0]
        _WCACHE[0] = {}
        return self

    def __exit__(self, *exc):
        _WCACHE[0] = self.prev
        return False


def _cached(key, make):
    c = _WCACHE[0]
    if c is None or key is None:
        return make()
    v = c.get(key)
    if v is None:
        v = c[key] = make()
    return v


def _dense(x2, rpad, cpad):
    """A 2-D fp32 view (any strides) as a dense [rpad, cpad] matrix, zero beyond its own extent: the view itself when it already is
    one (16-byte aligned), else one launch of aot_copy2d_pad_f32."""
    R, C = x2.shape
    if rpad == R and cpad == C and x2.stride(1) == 1 and x2.stride(0) == C and x2.data_ptr() % 16 == 0:
        return x2
    out = torch.empty(rpad, cpad, dtype=torch.float32, device=x2.device)
    _chk(load().aot_copy2d_pad_f32(_dev(x2), _dev(out), R, C, x2.stride(0), x2.stride(1), rpad, cpad, cpad, stream_ptr()), 'aot_copy2d_pad_f32')
    return out


def _f32c(t):
    """fp32, contiguous (the streaming kernels walk raw memory)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ---- matmul ------------------------------------------------------------------------------------------------------------
MATMUL_LOG = None      # a dict: shapes of the products that take the strided kernel are counted into it


def _matmul_raw(a, b, bias=None, alpha=1.0, out=None):
    """a [bt, m, k], b [bt, k, n]: any strides (views welcome) -> contiguous [bt, m, n].  A few large matrices (the gated
    propagation's QK^T / PV and their gradients: one per sample) go to the LDS-direct GEMM kernels, the reduction length
    zero-padded to a multiple of 32; everything else -- many small matrices per head / window, odd widths -- to the strided kernel."""
    bt, m, k = a.shape
    n = b.shape[2]
    if b.shape[0] != bt or b.shape[1] != k:
        raise aot_hip.AotHipError('matmul shapes %s x %s' % (tuple(a.shape), tuple(b.shape)))
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise aot_hip.AotHipError('matmul operands must be float32')
    small = lambda r, c: 4 * r * (c + 64) < 2 ** 31           # the tile kernels address an operand with 32-bit byte offsets
    if out is None and alpha == 1.0 and bt <= 4 and (k >= 64 or n > 32) and 2.0 * m * n * k >= 5e7 and small(m, k) and small(k, n) \
            and small(m, n):
        # zero-padding makes any shape fit the tile kernels: the reduction length to the split-K granule, a narrow output (the
        # decoder's 11 logits) to 64 columns; long reductions into few output tiles (weight gradients) are split over K.  Each
        # operand view (row-major, transposed, broadcast) becomes a dense padded matrix in ONE launch (aot_copy2d_pad_f32) -- or is
        # used where it lies; the product lands in its slab of the result.
        tiles = -(-m // 64) * -(-max(n, 64) // 64)
        ks = max(1, min(15, 256 // tiles)) if k >= 4096 else 1
        kp = -(-k // (32 * ks)) * (32 * ks)
        npad = n if (n % 4 == 0 and n > 32) else max(64, -(-n // 4) * 4)
        bias_p = bias if (bias is None or npad == n) else torch.nn.functional.pad(bias, (0, npad - n))
        c = torch.empty(bt, m, n, dtype=torch.float32, device=a.device)
        direct = npad == n and (m * n) % 4 == 0
        for i in range(bt):
            ai, bti = _dense(a[i], m, kp), _dense(b[i].t(), npad, kp)      # (the kernel reads B through its k-contiguous twin only)
            if direct:
                _gemm_lean(ai, None, bti, bias_p, ks=ks, out=c[i])
            else:
                c[i].copy_(_gemm_lean(ai, None, bti, bias_p, ks=ks)[:, :n])
        return c
    if MATMUL_LOG is not None:          # (tools/dev: which products stay on the strided kernel)
        key = (bt, m, k, n, float(alpha), tuple(a.stride()), tuple(b.stride()))
        MATMUL_LOG[key] = MATMUL_LOG.get(key, 0) + 1
    c = out if out is not None else torch.empty(bt, m, n, dtype=torch.float32, device=a.device)
    sa, sb = a.stride(), b.stride()
    _chk(load().aot_matmul_strided_f32(_dev(a), _dev(b), _opt(bias), _dev(c), bt, m, n, k, sa[0], sa[1], sa[2], sb[0], sb[1],
                                       sb[2], m * n, n, float(alpha), 0, stream_ptr()), 'aot_matmul_strided_f32')
    return c


class _Matmul(Function):
    @staticmethod
    def forward(ctx, a, b, bias, alpha):
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        ctx.has_bias = bias is not None
        return _matmul_raw(a, b, bias, alpha)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = _f32c(dc)
        da = db = dbias = None
        if ctx.needs_input_grad[0]:
            da = _matmul_raw(dc, b.transpose(1, 2), alpha=ctx.alpha)            # dC . B^T
        if ctx.needs_input_grad[1]:
            db = _matmul_raw(a.transpose(1, 2), dc, alpha=ctx.alpha)            # A^T . dC
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = _colsum(dc.view(dc.shape[0] * dc.shape[1], dc.shape[2]))      # (a 1-row product on the strided kernel took 440 us)
        return da, db, dbias, None


def matmul(a, b, bias=None, alpha=1.0):
    """alpha * a @ b (+ bias): a [bt, m, k], b [bt, k, n] (any strides), bias [n]."""
    return _Matmul.apply(a, b, bias, alpha)


# ---- column permutation -----------------------------------------------------------------------------------------------------------
class _PermuteCols(Function):
    @staticmethod
    def forward(ctx, x, perm, inv):
        x = _f32c(x)
        R, C = x.shape
        out = torch.empty(R, perm.numel(), dtype=torch.float32, device=x.device)
        _chk(load().aot_gather_cols_f32(_dev(x), _dev(perm), _dev(out), R, C, perm.numel(), stream_ptr()), 'aot_gather_cols_f32')
        ctx.save_for_backward(inv)
        return out

    @staticmethod
    def backward(ctx, dy):
        inv, = ctx.saved_tensors
        dy = _f32c(dy)
        R, C = dy.shape
        dx = torch.empty(R, inv.numel(), dtype=torch.float32, device=dy.device)
        _chk(load().aot_gather_cols_f32(_dev(dy), _dev(inv), _dev(dx), R, C, inv.numel(), stream_ptr()), 'aot_gather_cols_f32')
        return dx, None, None


def permute_cols(x, perm):
    """out[:, t] = x[:, perm[t]] for a PERMUTATION perm of the columns of x [R, C] (what a product with the 0 / 1 matrix M[perm[t], t] = 1
    computes); the gradient is the gather by the inverse permutation."""
    perm = perm.to(device=x.device, dtype=torch.int32).contiguous()
    return _PermuteCols.apply(x, perm, torch.argsort(perm).to(torch.int32).contiguous())


# ---- nn.Linear on the LDS-direct GEMM kernels ---------------------------------------------------------------------------------
# y = x W^T + b, dx = dy W, dW = dy^T x are plain row-major GEMMs: they run on the fp32 tile kernels of the inference path
# (aot_conv2d_nhwc_f32, csrc/gemm_lds.hip: both operands by LDS-DMA) instead of the strided general kernel whenever the shapes allow
# it.  That entry takes the second operand twice -- k-major [K, N] for its general kernels and as k-contiguous rows [N, K] for the
# tile kernel -- so: forward = (W^T copy, W), dgrad = (W, the same W^T copy), wgrad = (x, x^T) with dy^T as the first operand and the
# row count (the reduction length) zero-padded to the split-K granule; bias gradient = column sums (aot_col_reduce_f32);
# the operand copies of the weight gradient (dy^T, x^T, zero-padded) are one aot_transpose_pad_f32 launch each.
_LEAN_MIN_ROWS = 64


def _lean_ok(M, K, N):
    return M >= _LEAN_MIN_ROWS and K % 32 == 0 and N % 4 == 0 and N > 32


def _gemm_lean(a, w_kn, w_nk, bias=None, ks=1, prec='f32', key=None, out=None):
    """a [M, K] @ w_kn [K, N] (+ bias) with w_nk = w_kn^T; ks > 1: split-K through a scratch slab (K / 32 divisible by ks).
    prec = 'bf16' (and no split): the bf16 matrix cores, operands rounded to nearest even; `key` names w_kn for the per-step cache
    of its packed plane.  w_kn may be None on the fp32 path (the LDS-direct kernels read the k-contiguous twin only)."""
    M, K = a.shape
    N = w_nk.shape[0]
    if prec == 'bf16' and ks == 1 and 4 * M * max(K, N) < 2 ** 31:
        wq = _cached(None if key is None else (key, 'bf16'), lambda: aot_hip.pack_bf16(w_kn))
        return aot_hip.gemm_bf16_packed(a, wq, N, bias)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    scratch = torch.empty(ks * M * N, dtype=torch.float32, device=a.device) if ks > 1 else None
    aot_hip.conv2d_cfg(a, w_kn, bias, out, 1, M, K, 1, M, N, cfg=(-2 if w_kn is not None else 197) if ks == 1 else 196 + ks, wt=w_nk,
                       scratch=scratch)
    return out


_REDUCE_WS = {}


def _col_reduce(dy, xhat):
    """(sum_r dy xhat, sum_r dy) over the rows of [R, C] (fp64 partials in fixed order, one launch: aot_col_reduce_f32); xhat None:
    column sums only.  The scratch (chunk partials, self re-arming tickets) is kept per device and stream."""
    R, C = dy.shape
    nchunk = max(1, min(64, R // 128))
    key = (dy.device, torch.cuda.current_stream(dy.device).cuda_stream)
    ws = _REDUCE_WS.get(key)
    if ws is None or ws[0].numel() < 2 * nchunk * C or ws[1].numel() < (C + 31) // 32:
        ws = _REDUCE_WS[key] = (torch.empty(2 * 64 * max(C, 2048), dtype=torch.float64, device=dy.device),
                                torch.zeros(max(64, (max(C, 2048) + 31) // 32), dtype=torch.int32, device=dy.device))
    dg = torch.empty(C, dtype=torch.float32, device=dy.device) if xhat is not None else None
    db = torch.empty(C, dtype=torch.float32, device=dy.device)
    _chk(load().aot_col_reduce_f32(_dev(dy), _opt(xhat), _opt(dg), _dev(db), R, C, _dev(ws[0]), _dev(ws[1]), nchunk, stream_ptr()),
         'aot_col_reduce_f32')
    return dg, db


def _colsum(x):
    """column sums of x [R, C] (fp64 partials, fixed order)."""
    return _col_reduce(x, None)[1]


def _transpose_pad(x, rpad, transpose=True):
    """x [R, C] -> x^T zero-padded to [C, rpad] (or, transpose = False, x followed by zero rows: [rpad, C]) in one launch."""
    R, C = x.shape
    out = torch.empty((C, rpad) if transpose else (rpad, C), dtype=torch.float32, device=x.device)
    _chk(load().aot_transpose_pad_f32(_dev(x), _dev(out), R, C, x.stride(0), out.stride(0), rpad, 1 if transpose else 0, stream_ptr()),
         'aot_transpose_pad_f32')
    return out


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, wkey):
        x, weight = _f32c(x), _f32c(weight)
        w_kn = _cached(None if wkey is None else (wkey, 'kn'), lambda: weight.t().contiguous())
        ctx.save_for_backward(x, weight, w_kn)
        ctx.has_bias = bias is not None
        ctx.prec = _PRECISION[0]
        ctx.wkey = wkey
        return _gemm_lean(x, w_kn, weight, _f32c(bias) if bias is not None else None, prec=ctx.prec,
                          key=None if wkey is None else (wkey, 'fwd'))

    @staticmethod
    def backward(ctx, dy):
        x, weight, w_kn = ctx.saved_tensors
        dy = _f32c(dy)
        M, K = x.shape
        N = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if _lean_ok(M, N, K):
                dx = _gemm_lean(dy, weight, w_kn, prec=ctx.prec,                      # dy [M, N] . W [N, K]
                                key=None if ctx.wkey is None else (ctx.wkey, 'dgrad'))
            else:
                dx = _matmul_raw(dy.unsqueeze(0), weight.unsqueeze(0))[0]
        if ctx.needs_input_grad[1]:
            ks = max(1, min(15, 256 // max(1, -(-N // 64) * -(-K // 64))))              # enough workgroups for 256 CUs
            Mp = -(-M // (32 * ks)) * (32 * ks)
            if M >= 1024 and _lean_ok(N, Mp, K) and (Mp // 32) % ks == 0 and 4 * Mp * max(N, K) < 2 ** 31:
                dyt = _transpose_pad(dy, Mp)                                            # dy^T [N, Mp], zero behind column M
                if ctx.prec == 'bf16' and 4 * N * K < 2 ** 31:
                    # bf16 products: x packed (rounded) into the tile order instead of transposed, split-K on the bf16 kernel
                    xp = x if Mp == M else _transpose_pad(x, Mp, transpose=False)
                    dw = aot_hip.gemm_bf16_packed(dyt, aot_hip.pack_bf16(xp), K, ks=ks)
                else:
                    dw = _gemm_lean(dyt, None, _transpose_pad(x, Mp), ks=ks)         # dy^T [N, M] . x [M, K]: x through its twin x^T
            else:
                dw = _matmul_raw(dy.t().unsqueeze(0), x.unsqueeze(0))[0]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy)
        return dx, dw, db, None


def linear(x, weight, bias=None, wkey=None):
    """nn.Linear on token-major x [M, in]: weight [out, in] as the parameter holds it.  wkey: what identifies the weight inside a
    weight_cache() scope (default: the tensor's own address -- right for parameters, not for temporaries)."""
    M, K = x.shape
    if _lean_ok(M, K, weight.shape[0]):
        if wkey is None and isinstance(weight, torch.nn.Parameter):
            wkey = (weight.data_ptr(), tuple(weight.shape))
        return _Linear.apply(x, weight, bias, wkey)
    return matmul(x.unsqueeze(0), weight.t().unsqueeze(0), bias)[0]


# ---- KxK convolution = im2col + matmul ---------------------------------------------------------------------------------
def _osz(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


class _Im2col(Function):
    @staticmethod
    def forward(ctx, x, geom):
        B, H, W, C, KH, KW, stride, pad, dil = geom
        OH, OW = _osz(H, KH, stride, pad, dil), _osz(W, KW, stride, pad, dil)
        x = _f32c(x)
        cols = torch.empty(B * OH * OW, KH * KW * C, dtype=torch.float32, device=x.device)
        _chk(load().aot_im2col_f32(_dev(x), _dev(cols), B, H, W, C, OH, OW, KH, KW, stride, pad, dil, stream_ptr()), 'aot_im2col_f32')
        ctx.geom = geom
        return cols

    @staticmethod
    def backward(ctx, dcols):
        B, H, W, C, KH, KW, stride, pad, dil = ctx.geom
        OH, OW = _osz(H, KH, stride, pad, dil), _osz(W, KW, stride, pad, dil)
        dcols = _f32c(dcols)
        dx = torch.empty(B * H * W, C, dtype=torch.float32, device=dcols.device)
        _chk(load().aot_col2im_f32(_dev(dcols), _dev(dx), B, H, W, C, OH, OW, KH, KW, stride, pad, dil, stream_ptr()), 'aot_col2im_f32')
        return dx, None


def conv2d(x, weight, bias, B, H, W, stride=1, pad=0, dil=1):
    """nn.Conv2d (groups = 1) on B NHWC maps x [B*H*W, Cin] (Cin % 4 == 0 for KxK): weight [Cout, Cin, KH, KW] as the
    parameter holds it.  Returns ([B*OH*OW, Cout], OH, OW)."""
    cout, cin, kh, kw = weight.shape
    OH, OW = _osz(H, kh, stride, pad, dil), _osz(W, kw, stride, pad, dil)
    wkey = getattr(weight, '_aot_wkey', None)
    if wkey is None and isinstance(weight, torch.nn.Parameter):
        wkey = (weight.data_ptr(), tuple(weight.shape))
    if wkey is not None:
        wkey = wkey + (x.shape[1],)
    if kh == 1 and kw == 1 and stride == 1 and pad == 0:
        return linear(x, weight.view(cout, cin), bias, wkey), OH, OW
    if x.shape[1] != cin:           # channel padding of the input (one-hot maps: 11 -> 12): pad the weight with zeros
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, x.shape[1] - cin))
        cin = x.shape[1]
    cols = _Im2col.apply(x, (B, H, W, cin, kh, kw, stride, pad, dil))
    wmat = weight.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)          # k = (ky, kx, c), as im2col lays the taps out
    return linear(cols, wmat, bias, wkey), OH, OW


# ---- depthwise convolution ---------------------------------------------------------------------------------------------
class _DwConv(Function):
    @staticmethod
    def forward(ctx, x, wk, geom):
        B, H, W, C, K, stride, pad, dil = geom
        OH, OW = _osz(H, K, stride, pad, dil), _osz(W, K, stride, pad, dil)
        x, wk = _f32c(x), _f32c(wk)
        y = torch.empty(B * OH * OW, C, dtype=torch.float32, device=x.device)
        aot_hip.dwconv2d(x, wk, None, y, H, W, C, OH, OW, K, stride, pad, dil, B=B)
        ctx.save_for_backward(x, wk)
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wk = ctx.saved_tensors
        B, H, W, C, K, stride, pad, dil = ctx.geom
        OH, OW = _osz(H, K, stride, pad, dil), _osz(W, K, stride, pad, dil)
        dy = _f32c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _chk(load().aot_dwconv2d_bwd_data_f32(_dev(dy), _dev(wk), _dev(dx), B, H, W, C, OH, OW, K, K, stride, pad, dil,
                                                  stream_ptr()), 'aot_dwconv2d_bwd_data_f32')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(wk)
            _chk(load().aot_dwconv2d_bwd_weight_f32(_dev(dy), _dev(x), _dev(dw), B, H, W, C, OH, OW, K, K, stride, pad, dil,
                                                    stream_ptr()), 'aot_dwconv2d_bwd_weight_f32')
        return dx, dw, None


def dwconv2d(x, weight, B, H, W, stride=1, pad=0, dil=1):
    """Depthwise nn.Conv2d (groups = C, no bias) on B NHWC maps x [B*H*W, C] (C % 4 == 0): weight [C, 1, K, K]."""
    C, _, K, _ = weight.shape
    wk = weight.reshape(C, K * K).t()                  # [K*K, C], the kernels' tap-major layout
    y = _DwConv.apply(x, wk, (B, H, W, C, K, stride, pad, dil))
    return y, _osz(H, K, stride, pad, dil), _osz(W, K, stride, pad, dil)


def maxpool3x3s2(x, H, W):
    """nn.MaxPool2d(3, 2, 1) on one NHWC map x [H*W, C] -> ([OH*OW, C], OH, OW).  Forward only: it sits behind the ResNet stem,
    which every reference recipe freezes (TRAIN_ENCODER_FREEZE_AT >= 1), so no gradient ever reaches it."""
    if x.requires_grad:
        raise NotImplementedError('the max pool has no backward kernel: a trainable ResNet stem (TRAIN_ENCODER_FREEZE_AT = 0) '
                                  'is not built')
    x = _f32c(x)
    OH, OW = _osz(H, 3, 2, 1, 1), _osz(W, 3, 2, 1, 1)
    y = torch.empty(OH * OW, x.shape[1], dtype=torch.float32, device=x.device)
    aot_hip.maxpool3x3s2(x, y, H, W, x.shape[1], OH, OW)
    return y, OH, OW


# ---- pointwise / normalisation -----------------------------------------------------------------------------------------
class _Act(Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _f32c(x)
        y = torch.empty_like(x)
        _chk(load().aot_act_f32(_dev(x), _dev(y), x.numel(), kind, stream_ptr()), 'aot_act_f32')
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        _chk(load().aot_act_bwd_f32(_dev(x), _dev(dy), _dev(dx), x.numel(), ctx.kind, stream_ptr()), 'aot_act_bwd_f32')
        return dx, None


def act(x, kind):
    return x if kind == 'none' else _Act.apply(x, ACT[kind])


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        y = torch.empty_like(x)
        aot_hip.layernorm(x, gamma, beta, y, eps=eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        M, C = x.shape
        dx, xhat = torch.empty_like(x), torch.empty_like(x)
        _chk(load().aot_layernorm_bwd_f32(_dev(x), _dev(dy), _dev(gamma), _dev(dx), _dev(xhat), M, C, ctx.eps, stream_ptr()),
             'aot_layernorm_bwd_f32')
        dg, db = _col_reduce(dy, xhat)
        return dx, dg, db, None


def layernorm(x, gamma, beta, eps=1e-5):
    """nn.LayerNorm over the last dim of x [M, C]."""
    return _LayerNorm.apply(x, gamma, beta, eps)


class _GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, B, eps):
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        dev = x.device
        nsplit = 8
        bufs = (torch.empty(B * groups * nsplit * 2, dtype=torch.float64, device=dev),
                torch.empty(B * groups * 2, dtype=torch.float64, device=dev),
                torch.zeros(B * groups, dtype=torch.int32, device=dev))
        y = torch.empty_like(x)
        aot_hip.groupnorm(x, gamma, beta, y, groups, bufs, act=aot_hip.ACT_NONE, eps=eps, nsplit=nsplit, B=B)
        ctx.save_for_backward(x, gamma, bufs[1])
        ctx.groups, ctx.B = groups, B
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        dy = _f32c(dy)
        R, C = x.shape
        dx, xhat = torch.empty_like(x), torch.empty_like(x)
        M = R // ctx.B
        nchunk = max(1, min(64, M // 256))
        slots = ctx.B * ctx.groups
        key = ('gn', x.device, torch.cuda.current_stream(x.device).cuda_stream)
        ws = _REDUCE_WS.get(key)
        if ws is None or ws[1].numel() < slots:
            ws = _REDUCE_WS[key] = (torch.empty(2 * 64 * max(slots, 256), dtype=torch.float64, device=x.device),
                                    torch.zeros(max(slots, 256), dtype=torch.int32, device=x.device),
                                    torch.empty(2 * max(slots, 256), dtype=torch.float32, device=x.device))
        if C % 4 == 0:
            _chk(load().aot_groupnorm_bwd2_f32(_dev(x), _dev(dy), _dev(stats), _dev(gamma), _dev(dx), _dev(xhat), _dev(ws[0]), _dev(ws[1]),
                                               _dev(ws[2]), ctx.B, M, C, ctx.groups, nchunk, stream_ptr()), 'aot_groupnorm_bwd2_f32')
        else:
            _chk(load().aot_groupnorm_bwd_f32(_dev(x), _dev(dy), _dev(stats), _dev(gamma), _dev(dx), _dev(xhat), ctx.B, M, C,
                                              ctx.groups, stream_ptr()), 'aot_groupnorm_bwd_f32')
        dg, db = _col_reduce(dy, xhat)
        return dx, dg, db, None, None, None


def groupnorm(x, gamma, beta, groups, B=1, eps=1e-5):
    """nn.GroupNorm on B NHWC maps x [B*M, C] (statistics per map and group)."""
    return _GroupNorm.apply(x, gamma, beta, groups, B, eps)


class _SoftmaxRows(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        rows = x.numel() // x.shape[-1]
        _chk(load().aot_softmax_rows_f32(_dev(x), _dev(y), rows, x.shape[-1], stream_ptr()), 'aot_softmax_rows_f32')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(y)
        _chk(load().aot_softmax_rows_bwd_f32(_dev(y), _dev(dy), _dev(dx), y.numel() // y.shape[-1], y.shape[-1], stream_ptr()),
             'aot_softmax_rows_bwd_f32')
        return dx


def softmax_rows(x):
    """softmax over the last dim (entries at -inf give exactly 0)."""
    return _SoftmaxRows.apply(x)


class _Bilinear(Function):
    @staticmethod
    def forward(ctx, x, geom):
        B, IH, IW, OH, OW, align = geom
        x = _f32c(x)
        C = x.shape[1]
        y = torch.empty(B * OH * OW, C, dtype=torch.float32, device=x.device)
        aot_hip.bilinear(x, y, IH, IW, OH, OW, C, align, B=B)
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, dy):
        B, IH, IW, OH, OW, align = ctx.geom
        dy = _f32c(dy)
        C = dy.shape[1]
        dx = torch.empty(B * IH * IW, C, dtype=torch.float32, device=dy.device)
        _chk(load().aot_bilinear_bwd_nhwc_f32(_dev(dy), _dev(dx), B, IH, IW, OH, OW, C, int(align), stream_ptr()),
             'aot_bilinear_bwd_nhwc_f32')
        return dx, None


def bilinear(x, B, IH, IW, OH, OW, align_corners):
    """F.interpolate(mode='bilinear') on B NHWC maps x [B*IH*IW, C] (C % 4 == 0)."""
    return _Bilinear.apply(x, (B, IH, IW, OH, OW, bool(align_corners)))


class _WindowGather(Function):
    @staticmethod
    def forward(ctx, dense, geom):
        h, w, R, fill = geom
        dense = _f32c(dense)
        G, N = dense.shape[0], h * w
        win = torch.empty(G, N, (2 * R + 1) ** 2, dtype=torch.float32, device=dense.device)
        _chk(load().aot_window_gather_f32(_dev(dense), _dev(win), G, h, w, R, float(fill), stream_ptr()), 'aot_window_gather_f32')
        ctx.geom = geom
        return win

    @staticmethod
    def backward(ctx, dwin):
        h, w, R, _ = ctx.geom
        dwin = _f32c(dwin)
        G, N = dwin.shape[0], h * w
        dd = torch.empty(G, N, N, dtype=torch.float32, device=dwin.device)
        _chk(load().aot_window_scatter_f32(_dev(dwin), _dev(dd), G, h, w, R, 0.0, stream_ptr()), 'aot_window_scatter_f32')
        return dd, None


class _WindowScatter(Function):
    @staticmethod
    def forward(ctx, win, geom):
        h, w, R, fill = geom
        win = _f32c(win)
        G, N = win.shape[0], h * w
        dense = torch.empty(G, N, N, dtype=torch.float32, device=win.device)
        _chk(load().aot_window_scatter_f32(_dev(win), _dev(dense), G, h, w, R, float(fill), stream_ptr()), 'aot_window_scatter_f32')
        ctx.geom = geom
        return dense

    @staticmethod
    def backward(ctx, dd):
        h, w, R, _ = ctx.geom
        dd = _f32c(dd)
        G, N = dd.shape[0], h * w
        dwin = torch.empty(G, N, (2 * R + 1) ** 2, dtype=torch.float32, device=dd.device)
        _chk(load().aot_window_gather_f32(_dev(dd), _dev(dwin), G, h, w, R, 0.0, stream_ptr()), 'aot_window_gather_f32')
        return dwin, None


def window_gather(dense, h, w, max_dis, fill=0.0):
    """dense [G, N, N] -> [G, N, (2R+1)^2]: the entries at each query's window keys (`fill` where the key is outside the map)."""
    return _WindowGather.apply(dense, (h, w, max_dis, fill))


def window_scatter(win, h, w, max_dis, fill=0.0):
    """[G, N, (2R+1)^2] -> dense [G, N, N] (`fill` outside each query's window): local2global, attention.py:378-417."""
    return _WindowScatter.apply(win, (h, w, max_dis, fill))


class _ToNCHW(Function):
    @staticmethod
    def forward(ctx, x, H, W):
        x = _f32c(x)
        C = x.shape[1]
        y = torch.empty(1, C, H, W, dtype=torch.float32, device=x.device)
        aot_hip.nhwc_to_nchw(x, y, C, H, W)
        ctx.geom = (C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        C, H, W = ctx.geom
        dy = _f32c(dy)
        dx = torch.empty(H * W, C, dtype=torch.float32, device=dy.device)
        aot_hip.nchw_to_nhwc(dy, dx, C, H, W, C)
        return dx, None, None


def to_nchw(x, H, W):
    """token-major [H*W, C] -> [1, C, H, W] (the layout the losses and callers of the reference surface take)."""
    return _ToNCHW.apply(x, H, W)


class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x, cpad):
        x = _f32c(x)
        _, C, H, W = x.shape
        y = torch.empty(H * W, cpad, dtype=torch.float32, device=x.device)
        aot_hip.nchw_to_nhwc(x, y, C, H, W, cpad)
        ctx.geom = (C, H, W, cpad)
        return y

    @staticmethod
    def backward(ctx, dy):
        C, H, W, cpad = ctx.geom
        dy = _f32c(dy)
        full = torch.empty(1, cpad, H, W, dtype=torch.float32, device=dy.device)
        aot_hip.nhwc_to_nchw(dy, full, cpad, H, W)
        return full[:, :C].contiguous(), None


def to_nhwc(x, cpad=None):
    """[1, C, H, W] -> token-major [H*W, cpad] (channels >= C zero): the one-hot / probability map in front of the identity bank."""
    return _ToNHWC.apply(x, cpad if cpad is not None else x.shape[1])
