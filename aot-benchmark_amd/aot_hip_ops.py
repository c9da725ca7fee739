"""``torch.ops.aot_hip.*``: the C-ABI kernels registered as PyTorch custom ops.

``import aot_hip_ops`` defines one dispatcher op per fused stage, implemented for the ROCm ("CUDA") dispatch key only
by the ctypes wrappers of ``aot_hip`` -- a CPU tensor therefore fails in the dispatcher ("no CPU fallback"), and the ops
run on torch's current HIP stream.  All ops write into caller-provided outputs (``Tensor(a!)``), like the C ABI.

The engine itself calls ``aot_hip.<fn>`` directly: a frame is ~200 launches and the dispatcher adds several microseconds
to each, which matters once three clips share one GPU.  The ops are the integration surface for code
that lives in the PyTorch ecosystem (``torch.ops`` call sites in a patched reference module, FX graphs, profilers that
name ops).
"""
import torch

import aot_hip

_lib = torch.library.Library('aot_hip', 'DEF')

_DEFS = {
    'conv2d_nhwc': '(Tensor x, Tensor w, Tensor? bias, Tensor? res, Tensor(a!) out, int H, int W, int Cin, int OH, int OW, '
                   'int Cout, int KH, int KW, int stride, int pad, int dil, int act) -> ()',
    'dwconv2d_nhwc': '(Tensor x, Tensor w, Tensor? bias, Tensor(a!) out, int H, int W, int C, int OH, int OW, int K, int stride, '
                     'int pad, int dil, int act) -> ()',
    'layernorm': '(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) out, float eps) -> ()',
    'groupnorm': '(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) out, int groups, Tensor(b!) scratch, Tensor(c!) stats, '
                 'Tensor(d!) ticket, int act, float eps, int lanes) -> ()',
    'gn_act_dwconv5': '(Tensor x, Tensor gamma, Tensor beta, Tensor w, Tensor(a!) out, int groups, Tensor(b!) scratch, '
                      'Tensor(c!) stats, Tensor(d!) ticket, int h, int w_, int act, float eps, int lanes) -> ()',
    'attn': '(Tensor q, Tensor k, Tensor v, Tensor(a!) out, int T, int H, float scale_div, Tensor(b!)? part, int nsplit) -> ()',
    'attn_topk': '(Tensor q, Tensor k, Tensor v, Tensor(a!) out, int T, int H, float scale_div, int top_k, Tensor(b!) scores) -> ()',
    'gated_attn_topk': '(Tensor q, Tensor k, Tensor v, Tensor? gate, Tensor(a!) out, int T, float scale_div, int top_k, '
                       'Tensor(b!) scores) -> ()',
    'gated_attn': '(Tensor q, Tensor k, Tensor v, Tensor? gate, Tensor(a!) out, int T, float scale_div, Tensor(b!)? part, '
                  'int nsplit) -> ()',
    'local_attn': '(Tensor q, Tensor k, Tensor v, Tensor relk_w, Tensor relk_b, Tensor relv_t, Tensor(a!) out, int h, int w, int H, '
                  'float scale_div, int max_dis) -> ()',
    'local_gated': '(Tensor q, Tensor k, Tensor v, Tensor? gate, Tensor relk_t, Tensor relk_b, Tensor(a!) prob, Tensor(b!) out, int h, '
                   'int w, float scale_div, int max_dis) -> ()',
    'idbank': '(Tensor mask, Tensor table, Tensor? sumtab, Tensor bias, Tensor(a!) out, int H, int W, int OH, int OW, int K, int stride, '
              'int pad, int C, int nlabel, int lanes, int group_size) -> ()',
    'bilinear_nhwc': '(Tensor x, Tensor? add, Tensor(a!) out, int IH, int IW, int OH, int OW, int C, bool align_corners, int lanes, '
                     'bool add_shared) -> ()',
    'logits_finalize': '(Tensor logits, Tensor(a!) out4, Tensor(b!)? out, int IH, int IW, int C, int OH, int OW, int obj_total, '
                       'bool align_corners, int groups) -> ()',
    'preprocess': '(Tensor img, Tensor(a!) out, bool flip) -> ()',
    'fuse_probs': '(Tensor logits, int[] flips, Tensor? new_label) -> (Tensor, Tensor)',
    'label_resize': '(Tensor label, int out_h, int out_w, bool flip) -> Tensor',
}
for _name, _schema in _DEFS.items():
    _lib.define(_name + _schema)


def _impl(name):
    def deco(fn):
        _lib.impl(name, fn, 'CUDA')
        return fn
    return deco


@_impl('conv2d_nhwc')
def _conv2d(x, w, bias, res, out, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil, act):
    aot_hip.conv2d(x, w, bias, out, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, dil, res=res, act=act)


@_impl('dwconv2d_nhwc')
def _dwconv2d(x, w, bias, out, H, W, C, OH, OW, K, stride, pad, dil, act):
    aot_hip.dwconv2d(x, w, bias, out, H, W, C, OH, OW, K, stride, pad, dil, act=act)


@_impl('layernorm')
def _layernorm(x, gamma, beta, out, eps):
    aot_hip.layernorm(x, gamma, beta, out, eps=eps)


@_impl('groupnorm')
def _groupnorm(x, gamma, beta, out, groups, scratch, stats, ticket, act, eps, lanes):
    nsplit = scratch.numel() // (2 * groups * lanes)
    aot_hip.groupnorm(x, gamma, beta, out, groups, (scratch, stats, ticket), act=act, eps=eps, nsplit=nsplit, B=lanes)


@_impl('gn_act_dwconv5')
def _gn_act_dwconv5(x, gamma, beta, w, out, groups, scratch, stats, ticket, h, w_, act, eps, lanes):
    nsplit = scratch.numel() // (2 * groups * lanes)
    aot_hip.gn_act_dwconv5(x, gamma, beta, w, out, groups, (scratch, stats, ticket), h, w_, act=act, eps=eps, nsplit=nsplit, B=lanes)


@_impl('attn')
def _attn(q, k, v, out, T, H, scale_div, part, nsplit):
    aot_hip.attention(q, k, v, out, T, H, scale_div, part=part, nsplit=nsplit)


@_impl('attn_topk')
def _attn_topk(q, k, v, out, T, H, scale_div, top_k, scores):
    aot_hip.attention_topk(q, k, v, out, T, H, scale_div, top_k, scores)


@_impl('gated_attn_topk')
def _gated_attn_topk(q, k, v, gate, out, T, scale_div, top_k, scores):
    aot_hip.gated_attention_topk(q, k, v, gate, out, T, scale_div, top_k, scores)


@_impl('gated_attn')
def _gated_attn(q, k, v, gate, out, T, scale_div, part, nsplit):
    aot_hip.gated_attention(q, k, v, gate, out, T, scale_div, part=part, nsplit=nsplit)


@_impl('local_attn')
def _local_attn(q, k, v, relk_w, relk_b, relv_t, out, h, w, H, scale_div, max_dis):
    aot_hip.local_attention(q, k, v, relk_w, relk_b, relv_t, out, h, w, H, scale_div, max_dis=max_dis)


@_impl('local_gated')
def _local_gated(q, k, v, gate, relk_t, relk_b, prob, out, h, w, scale_div, max_dis):
    aot_hip.local_gated(q, k, v, gate, relk_t, relk_b, prob, out, h, w, scale_div, max_dis=max_dis)


@_impl('idbank')
def _idbank(mask, table, sumtab, bias, out, H, W, OH, OW, K, stride, pad, C, nlabel, lanes, group_size):
    aot_hip.idbank(mask, table, bias, out, H, W, OH, OW, K, stride, pad, C, nlabel, sumtab=sumtab, G=lanes, group_size=group_size)


@_impl('bilinear_nhwc')
def _bilinear(x, add, out, IH, IW, OH, OW, C, align_corners, lanes, add_shared):
    aot_hip.bilinear(x, out, IH, IW, OH, OW, C, align_corners, add=add, B=lanes, add_shared=add_shared)


@_impl('logits_finalize')
def _logits_finalize(logits, out4, out, IH, IW, C, OH, OW, obj_total, align_corners, groups):
    aot_hip.logits_finalize(logits, out4, out, IH, IW, C, OH, OW, obj_total, align_corners, G=groups)


@_impl('preprocess')
def _preprocess(img, out, flip):
    aot_hip.preprocess(img, out.shape[-2], out.shape[-1], flip, out=out)


@_impl('fuse_probs')
def _fuse_probs(logits, flips, new_label):
    fused, aug, _ = aot_hip.fuse_probs(logits, [bool(f) for f in flips], new_label=new_label)
    return fused, aug


@_impl('label_resize')
def _label_resize(label, out_h, out_w, flip):
    return aot_hip.label_resize(label, out_h, out_w, flip)


def op_names():
    return sorted(_DEFS)
