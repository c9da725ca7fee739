"""CPU oracle for the AOT / DeAOT per-frame inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``aot-benchmark_amd/`` may import this
module: it is the checker for ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``, never the thing that is shipped or measured.

This is a from-scratch *functional* restatement (plain torch ops on a
``state_dict`` that uses the reference's key names) of the algorithm in
yoxu515/aot-benchmark.  Every function cites the reference file:line it
follows (paths relative to /root/reference).  Differences in *form* (not in
math):

* the short-term windowed attention is evaluated directly (shift-and-dot over
  the 15x15 window) instead of through ``F.unfold`` + ``local2global`` + a dense
  N x N matmul (networks/layers/attention.py:308-428, 789-903);
* out-of-image window slots are skipped instead of being pushed to -1e8; in
  fp32 ``exp(-1e8 - max)`` is exactly 0, so the two are identical;
* the memory bank is appended, not prepended (softmax-sum is order invariant
  up to rounding).

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against *outputs of the reference
itself* run in the build container: ``tests/golden/make_golden.py`` imports
/root/reference, drives the demo loop (tools/demo.py:187-235) and stores the
results under ``tests/golden/``; ``tests/test_oracle_golden.py`` replays them
through this module (clips over all model families up to the full-size BASELINE
configs and 44 objects, the reference's MultiheadAttention / GatedPropagation with
their top_k / max_mem_len_ratio knobs, its MultiRestrictSize / MultiToTensor
transform classes, all 13 model presets, its training engine's forward
(``train_forward``: four batches) and its ``Evaluator.evaluating`` loop run
unmodified on an in-memory sequence (``sequence_eval``: single-scale and
multi-scale + flip, with a new object injected mid-clip)).

One function is PARITY UNPINNED: ``cv2_cubic_resize`` restates OpenCV's INTER_CUBIC
(the reference's un-vendored, unpinned dependency ``opencv-python``, absent here)
from its published algorithm and is only cross-checked against torch's bicubic
(the evaluator-loop golden was made with this restatement standing in for
``cv2.resize``, so it pins the loop around the filter, not the filter).
"""
import math

import numpy as np

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# model presets -- configs/models/*.py + the 8 engine attributes of
# configs/default.py:79-86 that AOT.__init__ reads (all dropouts are identity
# in eval mode, so only the structural ones are kept here)
# --------------------------------------------------------------------------
_BASE = dict(vos='aot', encoder='mobilenetv2', enc_dims=(24, 32, 96, 1280), emb=256,
             lstt_num=1, heads=8, max_obj=10, align_corners=True,
             intermediate_lstt=True, mem_gap=9999)
SPECS = {
    'aott': dict(_BASE),
    'aots': dict(_BASE, lstt_num=2),
    'aotb': dict(_BASE, lstt_num=3),
    'aotl': dict(_BASE, lstt_num=3, mem_gap=5),
    'r50_aotl': dict(_BASE, encoder='resnet50', enc_dims=(256, 512, 1024, 1024), lstt_num=3, mem_gap=5),
    'r101_aotl': dict(_BASE, encoder='resnet101', enc_dims=(256, 512, 1024, 1024), lstt_num=3, mem_gap=5),
    'deaott': dict(_BASE, vos='deaot', heads=1, intermediate_lstt=False),
    'deaots': dict(_BASE, vos='deaot', heads=1, intermediate_lstt=False, lstt_num=2),
    'deaotb': dict(_BASE, vos='deaot', heads=1, intermediate_lstt=False, lstt_num=3),
    'deaotl': dict(_BASE, vos='deaot', heads=1, intermediate_lstt=False, lstt_num=3, mem_gap=5),
    'r50_deaotl': dict(_BASE, vos='deaot', heads=1, intermediate_lstt=False, encoder='resnet50',
                       enc_dims=(256, 512, 1024, 1024), lstt_num=3, mem_gap=5),
    'swinb_aotl': dict(_BASE, encoder='swin_base', enc_dims=(128, 256, 512, 512), lstt_num=3, mem_gap=5,
                       align_corners=False),
    'swinb_deaotl': dict(_BASE, vos='deaot', heads=1, intermediate_lstt=False, encoder='swin_base',
                         enc_dims=(128, 256, 512, 512), lstt_num=3, mem_gap=5, align_corners=False),
}


def _lin(x, sd, p):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _conv(x, sd, p, stride=1, padding=0, dilation=1, groups=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, padding, dilation, groups)


def _fbn(x, sd, p):
    # networks/layers/normalization.py:31-43 (eval branch: F.batch_norm, eps 1e-5)
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, 1e-5)


def silu(x):
    # networks/layers/attention.py:585-586
    return x * torch.sigmoid(x)


def seq_to_2d(t, size_2d):
    # networks/layers/basic.py:88-92
    h, w = size_2d
    _, n, c = t.shape
    return t.view(h, w, n, c).permute(2, 3, 0, 1).contiguous()


def one_hot_mask(mask, cls_num):
    # utils/image.py:69-74
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)
    idx = torch.arange(0, cls_num + 1, device=mask.device).view(1, -1, 1, 1)
    return (mask == idx).to(torch.get_default_dtype() if not mask.is_floating_point() else mask.dtype)


# --------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------
def resnet50_features(sd, x, p='encoder', blocks=(3, 4, 6)):
    """networks/encoders/resnet.py:140-157 (ResNet-50 [3,4,6] / ResNet-101 [3,4,23], :171-189, without layer4, stride 16)."""
    x = F.relu(_fbn(_conv(x, sd, p + '.conv1', 2, 3), sd, p + '.bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    xs = []
    for li, (nblk, stride) in enumerate(((blocks[0], 1), (blocks[1], 2), (blocks[2], 2)), start=1):
        for b in range(nblk):
            q = '%s.layer%d.%d' % (p, li, b)
            s = stride if b == 0 else 1
            # Bottleneck.forward, resnet.py:34-54
            o = F.relu(_fbn(_conv(x, sd, q + '.conv1'), sd, q + '.bn1'))
            o = F.relu(_fbn(_conv(o, sd, q + '.conv2', s, 1), sd, q + '.bn2'))
            o = _fbn(_conv(o, sd, q + '.conv3'), sd, q + '.bn3')
            if (q + '.downsample.0.weight') in sd:
                x = _fbn(_conv(x, sd, q + '.downsample.0', s), sd, q + '.downsample.1')
            x = F.relu(o + x)
        xs.append(x)
    xs.append(x)
    return xs


def mobilenetv2_plan():
    """Block plan of MobileNetV2 at output stride 16 (mobilenetv2.py:142-215).

    Returns a list of (feature_index, inp, oup, stride, dilation, expand)."""
    setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2],
               [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
    plan, inp, cur, rate, idx = [], 32, 2, 1, 1
    for t, c, n, s in setting:
        if cur == 16:
            stride, dil = 1, rate
            rate *= s
        else:
            stride, dil = s, 1
            cur *= s
        for i in range(n):
            plan.append((idx, inp, c, stride if i == 0 else 1, dil if i == 0 else rate, t))
            inp = c
            idx += 1
    return plan


def mobilenetv2_features(sd, x, p='encoder'):
    """networks/encoders/mobilenetv2.py:219-224; stages = features[0:4],[4:7],[7:14],[14:]."""
    def cbr(x, q, stride=1, pad=0, dil=1, groups=1):
        return F.relu6(_fbn(_conv(x, sd, q + '.0', stride, pad, dil, groups), sd, q + '.1'))
    x = cbr(x, p + '.features.0', 2, 1)
    outs = {}
    for idx, inp, oup, stride, dil, t in mobilenetv2_plan():
        q = '%s.features.%d.conv' % (p, idx)
        hid = int(round(inp * t))
        y, j = x, 0
        if t != 1:
            y = cbr(y, q + '.0')
            j = 1
        y = cbr(y, '%s.%d' % (q, j), stride, dil, dil, hid)
        y = _fbn(_conv(y, sd, '%s.%d' % (q, j + 1)), sd, '%s.%d' % (q, j + 2))
        x = x + y if (stride == 1 and inp == oup) else y
        outs[idx] = x
    x = cbr(x, p + '.features.18')
    return [outs[3], outs[6], outs[13], x]


def swin_features(sd, x, p='encoder', depths=(2, 2, 18), heads=(4, 8, 16), ws=7):
    """Swin-B trunk, 3 stages (networks/encoders/swin/swin_transformer.py:684-716, build.py:11-27)."""
    B, _, H, W = x.shape
    if W % 4:
        x = F.pad(x, (0, 4 - W % 4))                                                    # PatchEmbed :474-481
    if H % 4:
        x = F.pad(x, (0, 0, 0, 4 - H % 4))
    x = _conv(x, sd, p + '.patch_embed.proj', 4)
    h, w = x.shape[2:]
    x = _ln(x.flatten(2).transpose(1, 2), sd, p + '.patch_embed.norm')                  # [B, hw, C]
    outs = []
    for li, (depth, nh) in enumerate(zip(depths, heads)):
        C = x.shape[-1]
        hp, wp = -(-h // ws) * ws, -(-w // ws) * ws
        shift = ws // 2
        # BasicLayer.forward :392-411: region labels of the shifted map -> additive -100 mask
        img = torch.zeros(1, hp, wp, 1)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, hp // ws, ws, wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
        amask = mw.unsqueeze(1) - mw.unsqueeze(2)
        amask = amask.masked_fill(amask != 0, -100.0).masked_fill(amask == 0, 0.0).to(x.dtype)
        for bi in range(depth):
            q = '%s.layers.%d.blocks.%d' % (p, li, bi)
            sh = 0 if bi % 2 == 0 else shift
            # SwinTransformerBlock.forward :262-318
            y = _ln(x, sd, q + '.norm1').view(B, h, w, C)
            y = F.pad(y, (0, 0, 0, wp - w, 0, hp - h))
            if sh:
                y = torch.roll(y, shifts=(-sh, -sh), dims=(1, 2))
            win = y.view(B, hp // ws, ws, wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
            # WindowAttention.forward :159-199
            nW, Nt, _ = win.shape
            qkv = _lin(win, sd, q + '.attn.qkv').reshape(nW, Nt, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
            att = (qkv[0] * (C // nh) ** -0.5) @ qkv[1].transpose(-2, -1)
            rpb = sd[q + '.attn.relative_position_bias_table'][sd[q + '.attn.relative_position_index'].view(-1)]
            att = att + rpb.view(Nt, Nt, -1).permute(2, 0, 1).unsqueeze(0)
            if sh:
                att = att.view(B, nW // B, nh, Nt, Nt) + amask.unsqueeze(1).unsqueeze(0)
                att = att.view(-1, nh, Nt, Nt)
            att = torch.softmax(att, -1)
            o = (att @ qkv[2]).transpose(1, 2).reshape(nW, Nt, C)
            o = _lin(o, sd, q + '.attn.proj')
            o = o.view(B, hp // ws, wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, hp, wp, C)
            if sh:
                o = torch.roll(o, shifts=(sh, sh), dims=(1, 2))
            x = x + o[:, :h, :w, :].reshape(B, h * w, C)
            x = x + _lin(F.gelu(_lin(_ln(x, sd, q + '.norm2'), sd, q + '.mlp.fc1')), sd, q + '.mlp.fc2')   # Mlp :56-62
        xo = _ln(x, sd, '%s.norm%d' % (p, li))
        outs.append(xo.view(B, h, w, C).permute(0, 3, 1, 2).contiguous())
        if li < len(depths) - 1:                                                         # PatchMerging :338-359
            q = '%s.layers.%d.downsample' % (p, li)
            y = x.view(B, h, w, C)
            if h % 2 or w % 2:
                y = F.pad(y, (0, 0, 0, w % 2, 0, h % 2))
            y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
            h, w = (h + 1) // 2, (w + 1) // 2
            x = F.linear(_ln(y.view(B, h * w, 4 * C), sd, q + '.norm'), sd[q + '.reduction.weight'])
    outs.append(outs[-1])
    return outs


# --------------------------------------------------------------------------
# attention cores
# --------------------------------------------------------------------------
def mha_core(Q, K, V, H, max_mem_len_ratio=-1., top_k=-1):
    """MultiheadAttention.forward, networks/layers/attention.py:82-117, after the
    optional input linears and before ``projection``.  Q [Tq,B,C], K,V [Tk,B,C].
    Eval-time knobs: max_mem_len_ratio (:84-89) rescales Q for banks longer than
    ratio x the query length; top_k (:102-105) keeps the k largest scores per row."""
    Tq, B, C = Q.shape
    d = C // H
    Q = Q / (d ** 0.5)                                            # :82
    if max_mem_len_ratio > 0:                                     # :84-89
        mem_len_ratio = float(K.shape[0]) / Tq
        if mem_len_ratio > max_mem_len_ratio:
            Q = Q * (math.log(mem_len_ratio) / math.log(max_mem_len_ratio))
    q = Q.view(Tq, B, H, d).permute(1, 2, 0, 3)
    k = K.view(-1, B, H, d).permute(1, 2, 3, 0)
    v = V.view(-1, B, H, V.shape[-1] // H).permute(1, 2, 0, 3)
    qk = q @ k                                                    # :97
    if 0 < top_k < qk.shape[-1]:                                  # :102-105
        top, idx = torch.topk(qk, k=top_k, dim=-1)
        a = torch.zeros_like(qk).scatter_(-1, idx, torch.softmax(top, dim=-1))
    else:
        a = torch.softmax(qk, dim=-1)                             # :107
    return (a @ v).permute(2, 0, 1, 3).reshape(Tq, B, -1)         # :113-117


def local_window_scores(q2d, k2d, H, max_dis=7):
    """Windowed q.k of MultiheadLocalAttentionV2 / LocalGatedPropagation
    (attention.py:341-348, 828-835): returns s[B,H,W2,h,w] (0 outside the image)
    and the in-image mask [W2,h,w].  q2d must already be scaled."""
    B, C, h, w = q2d.shape
    d = C // H
    ws = 2 * max_dis + 1
    q = q2d.view(B, H, d, h, w)
    kp = F.pad(k2d.view(B, H, d, h, w), (max_dis, max_dis, max_dis, max_dis))
    s = q.new_zeros(B, H, ws * ws, h, w)
    valid = torch.zeros(ws * ws, h, w, dtype=torch.bool)
    ys = torch.arange(h).view(h, 1)
    xs = torch.arange(w).view(1, w)
    for dy in range(ws):
        for dx in range(ws):
            s[:, :, dy * ws + dx] = (q * kp[:, :, :, dy:dy + h, dx:dx + w]).sum(2)
            ky, kx = ys + dy - max_dis, xs + dx - max_dis
            valid[dy * ws + dx] = (ky >= 0) & (ky < h) & (kx >= 0) & (kx < w)
    return s, valid


def local_window_aggregate(a, v2d, H, max_dis=7):
    """sum_w a[B,H,W2,h,w] * v[B,H,dv,(y,x)+delta(w)] -> [B,H,dv,h,w]
    (what local2global + the dense matmul compute, attention.py:366-368, 850-853)."""
    B, C, h, w = v2d.shape
    dv = C // H
    ws = 2 * max_dis + 1
    vp = F.pad(v2d.view(B, H, dv, h, w), (max_dis, max_dis, max_dis, max_dis))
    o = v2d.new_zeros(B, H, dv, h, w)
    for dy in range(ws):
        for dx in range(ws):
            o += a[:, :, dy * ws + dx].unsqueeze(2) * vp[:, :, :, dy:dy + h, dx:dx + w]
    return o


def aot_local_attention(sd, p, q2d, k2d, v2d, H):
    """MultiheadLocalAttentionV2.forward with use_linear=False (attention.py:308-376)."""
    B, C, h, w = v2d.shape
    d = C // H
    rel = F.conv2d(q2d, sd[p + '.relative_emb_k.weight'], sd[p + '.relative_emb_k.bias'], groups=H)  # :327 (unscaled q)
    rel = rel.view(B, H, 225, h, w)
    s, valid = local_window_scores(q2d / (d ** 0.5), k2d, H)       # :330-348
    s = s + rel                                                    # :355
    s = s.masked_fill(~valid.view(1, 1, 225, h, w), float('-inf'))  # :357 (-1e8 -> exact 0 after softmax)
    a = torch.softmax(s, dim=2)                                    # :359
    bias = torch.einsum('bhwn,hcw->bhcn', a.view(B, H, 225, h * w), sd[p + '.relative_emb_v'])  # :363-364
    o = local_window_aggregate(a, v2d, H).view(B, H, d, h * w) + bias
    o = o.permute(3, 0, 1, 2).reshape(h * w, B, C)                 # :370-371
    return _lin(o, sd, p + '.projection')                          # :373


def dwconv5(x, sd, p, size_2d):
    # DWConv2d / GNActDWConv2d conv, basic.py:27-35,50-57 (5x5 depthwise, pad 2, no bias)
    h, w = size_2d
    _, B, C = x.shape
    x = x.view(h, w, B, C).permute(2, 3, 0, 1)
    x = F.conv2d(x, sd[p + '.weight'], None, 1, 2, 1, C)
    return x.reshape(B, C, h * w).permute(2, 0, 1)


def gated_propagation(sd, p, Q, K, V, U, size_2d, H, use_linear, d_att, max_mem_len_ratio=-1., top_k=-1):
    """GatedPropagation.forward, attention.py:636-712.  Eval-time knobs: max_mem_len_ratio (:674-679) rescales Q for banks
    longer than ratio x the query length; top_k (:689-693) keeps the k largest scores per row."""
    L, B, _ = Q.shape
    if use_linear:
        Q = K = _lin(Q, sd, p + '.linear_QK')                      # :649
        half = V.shape[-1] // 2

        def cat(a, b):                                             # :651-659
            if H > 1:
                hd = (a.shape[-1] + b.shape[-1]) // H
                a = a.view(-1, B, H, hd // 2)
                b = b.view(-1, B, H, hd // 2)
                return torch.cat([a, b], -1).view(-1, B, H * hd)
            return torch.cat([a, b], -1)
        V = silu(cat(_lin(V[..., :half], sd, p + '.linear_V1'), _lin(V[..., half:], sd, p + '.linear_V2')))
        U = silu(cat(_lin(U[..., :half], sd, p + '.linear_U1'), _lin(U[..., half:], sd, p + '.linear_U2')))
    Q = Q / (d_att ** 0.5)                                         # :672
    if max_mem_len_ratio > 0:                                      # :674-679
        mem_len_ratio = float(K.shape[0]) / Q.shape[0]
        if mem_len_ratio > max_mem_len_ratio:
            Q = Q * (math.log(mem_len_ratio) / math.log(max_mem_len_ratio))
    q = Q.view(-1, B, H, d_att).permute(1, 2, 0, 3)
    k = K.view(-1, B, H, d_att).permute(1, 2, 3, 0)
    v = V.view(-1, B, H, V.shape[-1] // H).permute(1, 2, 0, 3)
    qk = q @ k                                                     # :687
    if 0 < top_k < qk.shape[-1]:                                   # :689-693
        top, idx = torch.topk(qk, k=top_k, dim=-1)
        a = torch.zeros_like(qk).scatter_(-1, idx, torch.softmax(top, -1))
    else:
        a = torch.softmax(qk, -1)                                  # :697
    o = (a @ v).permute(2, 0, 1, 3).reshape(L, B, -1) * U          # :703-707
    o = dwconv5(o, sd, p + '.dw_conv.conv', size_2d)               # :709
    return _lin(o, sd, p + '.projection')                          # :710


def local_gated_propagation(sd, p, q2d, k2d, v2d, u, size_2d, H, d_att):
    """LocalGatedPropagation.forward with use_linear=False, attention.py:789-861."""
    B, C, h, w = v2d.shape
    rel = F.conv2d(q2d, sd[p + '.relative_emb_k.weight'], sd[p + '.relative_emb_k.bias'], groups=H)  # :814
    rel = rel.view(B, H, 225, h, w)
    s, valid = local_window_scores(q2d / (d_att ** 0.5), k2d, H)   # :817-835
    s = (s + rel).masked_fill(~valid.view(1, 1, 225, h, w), float('-inf'))  # :842-844
    a = torch.softmax(s, dim=2)                                    # :846
    o = local_window_aggregate(a, v2d, H).view(B, C, h * w).permute(2, 0, 1)  # :850-853
    o = o * u                                                      # :855
    o = dwconv5(o, sd, p + '.dw_conv.conv', size_2d)               # :857
    return _lin(o, sd, p + '.projection')                          # :858


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
class OracleModel:
    """Functional twin of networks/models/aot.py:9-115 and deaot.py:8-55."""

    def __init__(self, spec, state_dict, dtype=torch.float32):
        self.spec = dict(SPECS[spec]) if isinstance(spec, str) else dict(spec)
        self.sd = {k: (v.detach().to(dtype) if v.is_floating_point() else v.detach().clone())
                   for k, v in state_dict.items()}
        self.dtype = dtype
        self.max_obj_num = self.spec['max_obj']
        self.deaot = self.spec['vos'] == 'deaot'
        self.trace = None      # set to a dict to record named intermediates (per-kernel parity tests)

    def _t(self, name, value):
        if self.trace is not None:
            self.trace[name] = value

    # aot.py:81-84
    def encode_image(self, img):
        img = img.to(self.dtype)
        f = {'resnet50': resnet50_features, 'swin_base': swin_features,
             'resnet101': lambda sd, x: resnet50_features(sd, x, blocks=(3, 4, 23))}.get(self.spec['encoder'], mobilenetv2_features)
        xs = f(self.sd, img)
        xs[-1] = _conv(xs[-1], self.sd, 'encoder_projector')
        return xs

    # position.py:49-74 with num_pos_feats = emb/2, normalize=True (aot.py:67-68)
    def get_pos_emb(self, x):
        _, _, h, w = x.shape
        npf = self.spec['emb'] // 2
        y = torch.arange(h, dtype=torch.float32).view(h, 1).expand(h, w)
        xx = torch.arange(w, dtype=torch.float32).view(1, w).expand(h, w)
        y = y / (y[-1:, :] + 1e-6) * (2 * math.pi)
        xx = xx / (xx[:, -1:] + 1e-6) * (2 * math.pi)
        dim_t = torch.arange(npf, dtype=torch.float32)
        dim_t = 10000 ** (2 * (dim_t // 2) / npf)
        px = xx[:, :, None] / dim_t
        py = y[:, :, None] / dim_t
        px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
        py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
        return torch.cat((py, px), dim=2).permute(2, 0, 1).unsqueeze(0).to(self.dtype)

    # aot.py:76-79 / deaot.py:51-55
    def get_id_emb(self, one_hot):
        if self.spec['align_corners']:
            e = _conv(one_hot.to(self.dtype), self.sd, 'patch_wise_id_bank', 16, 8)
        else:
            e = _conv(one_hot.to(self.dtype), self.sd, 'patch_wise_id_bank', 16, 0)
        if self.deaot:
            e = _ln(e.permute(2, 3, 0, 1), self.sd, 'id_norm').permute(2, 3, 0, 1)
        return e

    # ---- AOT block: transformer.py:312-367 -------------------------------
    def _aot_block(self, i, tgt, long_mem, short_mem, id_emb, pos, size_2d):
        sd, H = self.sd, self.spec['heads']
        p = 'LSTT.layers.%d' % i
        x1 = _ln(tgt, sd, p + '.norm1')
        qk = x1 + pos
        sa = mha_core(_lin(qk, sd, p + '.self_attn.linear_Q'), _lin(qk, sd, p + '.self_attn.linear_K'),
                      _lin(x1, sd, p + '.self_attn.linear_V'), H)
        self._t('L%d.self_core' % i, sa)
        tgt = tgt + _lin(sa, sd, p + '.self_attn.projection')
        self._t('L%d.after_self' % i, tgt)
        x2 = _ln(tgt, sd, p + '.norm2')
        cQ = _lin(x2, sd, p + '.linear_Q')
        cK, cV = cQ, x2
        lQ = seq_to_2d(cQ, size_2d)
        if id_emb is not None:
            gK, gV = cK, _lin(cV + id_emb, sd, p + '.linear_V')    # fuse_key_value_id :364-367
            lK, lV = seq_to_2d(gK, size_2d), seq_to_2d(gV, size_2d)
        else:
            gK, gV = long_mem
            lK, lV = short_mem
        lt_core = mha_core(cQ, gK, gV, H, self.spec.get('lt_max_mem_len_ratio', -1.), self.spec.get('lt_top_k', -1))
        lt = _lin(lt_core, sd, p + '.long_term_attn.projection')
        st = aot_local_attention(sd, p + '.short_term_attn', lQ, lK, lV, H)
        self._t('L%d.curr_Q' % i, cQ)
        self._t('L%d.lt_core' % i, lt_core)
        self._t('L%d.lt' % i, lt)
        self._t('L%d.st' % i, st)
        tgt = tgt + lt + st
        self._t('L%d.after_lst' % i, tgt)
        x3 = _ln(tgt, sd, p + '.norm3')
        f = _lin(x3, sd, p + '.linear1')
        h, w = size_2d
        _, B, C = f.shape
        f2 = f.view(h, w, B, C).permute(2, 3, 0, 1)
        f2 = F.gelu(F.group_norm(f2, 32, sd[p + '.activation.gn.weight'], sd[p + '.activation.gn.bias'], 1e-5))
        f2 = F.conv2d(f2, sd[p + '.activation.conv.weight'], None, 1, 2, 1, C)   # basic.py:27-35
        f = f2.reshape(B, C, h * w).permute(2, 0, 1)
        self._t('L%d.ffn_dw' % i, f)
        tgt = tgt + _lin(f, sd, p + '.linear2')
        self._t('L%d.out' % i, tgt)
        return tgt, [[cK, cV], [gK, gV], [lK, lV]]

    # ---- DeAOT block: transformer.py:582-665 ------------------------------
    def _gpm_block(self, i, tgt, tgt_id, long_mem, short_mem, id_emb, pos, size_2d):
        sd, H, D = self.sd, self.spec['heads'], self.spec['emb']
        p = 'LSTT.layers.%d' % i
        d_att = D // 2 if H == 1 else D // H
        E = 2 * D
        x1 = _ln(tgt, sd, p + '.norm1')
        QV = _lin(x1, sd, p + '.linear_QV')
        cQ = cK = QV[..., :d_att * H]
        cV = silu(QV[..., d_att * H:])
        lQ = seq_to_2d(cQ, size_2d)
        cU = _lin(x1, sd, p + '.linear_U')
        if tgt_id is None:
            tgt_id = 0
            U = torch.cat([silu(cU), torch.ones_like(cU)], -1)
            cIDV = None
        else:
            xi = _ln(tgt_id, sd, p + '.id_norm1')
            cIDV = xi
            U = silu(torch.cat([cU, _lin(xi, sd, p + '.linear_ID_U')], -1))
        if id_emb is not None:
            gK, gV = cK, cV
            lK, lV = seq_to_2d(gK, size_2d), seq_to_2d(gV, size_2d)
            gIDV = self.fuse_id(i, cIDV, id_emb)
            lIDV = seq_to_2d(gIDV, size_2d)
        else:
            gK, gV, _, gIDV = long_mem
            lK, lV, _, lIDV = short_mem
        lt = gated_propagation(sd, p + '.long_term_attn', cQ, gK, torch.cat([gV, gIDV], -1), U, size_2d, H, False, d_att,
                               self.spec.get('lt_max_mem_len_ratio', -1.), self.spec.get('lt_top_k', -1))
        st = local_gated_propagation(sd, p + '.short_term_attn', lQ, lK, torch.cat([lV, lIDV], 1), U, size_2d, H, d_att)
        both = lt + st
        tgt = tgt + both[..., :D]
        tgt_id = tgt_id + both[..., D:]
        z = torch.cat([_ln(tgt, sd, p + '.norm2'), _ln(tgt_id, sd, p + '.id_norm2')], -1)
        sa = gated_propagation(sd, p + '.self_attn', z, z, z, z, size_2d, H, True, d_att)
        tgt = tgt + sa[..., :D]
        tgt_id = tgt_id + sa[..., D:]
        return tgt, tgt_id, [[cK, cV, None, cIDV], [gK, gV, None, gIDV], [lK, lV, None, lIDV]]

    def fuse_id(self, i, idv, id_emb):
        # GatedPropagationModule.fuse_key_value_id, transformer.py:659-665
        p = 'LSTT.layers.%d.linear_ID_V' % i
        x = id_emb if idv is None else torch.cat([idv, id_emb], 2)
        return silu(_lin(x, self.sd, p))

    def fuse_kv(self, i, k, v, id_emb):
        # LongShortTermTransformerBlock.fuse_key_value_id, transformer.py:364-367
        return k, _lin(v + id_emb, self.sd, 'LSTT.layers.%d.linear_V' % i)

    # aot.py:94-108 + transformer.py:95-140 / 205-255
    def LSTT_forward(self, curr_embs, long_mems, short_mems, curr_id_emb=None, pos_emb=None, size_2d=(30, 30)):
        n, c, h, w = curr_embs[-1].shape
        x = curr_embs[-1].view(n, c, h * w).permute(2, 0, 1)
        L = self.spec['lstt_num']
        outs, mems = [], []
        x_id = None
        for i in range(L):
            lm = long_mems[i] if long_mems is not None else None
            sm = short_mems[i] if short_mems is not None else None
            if self.deaot:
                x, x_id, m = self._gpm_block(i, x, x_id, lm, sm, curr_id_emb, pos_emb, size_2d)
                outs.append(torch.cat([x, x_id], 2))
            else:
                x, m = self._aot_block(i, x, lm, sm, curr_id_emb, pos_emb, size_2d)
                outs.append(x)
            mems.append(m)
        sd = self.sd
        if self.deaot:
            def norm(t, j):   # GroupNorm1D(512, 2 groups), basic.py:6-12
                return F.group_norm(t.permute(1, 2, 0), 2, sd['LSTT.decoder_norms.%d.gn.weight' % j],
                                    sd['LSTT.decoder_norms.%d.gn.bias' % j], 1e-5).permute(2, 0, 1)
        else:
            def norm(t, j):
                return _ln(t, sd, 'LSTT.decoder_norms.%d' % j)
        nn_norm = sum(1 for k in sd if k.startswith('LSTT.decoder_norms.') and k.endswith('.weight'))
        outs[-1] = norm(outs[-1], nn_norm - 1)                     # final_norm
        if self.spec['intermediate_lstt']:
            for j in range(L - 1):
                outs[j] = norm(outs[j], j)
        curr, long_, short_ = zip(*mems)
        return outs, list(curr), list(long_), list(short_)

    # aot.py:86-92 / deaot.py:43-49 + decoders/fpn.py:34-58
    def decode_id_logits(self, lstt_embs, shortcuts):
        sd, ac = self.sd, self.spec['align_corners']
        n, c, h, w = shortcuts[-1].shape
        ins = [shortcuts[-1]] + [e.view(h, w, n, -1).permute(2, 3, 0, 1) for e in lstt_embs]
        x = torch.cat(ins, 1) if self.spec['intermediate_lstt'] else ins[-1]

        def convgn(x, q, k):
            return F.group_norm(_conv(x, sd, 'decoder.%s.conv' % q, 1, k // 2), 8,
                                sd['decoder.%s.gn.weight' % q], sd['decoder.%s.gn.bias' % q], 1e-5)
        x = F.relu(convgn(x, 'conv_in', 1))
        x = F.relu(convgn(_conv(shortcuts[-2], sd, 'decoder.adapter_16x') + x, 'conv_16x', 3))
        x = F.interpolate(x, size=shortcuts[-3].shape[-2:], mode='bilinear', align_corners=ac)
        x = F.relu(convgn(_conv(shortcuts[-3], sd, 'decoder.adapter_8x') + x, 'conv_8x', 3))
        x = F.interpolate(x, size=shortcuts[-4].shape[-2:], mode='bilinear', align_corners=ac)
        x = F.relu(convgn(_conv(shortcuts[-4], sd, 'decoder.adapter_4x') + x, 'conv_4x', 3))
        return _conv(x, sd, 'decoder.conv_out')


# --------------------------------------------------------------------------
# engine (single <=10-object group): networks/engines/aot_engine.py:127-482,
# networks/engines/deaot_engine.py:20-56
# --------------------------------------------------------------------------
class OracleEngine:
    def __init__(self, model, long_term_mem_gap=None, short_term_mem_skip=1, long_term_mem_max=None):
        self.AOT = model
        self.long_term_mem_gap = model.spec['mem_gap'] if long_term_mem_gap is None else long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        self.long_term_mem_max = long_term_mem_max     # repo extension: bounded bank, see _update_long
        self.restart_engine()

    def restart_engine(self):                                      # aot_engine.py:445-477
        self.frame_step = 0
        self.last_mem_step = -1
        self.obj_nums = None
        self.pos_emb = None
        self.enc_size_2d = self.enc_hw = self.input_size_2d = None
        self.long_term_memories = None
        self.short_term_memories_list = []
        self.short_term_memories = None
        self.curr_enc_embs = None
        self.trace = {}

    def _id_emb(self, mask):                                       # assign_identity :168-179
        # a map with max_obj+1 channels is a probability map fed back as it is (:309-313, MODEL_USE_PREV_PROB)
        prob = mask.dim() == 4 and mask.shape[1] == self.AOT.max_obj_num + 1
        oh = mask.to(torch.get_default_dtype()) if prob else one_hot_mask(mask, self.AOT.max_obj_num)
        return self.AOT.get_id_emb(oh).view(1, -1, self.enc_hw).permute(2, 0, 1)

    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1, img_embs=None):   # :188-251
        self.obj_nums = obj_nums if isinstance(obj_nums, (list, tuple)) else [obj_nums]
        embs = self.AOT.encode_image(img) if img_embs is None else img_embs
        if self.input_size_2d is None:
            self.input_size_2d = tuple(img.shape[2:])
            self.enc_size_2d = tuple(embs[-1].shape[2:])
            self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]
        self.curr_enc_embs = embs
        if self.pos_emb is None:
            self.pos_emb = self.AOT.get_pos_emb(embs[-1]).view(1, -1, self.enc_hw).permute(2, 0, 1)
        id_emb = self._id_emb(mask)
        self.curr_id_emb = id_emb
        self.curr_lstt_output = self.AOT.LSTT_forward(embs, None, None, id_emb, self.pos_emb, self.enc_size_2d)
        _, _, long_m, short_m = self.curr_lstt_output
        if self.long_term_memories is None:
            self.long_term_memories = [list(m) for m in long_m]
        else:
            self._update_long(long_m)
        self.last_mem_step = self.frame_step
        self.short_term_memories_list = [short_m]
        self.short_term_memories = short_m

    def _update_long(self, new):                                   # :291-305 (appended, not prepended)
        upd = []
        for nm, om in zip(new, self.long_term_memories):
            upd.append([None if (a is None or b is None) else torch.cat([b, a], 0) for a, b in zip(nm, om)])
        if self.long_term_mem_max is not None:
            # bounded bank (not in the reference, which grows without limit): keep the first memorised frame and the
            # most recent long_term_mem_max - 1 others, i.e. drop the oldest non-first frame
            N = self.enc_hw
            upd = [[None if t is None else (t if t.shape[0] <= self.long_term_mem_max * N else
                                            torch.cat([t[:N], t[2 * N:]], 0)) for t in layer] for layer in upd]
        self.long_term_memories = upd

    def match_propogate_one_frame(self, img, img_embs=None):       # :340-354
        self.frame_step += 1
        self.curr_enc_embs = self.AOT.encode_image(img) if img_embs is None else img_embs
        self.curr_lstt_output = self.AOT.LSTT_forward(self.curr_enc_embs, self.long_term_memories,
                                                      self.short_term_memories, None, self.pos_emb,
                                                      self.enc_size_2d)

    def decode_current_logits(self, output_size=None):             # :356-380
        logits = self.AOT.decode_id_logits(self.curr_lstt_output[0], self.curr_enc_embs)
        for b, n in enumerate(self.obj_nums):
            logits[b, n + 1:] = -1e10
        self.pred_id_logits = logits
        if output_size is not None:
            logits = F.interpolate(logits, size=output_size, mode='bilinear',
                                   align_corners=self.AOT.spec['align_corners'])
        return logits

    def update_memory(self, mask, skip_long_term_update=False):    # :307-338 / deaot_engine.py:20-56
        id_emb = self._id_emb(mask)
        self.curr_id_emb = id_emb
        curr = self.curr_lstt_output[1]
        curr2d = []
        for i in range(len(curr)):
            if self.AOT.deaot:
                k, v, _, idv = curr[i]
                idv = self.AOT.fuse_id(i, idv, id_emb)
                curr[i][3] = idv
                curr2d.append([seq_to_2d(k, self.enc_size_2d), seq_to_2d(v, self.enc_size_2d), None,
                               seq_to_2d(idv, self.enc_size_2d)])
            else:
                k, v = self.AOT.fuse_kv(i, curr[i][0], curr[i][1], id_emb)
                curr[i][0], curr[i][1] = k, v
                curr2d.append([seq_to_2d(k, self.enc_size_2d), seq_to_2d(v, self.enc_size_2d)])
        self.short_term_memories_list.append(curr2d)
        self.short_term_memories_list = self.short_term_memories_list[-self.short_term_mem_skip:]
        self.short_term_memories = self.short_term_memories_list[0]
        if self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            if not skip_long_term_update:
                self._update_long(curr)
            self.last_mem_step = self.frame_step


class OracleInferEngine:
    """AOTInferEngine (aot_engine.py:485-635): one OracleEngine per group of max_obj objects, image embedding
    computed once per frame and shared, logits merged by soft aggregation."""

    def __init__(self, model, long_term_mem_gap=None, max_aot_obj_num=None):
        self.AOT = model
        self.gap = long_term_mem_gap
        self.max_aot_obj_num = model.max_obj_num if max_aot_obj_num is None else min(max_aot_obj_num, model.max_obj_num)
        self.restart_engine()

    def restart_engine(self):
        self.aot_engines = []
        self.obj_nums = None

    def separate_mask(self, mask, obj_nums):                       # :515-545 (label-map branch)
        n = len(self.aot_engines)
        if n == 1:
            return [mask], [obj_nums]
        nums = [self.max_aot_obj_num] * n
        if obj_nums % self.max_aot_obj_num > 0:
            nums[-1] = obj_nums % self.max_aot_obj_num
        outs = []
        for i in range(n):
            lo, hi = i * self.max_aot_obj_num + 1, (i + 1) * self.max_aot_obj_num
            fg = ((mask >= lo) & (mask <= hi)).to(mask.dtype)
            outs.append((fg * mask - lo + 1) * fg)
        return outs, nums

    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1):   # :584-609
        if isinstance(obj_nums, (list, tuple)):
            obj_nums = obj_nums[0]
        self.obj_nums = obj_nums
        n = max(-(-obj_nums // self.max_aot_obj_num), 1)
        while n > len(self.aot_engines):
            self.aot_engines.append(OracleEngine(self.AOT, self.gap))
        masks, nums = self.separate_mask(mask, obj_nums)
        embs = None
        for e, m, k in zip(self.aot_engines, masks, nums):
            e.add_reference_frame(img, m, [k], frame_step, img_embs=embs)
            embs = e.curr_enc_embs
        self.input_size_2d = self.aot_engines[0].input_size_2d

    def match_propogate_one_frame(self, img):                      # :611-616
        embs = None
        for e in self.aot_engines:
            e.match_propogate_one_frame(img, img_embs=embs)
            embs = e.curr_enc_embs

    def decode_current_logits(self, output_size=None):             # :618-623 + soft_logit_aggregation :565-582
        logits = [e.decode_current_logits(output_size) for e in self.aot_engines]
        if len(logits) == 1:
            return logits[0]
        fg, bg = [], []
        for lg in logits:
            pr = torch.softmax(lg, dim=1)
            bg.append(pr[:, 0:1])
            fg.append(pr[:, 1:1 + self.max_aot_obj_num])
        bgp = torch.prod(torch.cat(bg, 1), dim=1, keepdim=True)
        return torch.logit(torch.cat([bgp] + fg, 1).clamp(1e-5, 1 - 1e-5))

    def update_memory(self, mask, skip_long_term_update=False):    # :625-630
        masks, _ = self.separate_mask(mask, self.obj_nums)
        for e, m in zip(self.aot_engines, masks):
            e.update_memory(m, skip_long_term_update)


# --------------------------------------------------------------------------
# training-step forward (networks/engines/aot_engine.py:33-108) and its losses
# --------------------------------------------------------------------------
def ce_topk_loss(logits, label, step, top_k_percent_pixels=0.15, hard_example_mining_step=50000.):
    """networks/layers/loss.py:137-188 for one sample: logits [1,C,H,W], label [1,H,W] (255 = ignore) -> [1].  Mean of the
    top-k per-pixel cross entropies, k annealed from all pixels to top_k_percent_pixels over the mining steps."""
    px = F.cross_entropy(logits.flatten(2), label.flatten(1).long(), ignore_index=255, reduction='none')
    if top_k_percent_pixels is None:
        return (px.sum() / (label != 255).sum()).view(1)
    n = float(px.shape[1])
    mining = hard_example_mining_step + 1e-5
    ratio = min(1.0, step / float(mining))
    k = int((ratio * top_k_percent_pixels + (1.0 - ratio)) * n)
    return torch.topk(px, k=k, dim=1)[0].mean().view(1)


def soft_jaccard_loss(logits, label, eps=1e-6):
    """loss.py:26-52,119-135 (tversky, alpha = beta = 1) for one sample: 1 - soft IoU averaged over the classes that
    own at least one valid pixel."""
    C = logits.shape[1]
    prob = torch.softmax(logits, 1).permute(0, 2, 3, 1).reshape(-1, C)
    lab = label.reshape(-1)
    valid = lab != 255
    prob, lab = prob[valid], lab[valid]
    losses = []
    for c in range(C):
        fg = (lab == c).to(prob.dtype)
        if fg.sum() == 0:
            continue
        p0 = prob[:, c]
        num = (p0 * fg).sum()
        den = num + (p0 * (1 - fg)).sum() + ((1 - p0) * fg).sum()
        losses.append(1 - num / (den + eps))
    return (sum(losses) / len(losses)).view(1)


def train_forward(model, all_frames, all_masks, obj_nums, step, cfg, use_prev_pred=False, enable_prev_frame=False,
                  use_prev_prob=False, perms=None, long_term_mem_gap=9999):
    """aot_engine.py:33-108 sample by sample (every op of that path is per-sample; the reference runs them batched).
    all_frames [T*bs,3,H,W] / all_masks [T*bs,1,H,W] time-major; cfg: the five TRAIN_* values of _init_losses (:110-125);
    perms[b][o] = channel identity o of sample b is shuffled to (:168-172, reversed on the logits :364-367).
    Returns (loss, frame_loss [T, bs], masks [T, bs, H, W])."""
    bs = len(obj_nums)
    T = all_frames.shape[0] // bs
    L = model.max_obj_num + 1
    mining = cfg['TRAIN_HARD_MINING_RATIO'] * cfg['TRAIN_TOTAL_STEPS']
    aux_step = cfg['TRAIN_TOTAL_STEPS'] * cfg['TRAIN_AUX_LOSS_RATIO'] + 1e-5
    aux_weight = cfg['TRAIN_AUX_LOSS_WEIGHT'] * max(aux_step - step, 0.) / aux_step
    n_aux = 2 if enable_prev_frame else 1
    frame_loss = torch.zeros(T, bs)
    masks = torch.zeros(T, bs, *all_masks.shape[-2:], dtype=torch.long)
    for b in range(bs):
        eng = OracleEngine(model, long_term_mem_gap=long_term_mem_gap)     # cfg.TRAIN_LONG_TERM_MEM_GAP (2 for the L models)
        objs = int(obj_nums[b])
        perm = None if perms is None else perms[b]
        inv = None if perm is None else torch.argsort(perm)

        def ident(m):                                   # what assign_identity sees: the shuffled one-hot / prob map
            oh = m if m.shape[1] == L else one_hot_mask(m, model.max_obj_num)
            return oh if perm is None else oh[:, inv]   # new channel t <- old channel o with perm[o] = t

        def score(t):
            gt = all_masks[t * bs + b:t * bs + b + 1]
            lg = eng.decode_current_logits()
            if perm is not None:
                lg = lg[:, perm]
                lg[:, objs + 1:] = -1e10
                eng.pred_id_logits = lg
            lg = F.interpolate(lg, size=gt.shape[-2:], mode='bilinear', align_corners=model.spec['align_corners'])
            sl = lg[:, :objs + 1]
            frame_loss[t, b] = (0.5 * ce_topk_loss(sl, gt[:, 0], step, cfg['TRAIN_TOP_K_PERCENT_PIXELS'], mining)
                                + 0.5 * soft_jaccard_loss(sl, gt[:, 0]))[0]
            masks[t, b] = lg.argmax(1)[0]
            return torch.softmax(lg, 1) if use_prev_prob else lg.argmax(1, keepdim=True).float()

        img = lambda t: all_frames[t * bs + b:t * bs + b + 1]
        gtm = lambda t: all_masks[t * bs + b:t * bs + b + 1]
        eng.add_reference_frame(img(0), ident(gtm(0)), [L - 1 if perm is not None else objs], frame_step=0)
        score(0)
        t = 1
        if enable_prev_frame:                           # set_prev_frame :253-289
            eng.frame_step = 1
            eng.add_reference_frame(img(1), ident(gtm(1)), eng.obj_nums)
            score(1)
            t = 2
        while t < T:
            eng.match_propogate_one_frame(img(t))
            pred = score(t)
            if t < T - 1:
                eng.update_memory(ident(pred if use_prev_pred else gtm(t)))
            t += 1
    loss = aux_weight * frame_loss[:n_aux].reshape(-1).mean() + frame_loss[n_aux:].reshape(-1).mean()
    return loss, frame_loss, masks


def run_clip(engine, frames, first_mask, obj_nums, output_size, teacher_masks=None, keep=('logits4',)):
    """The demo loop, tools/demo.py:187-235, on in-memory tensors.

    frames: list of [1,3,H,W]; first_mask [1,1,H,W] float labels.  Returns a list
    (one entry per propagated frame) of dicts with 'mask' (uint8 [h,w] at
    output_size), optionally 'logits4' (stride-4 logits) and 'logits'."""
    out = []
    engine.restart_engine()
    with torch.no_grad():
        engine.add_reference_frame(frames[0], first_mask, obj_nums, frame_step=0)
        for t in range(1, len(frames)):
            engine.match_propogate_one_frame(frames[t])
            logit = engine.decode_current_logits(output_size)
            prob = torch.softmax(logit, dim=1)
            label = torch.argmax(prob, dim=1, keepdim=True).to(logit.dtype)
            rec = {'mask': label[0, 0].to(torch.uint8)}
            if 'logits4' in keep:
                rec['logits4'] = engine.pred_id_logits.clone()
            if 'logits' in keep:
                rec['logits'] = logit.clone()
            fb = label if teacher_masks is None else teacher_masks[t - 1].view(1, 1, *output_size).to(logit.dtype)
            fb = F.interpolate(fb, size=engine.input_size_2d, mode='nearest')
            engine.update_memory(fb)
            out.append(rec)
    return out


# --------------------------------------------------------------------------
# evaluator-side steps: dataloaders/video_transforms.py:594-715,
# networks/managers/evaluator.py:265-446
# --------------------------------------------------------------------------
def restrict_size(h, w, max_short_edge=None, max_long_edge=800, scale=1.0, align_corners=True, max_stride=16):
    """MultiRestrictSize's size arithmetic, video_transforms.py:612-653 (pinned on the reference class by
    tests/golden/transforms.json)."""
    sc = 1.
    if max_short_edge is not None:
        short_edge = w if h > w else h
        if short_edge > max_short_edge:
            sc *= float(max_short_edge) / short_edge
    new_h, new_w = sc * h, sc * w
    sc = 1.
    if max_long_edge is not None:
        long_edge = new_h if new_h > new_w else new_w
        if long_edge > max_long_edge:
            sc *= float(max_long_edge) / long_edge
    new_h, new_w = sc * new_h, sc * new_w
    new_h, new_w = int(new_h * scale), int(new_w * scale)
    if align_corners:
        if (new_h - 1) % max_stride != 0:
            new_h = int(np.around((new_h - 1) / max_stride) * max_stride + 1)
        if (new_w - 1) % max_stride != 0:
            new_w = int(np.around((new_w - 1) / max_stride) * max_stride + 1)
    else:
        if new_h % max_stride != 0:
            new_h = int(np.around(new_h / max_stride) * max_stride)
        if new_w % max_stride != 0:
            new_w = int(np.around(new_w / max_stride) * max_stride)
    return new_h, new_w


def cv2_cubic_resize(img, oh, ow):
    """cv2.resize(img, (ow, oh), interpolation=cv2.INTER_CUBIC) for a float32 H x W x C image, restated from OpenCV's
    published algorithm (modules/imgproc/src/resize.cpp: fx = (dx + 0.5) * scale - 0.5, sx = floor(fx), interpolateCubic
    with A = -0.75, taps clamped to the border, horizontal pass then vertical pass, float32 arithmetic).
    PARITY UNPINNED for this one function: OpenCV (the reference's dependency, unpinned in its requirements) is not
    installed here; tests cross-check it against torch's independent bicubic (same formula)."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape[:2]
    if (oh, ow) == (H, W):
        return img.copy()

    def taps(n_out, n_in):
        scale = float(n_in) / n_out
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        x = (f - i0).astype(np.float32)
        A = np.float32(-0.75)
        c = np.empty((n_out, 4), np.float32)
        c[:, 0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
        c[:, 1] = ((A + 2) * x - (A + 3)) * x * x + 1
        c[:, 2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
        c[:, 3] = np.float32(1) - c[:, 0] - c[:, 1] - c[:, 2]
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None], 0, n_in - 1)
        return idx, c

    ix, cx = taps(ow, W)
    iy, cy = taps(oh, H)
    rows = np.zeros((H, ow) + img.shape[2:], np.float32)
    for t in range(4):
        rows += img[:, ix[:, t]] * cx[:, t].reshape((1, ow) + (1,) * (img.ndim - 2))
    out = np.zeros((oh, ow) + img.shape[2:], np.float32)
    for t in range(4):
        out += rows[iy[:, t]] * cy[:, t].reshape((oh, 1) + (1,) * (img.ndim - 2))
    return out


def to_tensor_normalise(img):
    """MultiToTensor on one float32 H x W x 3 image, video_transforms.py:703-711 (numpy arithmetic kept literally)."""
    tmp = np.asarray(img, dtype=np.float32)
    tmp = tmp / 255.
    tmp -= (0.485, 0.456, 0.406)
    tmp /= (0.229, 0.224, 0.225)
    return torch.from_numpy(np.ascontiguousarray(tmp.transpose((2, 0, 1))))


def flip_tensor(t, dim):                                           # utils/image.py:108-112
    return t.index_select(dim, torch.arange(t.size(dim) - 1, -1, -1))


def sequence_eval(model, frames, labels, obj_nums, flip=False, multiscale=(1,), max_short_edge=None,
                  max_long_edge=800 * 1.3, long_term_mem_gap=None, short_term_mem_skip=1):
    """The per-sequence loop of Evaluator.evaluating, networks/managers/evaluator.py:265-446, on in-memory frames
    (float32 H x W x 3 arrays, 0..255): test-time augmentations from MultiRestrictSize, one engine per augmentation,
    probability fusion, new-object merge, label feedback.  Returns [(fused label [H,W], fused prob [nc,H,W])] for frames 1.. ."""
    H, W = frames[0].shape[:2]
    ac = model.spec['align_corners']
    augs = []
    for sc in multiscale:                                          # video_transforms.py:609-682
        nh, nw = restrict_size(H, W, max_short_edge, max_long_edge, sc, ac)
        augs.append((nh, nw, False))
        if flip:
            augs.append((nh, nw, True))
    engines = [OracleInferEngine(model, long_term_mem_gap) for _ in augs]
    out = []

    def prep(img, nh, nw, fl):
        r = cv2_cubic_resize(img, nh, nw)
        if fl:
            r = r[:, ::-1].copy()
        return to_tensor_normalise(r).unsqueeze(0)

    with torch.no_grad():
        for t, img in enumerate(frames):
            ins = [prep(img, *a) for a in augs]
            label = labels.get(t)
            if label is not None:
                label = torch.as_tensor(np.asarray(label)).float().view(1, 1, H, W)
            if t == 0:
                for e, x, (nh, nw, fl) in zip(engines, ins, augs):
                    lab = flip_tensor(label, 3) if fl else label
                    lab = F.interpolate(lab, size=x.shape[2:], mode='nearest')          # evaluator.py:309-312
                    e.add_reference_frame(x, lab, int(obj_nums[0]))
                    _set_skip(e, short_term_mem_skip)
                continue
            all_preds = []
            for e, x, (nh, nw, fl) in zip(engines, ins, augs):
                e.match_propogate_one_frame(x)
                lg = e.decode_current_logits((H, W))
                if fl:
                    lg = flip_tensor(lg, 3)                                              # :329-330
                all_preds.append(torch.softmax(lg, dim=1))                              # :332
            all_labels = [torch.argmax(p, dim=1, keepdim=True).float() for p in all_preds]   # :340-347
            prob = torch.mean(torch.cat(all_preds, 0), dim=0, keepdim=True)             # :349-352
            pred = torch.argmax(prob, dim=1, keepdim=True).float()
            if label is not None:                                                       # :362-392
                keep = (label == 0).float()
                all_labels = [l * keep + label * (1 - keep) for l in all_labels]
                pred = pred * keep + label * (1 - keep)
                new_nums = int(pred.max().item())
                for e, x, l, (nh, nw, fl) in zip(engines, ins, all_labels, augs):
                    cl = flip_tensor(l, 3) if fl else l
                    cl = F.interpolate(cl, size=x.shape[2:], mode='nearest')
                    e.add_reference_frame(x, cl, new_nums, frame_step=t)
                    _set_skip(e, short_term_mem_skip)
                    e.decode_current_logits((H, W))
                    e.update_memory(cl)
            else:                                                                       # :394-408
                for e, x, l, (nh, nw, fl) in zip(engines, ins, all_labels, augs):
                    cl = flip_tensor(l, 3) if fl else l
                    e.update_memory(F.interpolate(cl, size=x.shape[2:], mode='nearest'))
            out.append((pred[0, 0], prob[0]))
    return out


def _set_skip(infer_engine, skip):
    for e in infer_engine.aot_engines:
        e.short_term_mem_skip = skip
