"""Attention layers of AOT (reference networks/layers/attention.py).

The modules own the parameters under the reference's ``state_dict`` names and call the fused gfx950
kernels through ``aot_hip``.  All activations are token-major 2-D tensors ``[N, C]`` (row stride may
exceed C: kernels take an explicit leading dimension, so column slices of wider buffers are used in
place of the reference's permute/contiguous copies).
"""
import math

import torch
from torch import nn

import aot_hip
from networks.layers.basic import DWConv2d
from networks.layers.normalization import fold_dwconv_bn, linear_t

_SIMDS = 1024                 # 256 CUs x 4 SIMDs; one 32-query x 1-head (or 1-chunk) tile = one wave
_OCC_EFF = (0.8, 0.9, 0.97, 0.99, 1.0)   # measured MFMA-pipe fill at 1..5 resident waves per SIMD
_TOPK_QROWS = 512             # query rows per launch of the top-k (sparse) forms: bounds their score scratch


def attn_splits(nq, units, t, occ=4, c0=3.0, wg_waves=1):
    """How many key ranges to cut the bank into at GRID level (csrc/attention.hip).

    The kernels are bound by the MFMA pipe of each SIMD, so their run time is the MAKESPAN over SIMDs: (waves per SIMD,
    rounded up) x (key tiles per wave + a fixed per-wave cost c0, in key-tile units).  Measured on MI355X
    (N=1674, 8 heads): the time follows this model within a few % for every bank size; the minima are the totals
    that land just under a whole number of waves per SIMD.  ``nq`` = query rows over all lanes, ``units`` = heads
    (multi-head form) or value chunks (gated form); ``wg_waves`` = waves of a workgroup that split the workgroup's key
    range among themselves and merge in LDS (4 for the d=32 kernel: the total key split is 4 x the returned value; 1 for
    the gated kernels); ``occ`` = resident waves per SIMD (4 for the d=32 kernel; 1 for the wide gated kernel, where
    extra waves just queue)."""
    waves1 = ((nq + 31) // 32) * units * wg_waves
    tiles = (t + 31) // 32
    best, best_cost = 1, None
    for ns in range(1, max(1, min(16 // wg_waves, tiles // (4 * wg_waves))) + 1):
        k = -(-waves1 * ns // _SIMDS)
        eff = _OCC_EFF[min(k, occ, 5) - 1] if occ > 1 else 1.0
        cost = k * (-(-tiles // (ns * wg_waves)) + c0) / eff + (0.15 * ns * wg_waves if ns > 1 else 0.0)   # + merge pass
        if best_cost is None or cost < best_cost * 0.999:
            best, best_cost = ns, cost
    return best


def gated_splits(nq, t, slots=512):
    """Grid-level key split of the gated (DeAOT) attention kernel: one 4-wave workgroup per (32 queries, key range), two
    resident per CU (two waves per SIMD) = 512 slots.  Measured on MI355X (tools/dev/mb_gated.py, profiles/r03g_mb_gated.txt,
    N = 1674, bank of 1..14 frames): the fastest split at EVERY bank size is the largest one whose grid still fits one
    dispatch round (9 x 53 = 477 workgroups: 909 us = 99 TF at M = 14 against 1131 us for 14 splits) -- a second round
    costs more than longer key ranges.  At least four key tiles per range, at most 16 ranges (the partial-slab scratch).
    slots = 256 for the bf16x6 twin (one workgroup per CU; the same rule holds there: profiles/r03r_gated_x6_nvb.txt)."""
    qt = (nq + 31) // 32
    tiles = (t + 31) // 32
    return max(1, min(slots // max(qt, 1), tiles // 4, 16))


def gated_splits_x6(nq, lanes, t, max_splits=16):
    """Grid-level key split of the bf16x6 gated kernel (attn_x6_wide64_kernel, round 6): one 4-wave workgroup per (lane, 64 queries,
    key range), one resident per CU.  Same rule as gated_splits(): the largest split whose grid fits one dispatch round of 256
    workgroups, at least four key tiles per range (27 query pairs at 480p -> 9 ranges, 243 workgroups)."""
    qt = lanes * ((nq + 63) // 64)
    tiles = (t + 31) // 32
    ns = max(1, min(256 // max(qt, 1), tiles // 4, max_splits))
    return max(ns, -(-tiles // 10900))      # the kernel addresses a key range through 32-bit offsets: < 10 922 key tiles per range


def _planned_len(t, nq, kv_brows):
    """Bank length a launch is PLANNED for: beyond the first memorised frame the launch geometry (the grid-level key split)
    follows the bank's CAPACITY (kv_brows: rows between lanes = rows of the pre-allocated bank), not its length of the
    moment -- the kernels cut the true length into that many ranges.  One captured launch then serves every bank length
    (hipGraph replay with the length in a device int, T_dev), and the eager engine, planning the same way, stays
    bit-identical to the replayed one."""
    if t <= nq:
        return t
    return max(t, int(kv_brows))


class MultiheadAttention(nn.Module):
    """Long-term attention over the memory bank and (with use_linear) self-attention
    (reference attention.py:29-126).  ``core`` is everything between the input linears and
    ``projection``: softmax((Q/sqrt(d)) K^T) V per head, flash style on the fp32 matrix cores."""

    def __init__(self, d_model, num_head=8, dropout=0., use_linear=True, d_att=None, use_dis=False,
                 qk_chunks=1, max_mem_len_ratio=-1, top_k=-1):
        super().__init__()
        self.d_model = d_model
        self.dropout_p = dropout          # on the attention weights (attention.py:110); training-time only
        self.num_head = num_head
        self.hidden_dim = d_model // num_head
        self.d_att = self.hidden_dim if d_att is None else d_att
        self.T = self.d_att ** 0.5
        self.use_linear = use_linear
        if use_dis:
            raise NotImplementedError('use_dis is a default-off knob no reference config enables (attention.py:99-100)')
        self.max_mem_len_ratio = float(max_mem_len_ratio)     # eval-time Q rescale for long banks, attention.py:84-89
        self.top_k = int(top_k)                               # eval-time sparse softmax, attention.py:102-105
        if use_linear:
            self.linear_Q = nn.Linear(d_model, d_model)
            self.linear_K = nn.Linear(d_model, d_model)
            self.linear_V = nn.Linear(d_model, d_model)
        self.projection = nn.Linear(d_model, d_model)

    def core(self, q, k, v, out, t, ws, stream, t_dev=None, B=1, kv_brows=0, x6=None):
        """q [B*Nq, C], k/v token-major (lane b: rows b*kv_brows .. + t) -> out [B*Nq, C] (pre-projection).  x6 = (planes, rows per
        lane): the same bank pre-split for the bf16x6 kernel (aot_attn_x6_f32), used instead of k / v."""
        nq = q.shape[0] // B
        scale_div = self.T
        if self.max_mem_len_ratio > 0:
            ratio = float(t) / nq
            if ratio > self.max_mem_len_ratio:      # Q *= log(ratio)/log(max ratio), folded into the divisor
                scale_div = self.T / (math.log(ratio) / math.log(self.max_mem_len_ratio))
        if 0 < self.top_k < t:
            # score scratch: sized by the bank CAPACITY (kv_brows: it only changes when the bank re-allocates, i.e. doubles)
            # and by a block of query rows, not by the bank length of the moment -- a buffer per distinct length would grow
            # quadratically with the number of memorised frames (scratch buffers are persistent and keyed by shape)
            rows = max(t, int(kv_brows))
            qb = min(nq, _TOPK_QROWS)
            scores = ws.get('attn_scores', (self.num_head * qb * ((rows + 3) // 4 * 4),), q.device)
            for b in range(B):       # the sparse form is a long-video knob; lanes one at a time
                kb, vb = k[b * kv_brows:], v[b * kv_brows:]
                for r0 in range(b * nq, (b + 1) * nq, qb):
                    r1 = min(r0 + qb, (b + 1) * nq)
                    aot_hip.attention_topk(q[r0:r1], kb, vb, out[r0:r1], t, self.num_head, scale_div, self.top_k, scores,
                                           stream=stream)
            return out
        t_plan = _planned_len(t, nq, kv_brows)
        ns = attn_splits(nq * B, self.num_head, t_plan, wg_waves=4)
        part = None
        if ns > 1:      # one slab set sized for the largest grid split (4): no per-bank-size allocations
            part = ws.get('attn_part', (4 * B * nq * (self.d_model + 2 * self.num_head),), q.device)
        # (with a device-side length the host-side T only bounds the launch: the planned length)
        if x6 is not None and self.hidden_dim == 32:
            aot_hip.attention_x6(q, x6, out, t if t_dev is None else t_plan, self.num_head, scale_div, part=part, nsplit=ns,
                                 T_dev=t_dev, B=B, stream=stream)
            return out
        aot_hip.attention(q, k, v, out, t if t_dev is None else t_plan, self.num_head, scale_div, part=part, nsplit=ns,
                          T_dev=t_dev, B=B, kv_brows=kv_brows, stream=stream)
        return out


class MultiheadLocalAttention(nn.Module):
    """Short-term attention over a (2*max_dis+1)^2 window of the previous frame (reference
    MultiheadLocalAttentionV2, attention.py:248-428, with use_linear=False).  One fused kernel replaces the
    correlation sampler / unfold, the relative-position conv, the masked softmax, local2global and the
    dense N x N aggregation.  (The reference's fallback class V3 is broken as shipped -- attention.py:527-532 --
    and V2 is the semantics its CUDA extension implements; parameter names are identical.)"""

    def __init__(self, d_model, num_head, dropout=0., max_dis=7, dilation=1, use_linear=False, d_att=None):
        super().__init__()
        if use_linear or dilation != 1:
            raise NotImplementedError('AOT uses use_linear=False, dilation=1 (reference transformer.py:284-288)')
        self.window_size = 2 * max_dis + 1
        self.dropout_p = dropout
        self.max_dis = max_dis
        self.num_head = num_head
        self.hidden_dim = d_model // num_head
        self.d_att = self.hidden_dim if d_att is None else d_att
        self.T = self.d_att ** 0.5
        self.relative_emb_k = nn.Conv2d(self.d_att * num_head, num_head * self.window_size * self.window_size,
                                        kernel_size=1, groups=num_head)
        self.relative_emb_v = nn.Parameter(torch.zeros([num_head, d_model // num_head,
                                                        self.window_size * self.window_size]))
        self.projection = nn.Linear(d_model, d_model)
        self._packed = None

    def pack(self):
        if self._packed is None:
            self._packed = aot_hip.pack_local_tables(self.relative_emb_k.weight, self.relative_emb_k.bias,
                                                     self.relative_emb_v, self.num_head, self.max_dis)
        return self._packed

    def core(self, q, k, v, out, size_2d, stream, B=1, kv_brows=0):
        relk_w, relk_b, relv_t = self.pack()
        h, w = size_2d
        aot_hip.local_attention(q, k, v, relk_w, relk_b, relv_t, out, h, w, self.num_head, self.T,
                                max_dis=self.max_dis, B=B, kv_brows=kv_brows, stream=stream)
        return out


class GatedPropagation(nn.Module):
    """Global gated propagation of DeAOT (reference attention.py:589-717): single head, q = k of width d_att,
    value/gate of width expand_d_vu; out = projection(dw_conv5x5((softmax(qk^T) v) * u)).  With use_linear the
    inputs are projected first (self-attention form, :648-669)."""

    def __init__(self, d_qk, d_vu, num_head=8, dropout=0., use_linear=True, d_att=None, use_dis=False, qk_chunks=1,
                 max_mem_len_ratio=-1, top_k=-1, expand_ratio=2.):
        super().__init__()
        if num_head != 1:
            raise NotImplementedError('DeAOT configs use a single head (configs/models/default_deaot.py:14-15)')
        if use_dis:
            raise NotImplementedError('use_dis is a default-off knob no reference config enables (attention.py:697-698)')
        self.max_mem_len_ratio = float(max_mem_len_ratio)     # eval-time Q rescale for long banks, attention.py:674-679
        self.top_k = int(top_k)                               # eval-time sparse softmax, attention.py:689-693
        self.expand_d_vu = int(d_vu * expand_ratio)
        self.dropout_p = dropout
        self.d_vu, self.d_qk, self.num_head = d_vu, d_qk, num_head
        self.hidden_dim = self.expand_d_vu // num_head
        self.d_att = d_qk // num_head if d_att is None else d_att
        self.T = self.d_att ** 0.5
        self.use_linear = use_linear
        if use_linear:
            self.linear_QK = nn.Linear(d_qk, self.d_att * num_head)
            half = self.hidden_dim * num_head // 2
            self.linear_V1 = nn.Linear(d_vu // 2, half)
            self.linear_V2 = nn.Linear(d_vu // 2, half)
            self.linear_U1 = nn.Linear(d_vu // 2, half)
            self.linear_U2 = nn.Linear(d_vu // 2, half)
        self.dw_conv = DWConv2d(self.expand_d_vu)
        self.projection = nn.Linear(self.expand_d_vu, d_vu)
        self._p = None

    def pack(self):
        if self._p is None:
            p = {'dw': fold_dwconv_bn(self.dw_conv.conv)[0]}
            p['proj_w'], p['proj_b'] = linear_t(self.projection)
            if self.use_linear:
                for n in ('QK', 'V1', 'V2', 'U1', 'U2'):
                    p[n + '_w'], p[n + '_b'] = linear_t(getattr(self, 'linear_' + n))
            self._p = p
        return self._p

    def core(self, q, k, v, gate, out, t, ws, stream, t_dev=None, B=1, kv_brows=0, x6=None):
        """(softmax((q/T) k^T) v) * gate : q [B*Nq,128], k [.,128], v [.,E], gate/out [B*Nq,E]; lane b reads rows
        b*kv_brows .. + t of k/v.  x6 = (K planes, V planes, rows per lane): the same bank pre-split for the bf16x6 kernel
        (aot_gated_attn_x6_f32), used instead of k / v."""
        nq = q.shape[0] // B
        scale_div = self.T
        if self.max_mem_len_ratio > 0:       # Q *= log(ratio)/log(max ratio), folded into the divisor (attention.py:674-679)
            ratio = float(t) / nq
            if ratio > self.max_mem_len_ratio:
                scale_div = self.T / (math.log(ratio) / math.log(self.max_mem_len_ratio))
        if 0 < self.top_k < t:
            return self._core_topk(q, k, v, gate, out, t, scale_div, ws, stream, B, kv_brows)
        t_plan = _planned_len(t, nq, kv_brows)
        use_x6 = x6 is not None and out.shape[1] == 1024 and q.shape[1] == 128
        if use_x6:
            ns = gated_splits_x6(nq, B, t_plan)
        else:
            ns = gated_splits(nq * B, t_plan) if out.shape[1] == 1024 else attn_splits(nq * B, out.shape[1] // 256, t_plan, occ=1, c0=1.0)
        part = None
        if ns > 1:
            part = ws.get('gattn_part', (16 * B * nq * (out.shape[1] + 2 * (out.shape[1] // 256)),), q.device)
        if use_x6:
            aot_hip.gated_attention_x6(q, x6, gate, out, t if t_dev is None else t_plan, scale_div, part=part, nsplit=ns,
                                       T_dev=t_dev, B=B, stream=stream)
            return out
        aot_hip.gated_attention(q, k, v, gate, out, t if t_dev is None else t_plan, scale_div, part=part, nsplit=ns,
                                T_dev=t_dev, B=B, kv_brows=kv_brows, stream=stream)
        return out

    def _core_topk(self, q, k, v, gate, out, t, scale_div, ws, stream, B, kv_brows):
        """top_k > 0 (attention.py:689-693): scores materialised once, radix select of the k-th largest per query row, ordered
        gather of the selected [V | ID_V] rows, gate fused (csrc/attn_topk.hip); lanes one at a time."""
        nq = q.shape[0] // B
        cap = max(t, int(kv_brows))            # bank capacity, see MultiheadAttention.core
        qb = min(nq, _TOPK_QROWS)
        scores = ws.get('gattn_scores', (qb * ((cap + 3) // 4 * 4),), q.device)
        for b in range(B):
            for r0 in range(b * nq, (b + 1) * nq, qb):
                rows = slice(r0, min(r0 + qb, (b + 1) * nq))
                aot_hip.gated_attention_topk(q[rows], k[b * kv_brows:], v[b * kv_brows:],
                                             gate[rows] if gate is not None else None, out[rows], t, scale_div, self.top_k,
                                             scores, stream=stream)
        return out

    def tail(self, raw, out, size_2d, ws, stream, res=None, B=1):
        """projection(dw_conv(raw)) (+ res): attention.py:709-710."""
        p = self.pack()
        h, w = size_2d
        E = raw.shape[1]
        tmp = ws.get('gp_dw', (raw.shape[0], E), raw.device)
        aot_hip.dwconv2d(raw, p['dw'], None, tmp, h, w, E, h, w, 5, 1, 2, 1, B=B, stream=stream)
        aot_hip.linear(tmp, p['proj_w'], p['proj_b'], out, res=res, stream=stream)
        return out


class LocalGatedPropagation(nn.Module):
    """Short-term gated propagation of DeAOT over the 15x15 window (reference attention.py:720-914, use_linear=False)."""

    def __init__(self, d_qk, d_vu, num_head, dropout=0., max_dis=7, dilation=1, use_linear=False, d_att=None,
                 use_dis=False, expand_ratio=2.):
        super().__init__()
        if num_head != 1 or use_linear or dilation != 1 or use_dis:
            raise NotImplementedError('DeAOT uses one head, use_linear=False, dilation=1 (transformer.py:550-559)')
        self.expand_d_vu = int(d_vu * expand_ratio)
        self.dropout_p = dropout
        self.window_size = 2 * max_dis + 1
        self.max_dis, self.num_head = max_dis, num_head
        self.hidden_dim = self.expand_d_vu // num_head
        self.d_att = d_qk // num_head if d_att is None else d_att
        self.T = self.d_att ** 0.5
        self.relative_emb_k = nn.Conv2d(self.d_att * num_head, num_head * self.window_size * self.window_size,
                                        kernel_size=1, groups=num_head)
        self.dw_conv = DWConv2d(self.expand_d_vu)
        self.projection = nn.Linear(self.expand_d_vu, d_vu)
        self._p = None

    def pack(self):
        if self._p is None:
            ws_ = self.window_size
            d = self.d_att
            wk = (self.relative_emb_k.weight.detach().double().reshape(ws_, ws_, d) * (float(d) ** 0.5)).float()
            wk = torch.nn.functional.pad(wk.permute(0, 2, 1), (0, 16 - ws_)).contiguous()            # [dy, c, 16]
            bk = torch.nn.functional.pad(self.relative_emb_k.bias.detach().float().reshape(ws_, ws_), (0, 16 - ws_)).contiguous()
            p = {'relk_t': wk, 'relk_b': bk, 'dw': fold_dwconv_bn(self.dw_conv.conv)[0]}
            p['proj_w'], p['proj_b'] = linear_t(self.projection)
            self._p = p
        return self._p

    def core(self, q, k, v, gate, out, size_2d, ws, stream, B=1, kv_brows=0):
        p = self.pack()
        h, w = size_2d
        prob = ws.get('lgp_prob', (B * self.window_size * self.window_size * h * w,), q.device)
        aot_hip.local_gated(q, k, v, gate, p['relk_t'], p['relk_b'], prob, out, h, w, self.T, max_dis=self.max_dis, B=B,
                            kv_brows=kv_brows, stream=stream)
        return out

    tail = GatedPropagation.tail
