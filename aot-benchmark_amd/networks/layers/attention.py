"""Attention layers of AOT (reference networks/layers/attention.py).

The modules own the parameters under the reference's ``state_dict`` names and call the fused gfx950
kernels through ``aot_hip``.  All activations are token-major 2-D tensors ``[N, C]`` (row stride may
exceed C: kernels take an explicit leading dimension, so column slices of wider buffers are used in
place of the reference's permute/contiguous copies).
"""
import torch
from torch import nn

import aot_hip
from networks.layers.normalization import linear_t

_TARGET_WAVES = 2048          # 256 CUs x 4 SIMDs x 2 waves: one 32-query x 1-head tile = one wave


def attn_splits(nq, heads, t):
    """How many key ranges to cut the bank into (csrc/attention.hip).  Measured on MI355X (scratch/mb_attn.py):
    the kernel wants several waves per SIMD but >= ~12 key tiles per split; tiny problems split just enough to
    put a wave on every SIMD."""
    waves = ((nq + 31) // 32) * heads
    tiles = (t + 31) // 32
    fill = min((_TARGET_WAVES + waves - 1) // waves, tiles // 4)
    return max(1, min(16, max(tiles // 12, fill)))


class MultiheadAttention(nn.Module):
    """Long-term attention over the memory bank and (with use_linear) self-attention
    (reference attention.py:29-126).  ``core`` is everything between the input linears and
    ``projection``: softmax((Q/sqrt(d)) K^T) V per head, flash style on the fp32 matrix cores."""

    def __init__(self, d_model, num_head=8, dropout=0., use_linear=True, d_att=None, use_dis=False,
                 qk_chunks=1, max_mem_len_ratio=-1, top_k=-1):
        super().__init__()
        self.d_model = d_model
        self.num_head = num_head
        self.hidden_dim = d_model // num_head
        self.d_att = self.hidden_dim if d_att is None else d_att
        self.T = self.d_att ** 0.5
        self.use_linear = use_linear
        if use_dis or top_k > 0 or max_mem_len_ratio > 0:
            raise NotImplementedError('use_dis / top_k / max_mem_len_ratio are default-off eval knobs '
                                      '(reference attention.py:37-47); not built yet')
        if use_linear:
            self.linear_Q = nn.Linear(d_model, d_model)
            self.linear_K = nn.Linear(d_model, d_model)
            self.linear_V = nn.Linear(d_model, d_model)
        self.projection = nn.Linear(d_model, d_model)

    def core(self, q, k, v, out, t, ws, stream, t_dev=None):
        """q [Nq, C], k/v [>=t, C] token-major -> out [Nq, C] (pre-projection)."""
        nq = q.shape[0]
        ns = attn_splits(nq, self.num_head, t)
        part = None
        if ns > 1:
            part = ws.get('attn_part', (ns * nq * (self.d_model + 2 * self.num_head),), q.device)
        aot_hip.attention(q, k, v, out, t, self.num_head, self.T, part=part, nsplit=ns, T_dev=t_dev, stream=stream)
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

    def core(self, q, k, v, out, size_2d, stream):
        relk_w, relk_b, relv_t = self.pack()
        h, w = size_2d
        aot_hip.local_attention(q, k, v, relk_w, relk_b, relv_t, out, h, w, self.num_head, self.T,
                                max_dis=self.max_dis, stream=stream)
        return out
