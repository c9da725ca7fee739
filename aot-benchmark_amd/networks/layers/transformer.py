"""LSTT stack of AOT and GPM stack of DeAOT (reference networks/layers/transformer.py:33-140, 143-255, 258-372, 501-670).

Every activation is token-major [B*N, C]: B LANES stacked along the rows -- the object groups of one frame (each group of
<= 10 objects is an independent pass over the SAME image features, reference AOTInferEngine, aot_engine.py:584-616) run
as one batch through every kernel instead of one engine after the other.  Memories are (tensor, rows-between-lanes)
pairs, so a lane's keys can live in its bank slot or in a stacked scratch buffer.

Per-layer launch sequence of the AOT block (N = h*w):
  LN1 (+pos, two outputs) -> [Q|K] GEMM, V GEMM -> flash self-attention -> projection GEMM (+residual)
  LN2 -> linear_Q GEMM -> flash long-term attention over the bank  +  fused windowed short-term attention
  (both write halves of one [B*N, 512] buffer) -> one K=512 GEMM = proj_lt + proj_st (+residual)
  LN3 -> linear1 GEMM -> GroupNorm statistics -> [GN-apply + GELU + 5x5 depthwise conv] -> linear2 GEMM (+residual)
"""
import os

import torch
from torch import nn

import aot_hip
from aot_hip import attach_wt
from networks.layers.attention import (GatedPropagation, LocalGatedPropagation, MultiheadAttention,
                                       MultiheadLocalAttention)
from networks.layers.basic import GNActDWConv2d, GroupNorm1D
from networks.layers.normalization import fold_dwconv_bn, linear_t


def _ln_params(ln):
    return ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous()


def _droppath_rate(droppath, idx, num_layers, scaling):
    """transformer.py:71-78: with droppath_scaling the rate grows linearly over the layers."""
    if not scaling:
        return droppath
    return 0 if num_layers == 1 else droppath * idx / (num_layers - 1)


class LongShortTermTransformerBlock(nn.Module):
    def __init__(self, d_model, self_nhead, att_nhead, dim_feedforward=1024, droppath=0.1, lt_dropout=0.,
                 st_dropout=0., droppath_lst=False, activation='gelu', local_dilation=1):
        super().__init__()
        self.d_model = d_model
        self.dim_ff = dim_feedforward
        # training-time regularisers (transformer.py:288-289,302; applied by models/train_forward.py only)
        self.droppath_p, self.droppath_lst, self.lst_dropout_p = droppath, droppath_lst, max(lt_dropout, st_dropout)
        # parameter names/shapes = reference transformer.py:273-300
        self.norm1 = nn.LayerNorm(d_model)
        self.linear_Q = nn.Linear(d_model, d_model)
        self.linear_V = nn.Linear(d_model, d_model)
        self.long_term_attn = MultiheadAttention(d_model, att_nhead, use_linear=False, dropout=lt_dropout)
        self.short_term_attn = MultiheadLocalAttention(d_model, att_nhead, dilation=local_dilation,
                                                       use_linear=False, dropout=st_dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.self_attn = MultiheadAttention(d_model, self_nhead)
        self.norm3 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.activation = GNActDWConv2d(dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self._p = None

    # ---- one-time weight packing ------------------------------------------------------------
    def pack(self):
        if self._p is not None:
            return self._p
        p = {}
        sa = self.self_attn
        wq, bq = linear_t(sa.linear_Q)
        wk, bk = linear_t(sa.linear_K)
        p['sa_qk_w'] = attach_wt(torch.cat([wq, wk], 1).contiguous())   # [256, 512]: one GEMM for Q and K of (x1 + pos)
        p['sa_qk_b'] = torch.cat([bq, bk]).contiguous()
        p['sa_v_w'], p['sa_v_b'] = linear_t(sa.linear_V)
        # Q, K and V of the self-attention as ONE product on the normed input (round 5): (x1 + pos) Wq = x1 Wq + pos Wq, and pos is
        # fixed for a clip, so pos [Wq | Wk] + bias is computed once per clip (prepare_pos) and rides as the shared residual map
        p['sa_qkv_w'] = attach_wt(torch.cat([wq, wk, p['sa_v_w'][:, :wq.shape[1]]], 1).contiguous())   # [256, 768]
        p['sa_o_w'], p['sa_o_b'] = linear_t(sa.projection)
        p['q_w'], p['q_b'] = linear_t(self.linear_Q)
        p['v_w'], p['v_b'] = linear_t(self.linear_V)
        wl, bl = linear_t(self.long_term_attn.projection)
        ws_, bs_ = linear_t(self.short_term_attn.projection)
        p['lst_w'] = attach_wt(torch.cat([wl, ws_], 0).contiguous())    # [512, 256]: lt and st projections in one GEMM
        p['lst_b'] = (bl + bs_).contiguous()
        p['w1'], p['b1'] = linear_t(self.linear1)
        p['w2'], p['b2'] = linear_t(self.linear2)
        p['dw'], _ = fold_dwconv_bn(self.activation.conv)
        for n in ('norm1', 'norm2', 'norm3'):
            p[n] = _ln_params(getattr(self, n))
        # LayerNorm as the prologue of the GEMM behind it (round 6, aot_layernorm_linear_bf16x6_f32): the affine half folded into the
        # weights -- norm1 -> merged Q|K|V (its residual map carries the biases), norm3 -> linear1
        p['ln1_qkv'] = aot_hip.fold_layernorm(p['sa_qkv_w'], None, *p['norm1'])
        p['ln3_w1'] = aot_hip.fold_layernorm(p['w1'], p['b1'], *p['norm3'])
        p['gn'] = (self.activation.gn.weight.detach().float().contiguous(),
                   self.activation.gn.bias.detach().float().contiguous())
        self.short_term_attn.pack()
        self._p = p
        return p

    def prepare_pos(self, pos, stream=None, out=None):
        """[pos Wq + bq | pos Wk + bk | bv] for a clip's position embedding pos [N, C]: the residual map of the merged Q|K|V product.
        Launched from the host once per clip (never inside a graph capture); out: the caller's [N, 3C] buffer (stable address)."""
        p = self.pack()
        N, C = pos.shape
        if out is None:
            out = torch.empty(N, 3 * C, dtype=torch.float32, device=pos.device)
        aot_hip.linear(pos, p['sa_qk_w'], p['sa_qk_b'], out[:, :2 * C], stream=stream)
        out[:, 2 * C:] = p['sa_v_b']
        return out

    # ---- reference transformer.py:312-362 -----------------------------------------------------
    def run(self, x, long_mem, short_mem, id_emb, pos, size_2d, ws, stream, B=1, dst=None, keep=None, x6=None, pos_qkv=None,
            out_norm=None):
        """x [B*N, C(ld)] token-major (B lanes).  out_norm = (gamma, beta, dst, eps): the stack's norm of this layer's output, written to dst
        by this call (from linear2's reduce launch where that layer runs split-K).  x6 = the bank's pre-split copy (planes, rows per lane) for the bf16x6 attention
        kernel, when the engine keeps one.  long_mem = (K, V, T, kv_brows[, T_dev]): lane b's bank = rows b*kv_brows .. + T
        (T_dev: device int holding T, for launches replayed from a graph while the bank grows);
        short_mem = (K, V, kv_brows).  dst = (k_out, v_out) [B*N, C] buffers for this frame's K (= linear_Q output) and, on
        a reference frame, the id-fused V (e.g. the lane's bank slot); allocated when None.  keep = the caller's arena
        (a Workspace) for the tensors that outlive this call; None: fresh tensors.
        Returns (out [B*N,C], curr_K, curr_V (normed input), fused_V or None)."""
        p = self.pack()
        M, C = x.shape
        N = M // B
        dev = x.device
        h, w = size_2d
        tag = id(self)

        def new(name, *s):            # tensors that outlive this call
            if keep is not None:
                return keep.get('%s_%d' % (name, tag), s, dev)
            return torch.empty(s, dtype=torch.float32, device=dev)

        # self-attention
        x1 = ws.get('x1', (M, C), dev)
        if pos_qkv is not None:       # one product for Q, K and V (pos_qkv = prepare_pos(pos): see pack())
            qkv = ws.get('sa_qkv', (M, 3 * C), dev)
            if aot_hip.x6_ln_fusable(M, C, 3 * C):      # ... with norm1 as its prologue: x1 is never written
                aot_hip.layernorm_linear_x6(x, *p['ln1_qkv'], qkv, eps=self.norm1.eps, res=pos_qkv, res_rows=N, stream=stream)
            else:
                aot_hip.layernorm(x, *p['norm1'], x1, eps=self.norm1.eps, stream=stream)
                aot_hip.linear(x1, p['sa_qkv_w'], None, qkv, res=pos_qkv, res_rows=N, stream=stream)
            qk, sv = qkv[:, :2 * C], qkv[:, 2 * C:]
        else:
            x1p = ws.get('x1p', (M, C), dev)
            aot_hip.layernorm(x, *p['norm1'], x1, add=pos, out2=x1p, add_rows=N, stream=stream)
            qk = ws.get('sa_qk', (M, 2 * C), dev)
            aot_hip.linear(x1p, p['sa_qk_w'], p['sa_qk_b'], qk, stream=stream)
            sv = ws.get('sa_v', (M, C), dev)
            aot_hip.linear(x1, p['sa_v_w'], p['sa_v_b'], sv, stream=stream)
        so = ws.get('sa_o', (M, C), dev)
        sx6 = None
        if x6 is not None and self.self_attn.hidden_dim == 32:
            # bf16x6 family: the frame's own K / V split into a scratch bank of one frame per lane (aot_attn_pack_x6_f32), then the
            # same kernel as the long-term attention
            cap = (N + 31) // 32 * 32
            sx6 = (ws.get_zeroed('sa_x6', (B * cap * C * 6,), dev, torch.int16), cap)
            aot_hip.attention_pack_x6(qk[:, C:], sv, sx6, N, B=B, src_brows=N, stream=stream)
        self.self_attn.core(qk[:, :C], qk[:, C:], sv, so, N, ws, stream, B=B, kv_brows=N, x6=sx6)
        xa = ws.get('xa', (M, C), dev)
        aot_hip.linear(so, p['sa_o_w'], p['sa_o_b'], xa, res=x, stream=stream)

        # long + short term attention
        x2 = new('x2', M, C)                                      # curr_V (normed input, transformer.py:333)
        aot_hip.layernorm(xa, *p['norm2'], x2, stream=stream)
        qc = dst[0] if dst is not None else new('qc', M, C)       # curr_Q == curr_K (:331-332)
        aot_hip.linear(x2, p['q_w'], p['q_b'], qc, stream=stream)
        fused_v = None
        if id_emb is not None:                                    # reference frame: memorise itself (:337-341)
            fused_v = self.fuse_kv_2d(x2, id_emb, ws, stream, out=dst[1] if dst is not None else None)
            gk, gv, t, g_brows, t_dev = qc, fused_v, N, N, ()
            lk, lv, l_brows = qc, fused_v, N
        else:
            gk, gv, t, g_brows, *t_dev = long_mem
            lk, lv, l_brows = short_mem
        cat = ws.get('lst_cat', (M, 2 * C), dev)
        self.long_term_attn.core(qc, gk, gv, cat[:, :C], t, ws, stream, t_dev=t_dev[0] if t_dev else None, B=B,
                                 kv_brows=g_brows, x6=x6 if id_emb is None else None)
        self.short_term_attn.core(qc, lk, lv, cat[:, C:], size_2d, stream, B=B, kv_brows=l_brows)
        xb = ws.get('xb', (M, C), dev)
        aot_hip.linear(cat, p['lst_w'], p['lst_b'], xb, res=xa, stream=stream)

        # feed-forward: linear1 -> GN(32) statistics -> [GN-apply + GELU + dw5x5] -> linear2
        F1 = self.dim_ff
        f = ws.get('ffn_a', (M, F1), dev)
        g = ws.get('ffn_b', (M, F1), dev)
        gn_fuse = F1 == 32 * 32 and aot_hip.x6_gn_fusable(M, C, F1, B) and not os.environ.get('AOT_NO_GN_FUSE')
        ln_fuse = aot_hip.x6_ln_fusable(M, C, F1)      # norm3 as linear1's prologue (round 6): x3 is never written
        if not ln_fuse:
            x3 = ws.get('x3', (M, C), dev)
            aot_hip.layernorm(xb, *p['norm3'], x3, eps=self.norm3.eps, stream=stream)
        if gn_fuse:
            # bf16x6, one lane: linear1's tile end writes the GroupNorm partial sums, the fused GN + GELU + dw5x5 kernel adds them up
            # in its prologue -- no statistics launch, no extra pass over the [M, 1024] map
            part = ws.get('ffn_gnpart', (2 * ((M + 63) // 64) * 32 * 2,), dev)
            if ln_fuse:
                P = aot_hip.layernorm_linear_x6(xb, *p['ln3_w1'], f, eps=self.norm3.eps, gn_part=part, stream=stream)
            else:
                P = aot_hip.linear_gn_x6(x3, p['w1'], p['b1'], f, part, stream=stream)
            aot_hip.gn_act_dwconv5_part(f, *p['gn'], p['dw'], g, 32, part, P, h, w, act=aot_hip.ACT_GELU, eps=self.activation.gn.eps,
                                        stream=stream)
            out = ws.get('layer_out_%d' % id(self), (M, C), dev)
            self._linear2(g, out, xb, out_norm, stream)
            return out, qc, x2, fused_v
        if ln_fuse:
            aot_hip.layernorm_linear_x6(xb, *p['ln3_w1'], f, eps=self.norm3.eps, stream=stream)
        else:
            aot_hip.linear(x3, p['w1'], p['b1'], f, stream=stream)
        if F1 == 32 * 32:
            aot_hip.gn_act_dwconv5(f, *p['gn'], p['dw'], g, 32, aot_hip.gn_buffers(ws, dev, B, 32, 8), h, w,
                                   act=aot_hip.ACT_GELU, nsplit=8, B=B, stream=stream)
        else:       # generic widths: GN-apply + GELU, then the depthwise conv
            aot_hip.groupnorm(f, *p['gn'], g, 32, aot_hip.gn_buffers(ws, dev, B, 32, 8), act=aot_hip.ACT_GELU, nsplit=8,
                              B=B, stream=stream)
            f2 = ws.get('ffn_c', (M, F1), dev)
            aot_hip.dwconv2d(g, p['dw'], None, f2, h, w, F1, h, w, 5, 1, 2, 1, B=B, stream=stream)
            g = f2
        out = ws.get('layer_out_%d' % id(self), (M, C), dev)
        self._linear2(g, out, xb, out_norm, stream)
        return out, qc, x2, fused_v

    def _linear2(self, g, out, xb, out_norm, stream):
        """linear2 + residual (transformer.py:359-362), and the stack's norm of the result when the caller hands it over."""
        p = self._p
        if out_norm is None:
            return aot_hip.linear(g, p['w2'], p['b2'], out, res=xb, stream=stream)
        gamma, beta, d, eps = out_norm
        return aot_hip.linear_ln_out(g, p['w2'], p['b2'], out, gamma, beta, d, eps=eps, res=xb, stream=stream)

    def fuse_kv_2d(self, v, id_emb, ws, stream, out=None, summed=None):
        """linear_V(V + id_emb) (transformer.py:364-367).  `summed` = V + id_emb already formed (fused id-bank launch)."""
        p = self.pack()
        M, C = v.shape
        if summed is None:
            summed = ws.get('fuse_tmp', (M, C), v.device)
            aot_hip.add(v, id_emb, summed, stream=stream)
        if out is None:
            out = torch.empty(M, C, dtype=torch.float32, device=v.device)
        aot_hip.linear(summed, p['v_w'], p['v_b'], out, stream=stream)
        return out

    def fuse_key_value_id(self, key, value, id_emb):
        """Reference API (transformer.py:364-367): K unchanged, V <- linear_V(V + id_emb); [N,1,C] tensors."""
        n, b, c = value.shape
        v2 = self.fuse_kv_2d(value.reshape(n * b, c).contiguous(), id_emb.reshape(n * b, c).contiguous(), self._ws(),
                             aot_hip.stream_ptr())
        return key, v2.view(n, b, c)

    def _ws(self):
        if not hasattr(self, '_own_ws'):
            from networks.layers.workspace import Workspace
            self._own_ws = Workspace()
        return self._own_ws


class LongShortTermTransformer(nn.Module):
    def __init__(self, num_layers=2, d_model=256, self_nhead=8, att_nhead=8, dim_feedforward=1024, emb_dropout=0.,
                 droppath=0.1, lt_dropout=0., st_dropout=0., droppath_lst=False, droppath_scaling=False,
                 activation='gelu', return_intermediate=False, intermediate_norm=True, final_norm=True,
                 block_version='v1'):
        super().__init__()
        if block_version != 'v1':
            raise NotImplementedError('only block v1 is used by the reference configs (transformer.py:61-68)')
        self.intermediate_norm = intermediate_norm
        self.final_norm = final_norm
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.mask_token = nn.Parameter(torch.randn([1, 1, d_model]))   # unused at inference (transformer.py:59,105)
        self.emb_dropout_p = emb_dropout
        self.layers = nn.ModuleList([
            LongShortTermTransformerBlock(d_model, self_nhead, att_nhead, dim_feedforward,
                                          _droppath_rate(droppath, i, num_layers, droppath_scaling), lt_dropout,
                                          st_dropout, droppath_lst, activation) for i in range(num_layers)])
        num_norms = (num_layers - 1 if intermediate_norm else 0) + (1 if final_norm else 0)
        self.decoder_norms = nn.ModuleList([nn.LayerNorm(d_model) for _ in range(num_norms)]) if num_norms > 0 else None

    def run(self, x0, long_mems, short_mems, id_emb, pos, size_2d, ws, stream, B=1, dst=None, keep=None, x6=None):
        """Runs the stack for B lanes on the projected encoder feature x0 [N, C] (shared by the lanes).  Returns
        (dec_in, mems): dec_in is the decoder's concatenated input [B*N, (L+1)*C] (models/aot.py:86-92) -- block 0 = x0,
        blocks 1.. = the layer outputs after their decoder norm (transformer.py:124-135), written in place so the concat is
        never a copy; mems[i] = (curr_K, curr_V, fused_V | None) of layer i."""
        N, C = x0.shape
        L = self.num_layers
        if keep is not None:
            out_cat = keep.get('lstt_out_cat', (B * N, (L + 1) * C), x0.device)
        else:
            out_cat = torch.empty(B * N, (L + 1) * C, dtype=torch.float32, device=x0.device)
        aot_hip.copy_rows(x0, out_cat, N, B=B, src_brows=0, dst_brows=N, stream=stream)    # the same image feature for every lane
        x = out_cat[:, :C]
        mems = []
        pq = getattr(pos, '_aot_pos_qkv', None) if pos is not None else None      # prepare_pos(): the merged Q|K|V product's residual maps
        for i, layer in enumerate(self.layers):
            is_last = i == L - 1
            norm = None
            if self.decoder_norms is not None:
                if is_last and self.final_norm:
                    norm = self.decoder_norms[-1]
                elif not is_last and self.return_intermediate and self.intermediate_norm:
                    norm = self.decoder_norms[i]
            d = out_cat[:, (i + 1) * C:(i + 2) * C]
            # the stack's norm of the layer output rides on the layer's last launch (round 6)
            x, ck, cv, fv = layer.run(x, long_mems[i] if long_mems is not None else None,
                                      short_mems[i] if short_mems is not None else None,
                                      id_emb, pos, size_2d, ws, stream, B=B, dst=dst[i] if dst is not None else None,
                                      keep=keep, x6=x6[i] if x6 is not None else None, pos_qkv=pq[i] if pq is not None else None,
                                      out_norm=(norm.weight, norm.bias, d, norm.eps) if norm is not None else None)
            mems.append((ck, cv, fv))
            if norm is None:
                d.copy_(x)
        return out_cat, mems

    def prepare_pos(self, pos, stream=None, outs=None):
        """Once per clip, from the host: every layer's [pos Wq + bq | pos Wk + bk | bv], kept on the position tensor itself (the
        engine passes the same tensor every frame; the maps go when it goes).  outs: per-layer [N, 3C] buffers of the caller."""
        pos._aot_pos_qkv = [layer.prepare_pos(pos, stream, outs[i] if outs is not None else None) for i, layer in enumerate(self.layers)]
        return pos

    def update_values(self, mems, id_sums, ws, stream, dst=None):
        """Memory update of every layer once the frame's mask is known (aot_engine.py:307-338): V <- linear_V(V + id_emb)
        with the sums V + id_emb already formed by the fused id-bank launch.  dst[i] = where layer i's fused V goes."""
        L = len(mems)
        if dst is not None and 1 < L <= 4 and all(s is not None for s in id_sums):
            # the layers' linear_V launches are independent and of one shape: one grouped launch (round 6; each alone fills 108 of 256 CUs)
            ps = [layer.pack() for layer in self.layers]
            return aot_hip.linear_group(list(id_sums), [p['v_w'] for p in ps], [p['v_b'] for p in ps], list(dst), stream=stream)
        return [self.layers[i].fuse_kv_2d(m[1], None, ws, stream, out=dst[i] if dst is not None else None, summed=id_sums[i])
                for i, m in enumerate(mems)]


class GatedPropagationModule(nn.Module):
    """DeAOT block (reference transformer.py:501-670).  The two branches live in ONE token-major state
    X = [tgt | tgt_id] of width 2*d_model, so both residual updates are GEMM epilogues on a 512-wide buffer."""

    def __init__(self, d_model, self_nhead, att_nhead, dim_feedforward=1024, droppath=0.1, lt_dropout=0.,
                 st_dropout=0., droppath_lst=False, activation='gelu', local_dilation=1, max_local_dis=7,
                 layer_idx=0, expand_ratio=2.):
        super().__init__()
        expand_d_model = int(d_model * expand_ratio)
        self.expand_d_model, self.d_model, self.att_nhead = expand_d_model, d_model, att_nhead
        d_att = d_model // 2 if att_nhead == 1 else d_model // att_nhead
        self.d_att, self.layer_idx = d_att, layer_idx
        self.droppath_p, self.droppath_lst, self.lst_dropout_p = droppath, droppath_lst, max(lt_dropout, st_dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear_QV = nn.Linear(d_model, d_att * att_nhead + expand_d_model)
        self.linear_U = nn.Linear(d_model, expand_d_model)
        if layer_idx == 0:
            self.linear_ID_V = nn.Linear(d_model, expand_d_model)
        else:
            self.id_norm1 = nn.LayerNorm(d_model)
            self.linear_ID_V = nn.Linear(d_model * 2, expand_d_model)
            self.linear_ID_U = nn.Linear(d_model, expand_d_model)
        self.long_term_attn = GatedPropagation(d_qk=d_model, d_vu=d_model * 2, num_head=att_nhead, use_linear=False,
                                               dropout=lt_dropout, d_att=d_att, top_k=-1, expand_ratio=expand_ratio)
        self.short_term_attn = LocalGatedPropagation(d_qk=d_model, d_vu=d_model * 2, num_head=att_nhead,
                                                     dilation=local_dilation, use_linear=False, dropout=st_dropout,
                                                     d_att=d_att, max_dis=max_local_dis, expand_ratio=expand_ratio)
        self.norm2 = nn.LayerNorm(d_model)
        self.id_norm2 = nn.LayerNorm(d_model)
        self.self_attn = GatedPropagation(d_model * 2, d_model * 2, self_nhead, d_att=d_att)
        self._p = None

    def pack(self):
        if self._p is None:
            p = {}
            wqv, bqv = linear_t(self.linear_QV)                       # [256, 128 + 512]
            da = self.d_att * self.att_nhead
            p['q_w'], p['q_b'] = wqv[:, :da], bqv[:da].contiguous()   # column views of one packed matrix (ldb = 640)
            p['v_w'], p['v_b'] = wqv[:, da:], bqv[da:].contiguous()
            if getattr(wqv, '_aot_wt', None) is not None:             # ... and the matching row blocks of its k-contiguous twin
                p['q_w']._aot_wt, p['v_w']._aot_wt = wqv._aot_wt[:da], wqv._aot_wt[da:]
            aot_hip._register_weight(p['q_w'])
            aot_hip._register_weight(p['v_w'])
            p['u_w'], p['u_b'] = linear_t(self.linear_U)
            widv, p['idv_b'] = linear_t(self.linear_ID_V)
            if self.layer_idx == 0:
                p['idv_w_id'] = widv
            else:
                D = self.d_model
                p['idv_w_prev'], p['idv_w_id'] = attach_wt(widv[:D].contiguous()), attach_wt(widv[D:].contiguous())
                p['idu_w'], p['idu_b'] = linear_t(self.linear_ID_U)
                p['id_norm1'] = _ln_params(self.id_norm1)
            for n in ('norm1', 'norm2', 'id_norm2'):
                p[n] = _ln_params(getattr(self, n))
            for m in (self.long_term_attn, self.short_term_attn, self.self_attn):
                m.pack()
            self._p = p
        return self._p

    def fuse_id_into(self, vcat, prev_idv, id_emb, ws, stream):
        """ID_V = silu(linear_ID_V([prev_ID_V,] id_emb)) written into vcat[:, E:] (transformer.py:659-665)."""
        p = self.pack()
        E = self.expand_d_model
        dst = vcat[:, E:]
        if self.layer_idx == 0:
            aot_hip.linear(id_emb, p['idv_w_id'], p['idv_b'], dst, act=aot_hip.ACT_SILU, stream=stream)
        else:
            tmp = ws.get('gpm_idv_tmp', (vcat.shape[0], E), vcat.device)
            aot_hip.linear(prev_idv, p['idv_w_prev'], p['idv_b'], tmp, stream=stream)
            aot_hip.linear(id_emb, p['idv_w_id'], None, dst, res=tmp, act=aot_hip.ACT_SILU, stream=stream)
        return vcat

    def run(self, X, long_mem, short_mem, id_emb, pos, size_2d, ws, stream, B=1, dst=None, keep=None, x6=None):
        """X [B*N, 2D] = [tgt | tgt_id] (tgt_id = 0 into layer 0), B lanes.  long_mem = (K, Vcat, T, kv_brows), short_mem =
        (K, Vcat, kv_brows); dst = (k_out [B*N, d_att], vcat_out [B*N, 2E]) for this frame's K and [V | ID_V].
        Returns (X_out, curr_K, curr_Vcat, curr_ID_V_input)."""
        p = self.pack()
        M = X.shape[0]
        N = M // B
        D, E, da = self.d_model, self.expand_d_model, self.d_att
        dev = X.device

        def new(name, *s):            # tensors that outlive this call (keep = the caller's arena)
            if keep is not None:
                return keep.get('gpm_%s_%d' % (name, self.layer_idx), s, dev)
            return torch.empty(s, dtype=torch.float32, device=dev)
        x1 = ws.get('gpm_x1', (M, D), dev)
        aot_hip.layernorm(X[:, :D], *p['norm1'], x1, stream=stream)
        qc = dst[0] if dst is not None else new('qc', M, da)                    # curr_Q == curr_K (:597)
        aot_hip.linear(x1, p['q_w'], p['q_b'], qc, stream=stream)
        vcat = dst[1] if dst is not None else new('vcat', M, 2 * E)               # [curr_V | ID_V]
        aot_hip.linear(x1, p['v_w'], p['v_b'], vcat[:, :E], act=aot_hip.ACT_SILU, stream=stream)
        if self.layer_idx == 0:                                             # U = [silu(U) | 1] (:602-606)
            nbuf = len(ws._bufs)
            U = ws.get('gpm_U0', (M, 2 * E), dev)
            if len(ws._bufs) != nbuf:          # freshly allocated: set the constant half once
                U[:, E:].fill_(1.0)
            aot_hip.linear(x1, p['u_w'], p['u_b'], U[:, :E], act=aot_hip.ACT_SILU, stream=stream)
            xi = None
        else:                                                               # U = silu([U | linear_ID_U(LN(tgt_id))]) (:608-611)
            xi = new('xi', M, D)
            aot_hip.layernorm(X[:, D:], *p['id_norm1'], xi, stream=stream)
            U = ws.get('gpm_U', (M, 2 * E), dev)
            aot_hip.linear(x1, p['u_w'], p['u_b'], U[:, :E], act=aot_hip.ACT_SILU, stream=stream)
            aot_hip.linear(xi, p['idu_w'], p['idu_b'], U[:, E:], act=aot_hip.ACT_SILU, stream=stream)
        if id_emb is not None:                                              # reference frame (:613-620)
            self.fuse_id_into(vcat, xi, id_emb, ws, stream)
            gk, gv, t, g_brows, t_dev = qc, vcat, N, N, ()
            lk, lv, l_brows = qc, vcat, N
        else:
            gk, gv, t, g_brows, *t_dev = long_mem
            lk, lv, l_brows = short_mem
        raw = ws.get('gpm_raw', (M, 2 * E), dev)
        self.long_term_attn.core(qc, gk, gv, U, raw, t, ws, stream, t_dev=t_dev[0] if t_dev else None, B=B,
                                 kv_brows=g_brows, x6=x6 if id_emb is None else None)
        Xm = ws.get('gpm_Xm', (M, 2 * D), dev)
        self.long_term_attn.tail(raw, Xm, size_2d, ws, stream, res=X, B=B)      # X + lt
        self.short_term_attn.core(qc, lk, lv, U, raw, size_2d, ws, stream, B=B, kv_brows=l_brows)
        self.short_term_attn.tail(raw, Xm, size_2d, ws, stream, res=Xm, B=B)     # + st   (:633-641)
        # self gated propagation on [LN(tgt) | LN(tgt_id)]  (:643-653)
        z = ws.get('gpm_z', (M, 2 * D), dev)
        aot_hip.layernorm(Xm[:, :D], *p['norm2'], z[:, :D], stream=stream)
        aot_hip.layernorm(Xm[:, D:], *p['id_norm2'], z[:, D:], stream=stream)
        sp = self.self_attn.pack()
        qk = ws.get('gpm_sqk', (M, da), dev)
        aot_hip.linear(z, sp['QK_w'], sp['QK_b'], qk, stream=stream)
        sv = ws.get('gpm_sv', (M, 2 * E), dev)
        su = ws.get('gpm_su', (M, 2 * E), dev)
        # the four value / gate projections: one shape, independent -- one grouped launch in the bf16x6 family (round 6)
        aot_hip.linear_group([z[:, :D], z[:, D:], z[:, :D], z[:, D:]], [sp['V1_w'], sp['V2_w'], sp['U1_w'], sp['U2_w']],
                             [sp['V1_b'], sp['V2_b'], sp['U1_b'], sp['U2_b']], [sv[:, :E], sv[:, E:], su[:, :E], su[:, E:]],
                             act=aot_hip.ACT_SILU, stream=stream)
        sx6 = None
        if x6 is not None and da == 128 and 2 * E == 1024 and not os.environ.get('AOT_NO_SELF_X6'):
            # bf16x6 engines (round 6): the frame's own K / [V1 | V2] split into a one-frame packed bank (aot_attn_pack_x6_part_f32), then the
            # 64-query kernel of the long-term propagation instead of the fp32 kernel (98.6 -> ~81 us per layer at 480p)
            cap = (N + 31) // 32 * 32
            sx6 = (ws.get_zeroed('gpm_sx6_k', (B * cap * da * 3,), dev, torch.int16),
                   ws.get_zeroed('gpm_sx6_v', (B * cap * 2 * E * 3,), dev, torch.int16), cap)
            aot_hip.gated_pack_x6(qk, sv, sx6, N, B=B, src_brows=N, stream=stream)
        self.self_attn.core(qk, qk, sv, su, raw, N, ws, stream, B=B, kv_brows=N, x6=sx6)
        Xo = ws.get('gpm_Xo_%d' % self.layer_idx, (M, 2 * D), dev)
        self.self_attn.tail(raw, Xo, size_2d, ws, stream, res=Xm, B=B)
        return Xo, qc, vcat, xi

    def fuse_key_value_id(self, key, value, id_emb):
        """Reference API (transformer.py:659-665): returns (None, ID_V [N,1,E])."""
        n, b, c = id_emb.shape
        E = self.expand_d_model
        vcat = torch.empty(n * b, 2 * E, dtype=torch.float32, device=id_emb.device)
        prev = value.reshape(n * b, -1).contiguous() if value is not None else None
        from networks.layers.workspace import Workspace
        if not hasattr(self, '_own_ws'):
            self._own_ws = Workspace()
        self.fuse_id_into(vcat, prev, id_emb.reshape(n * b, c).contiguous(), self._own_ws, aot_hip.stream_ptr())
        return None, vcat[:, E:].unsqueeze(1)


class DualBranchGPM(nn.Module):
    """DeAOT stack (reference transformer.py:143-255)."""

    def __init__(self, num_layers=2, d_model=256, self_nhead=8, att_nhead=8, dim_feedforward=1024, emb_dropout=0.,
                 droppath=0.1, lt_dropout=0., st_dropout=0., droppath_lst=False, droppath_scaling=False,
                 activation='gelu', return_intermediate=False, intermediate_norm=True, final_norm=True):
        super().__init__()
        self.intermediate_norm, self.final_norm = intermediate_norm, final_norm
        self.num_layers, self.return_intermediate = num_layers, return_intermediate
        self.d_model = d_model
        self.emb_dropout_p = emb_dropout
        self.layers = nn.ModuleList([
            GatedPropagationModule(d_model, self_nhead, att_nhead, dim_feedforward,
                                   _droppath_rate(droppath, i, num_layers, droppath_scaling), lt_dropout, st_dropout,
                                   droppath_lst, activation, layer_idx=i) for i in range(num_layers)])
        num_norms = (num_layers - 1 if intermediate_norm else 0) + (1 if final_norm else 0)
        self.decoder_norms = nn.ModuleList([GroupNorm1D(d_model * 2, 2) for _ in range(num_norms)]) if num_norms > 0 else None
        if intermediate_norm:
            raise NotImplementedError('DeAOT decodes the last GPM output only (default_deaot.py:12)')

    def run(self, x0, long_mems, short_mems, id_emb, pos, size_2d, ws, stream, B=1, dst=None, keep=None, x6=None):
        """B lanes on the shared feature x0 [N, D].  Returns (dec_in [B*N, 2D] = GroupNorm(2)(cat[tgt, tgt_id]) of the last
        layer, mems): mems[i] = (curr_K, curr_Vcat, curr_ID_V_input) of layer i."""
        N, D = x0.shape
        dev = x0.device
        X = ws.get('gpm_X0', (B * N, 2 * D), dev)
        aot_hip.copy_rows(x0, X, N, B=B, src_brows=0, dst_brows=N, stream=stream)
        X[:, D:].zero_()                                                    # tgt_id = 0 (transformer.py:602-603)
        mems = []
        for i, layer in enumerate(self.layers):
            X, ck, cv, xi = layer.run(X, long_mems[i] if long_mems is not None else None,
                                      short_mems[i] if short_mems is not None else None,
                                      id_emb, pos, size_2d, ws, stream, B=B, dst=dst[i] if dst is not None else None,
                                      keep=keep, x6=x6[i] if x6 is not None else None)
            mems.append((ck, cv, xi))
        if keep is not None:
            out = keep.get('gpm_dec_in', (B * N, 2 * D), dev)
        else:
            out = torch.empty(B * N, 2 * D, dtype=torch.float32, device=dev)
        gn = self.decoder_norms[-1].gn
        aot_hip.groupnorm(X, gn.weight, gn.bias, out, 2, aot_hip.gn_buffers(ws, dev, B, 2, 64), act=aot_hip.ACT_NONE,
                          nsplit=64, B=B, stream=stream)
        return out, mems

    def update_values(self, mems, id_emb, ws, stream, dst=None):
        """deaot_engine.py:29-45: only ID_V = silu(linear_ID_V([prev ID_V,] id_emb)) is refreshed, in place in the
        frame's [V | ID_V] buffer; K and V stay as produced."""
        return [self.layers[i].fuse_id_into(m[1], m[2], id_emb, ws, stream) for i, m in enumerate(mems)]
