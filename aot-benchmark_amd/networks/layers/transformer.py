"""LSTT stack of AOT (reference networks/layers/transformer.py:33-140, 258-372).

Per-layer launch sequence on the HIP path (token-major [N, C] activations, N = h*w):
  LN1 (+pos, two outputs) -> [Q|K] GEMM, V GEMM -> flash self-attention -> projection GEMM (+residual)
  LN2 -> linear_Q GEMM -> flash long-term attention over the bank  +  fused windowed short-term attention
  (both write halves of one [N, 512] buffer) -> one K=512 GEMM = proj_lt + proj_st (+residual)
  LN3 -> linear1 GEMM -> GroupNorm(32)+GELU -> 5x5 depthwise conv -> linear2 GEMM (+residual)
"""
import torch
from torch import nn

import aot_hip
from networks.layers.attention import MultiheadAttention, MultiheadLocalAttention
from networks.layers.basic import GNActDWConv2d
from networks.layers.normalization import fold_dwconv_bn, linear_t


def _ln_params(ln):
    return ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous()


class LongShortTermTransformerBlock(nn.Module):
    def __init__(self, d_model, self_nhead, att_nhead, dim_feedforward=1024, droppath=0.1, lt_dropout=0.,
                 st_dropout=0., droppath_lst=False, activation='gelu', local_dilation=1):
        super().__init__()
        self.d_model = d_model
        self.dim_ff = dim_feedforward
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
        p['sa_qk_w'] = torch.cat([wq, wk], 1).contiguous()          # [256, 512]: one GEMM for Q and K of (x1 + pos)
        p['sa_qk_b'] = torch.cat([bq, bk]).contiguous()
        p['sa_v_w'], p['sa_v_b'] = linear_t(sa.linear_V)
        p['sa_o_w'], p['sa_o_b'] = linear_t(sa.projection)
        p['q_w'], p['q_b'] = linear_t(self.linear_Q)
        p['v_w'], p['v_b'] = linear_t(self.linear_V)
        wl, bl = linear_t(self.long_term_attn.projection)
        ws_, bs_ = linear_t(self.short_term_attn.projection)
        p['lst_w'] = torch.cat([wl, ws_], 0).contiguous()           # [512, 256]: lt and st projections in one GEMM
        p['lst_b'] = (bl + bs_).contiguous()
        p['w1'], p['b1'] = linear_t(self.linear1)
        p['w2'], p['b2'] = linear_t(self.linear2)
        p['dw'], _ = fold_dwconv_bn(self.activation.conv)
        for n in ('norm1', 'norm2', 'norm3'):
            p[n] = _ln_params(getattr(self, n))
        p['gn'] = (self.activation.gn.weight.detach().float().contiguous(),
                   self.activation.gn.bias.detach().float().contiguous())
        self.short_term_attn.pack()
        self._p = p
        return p

    # ---- reference transformer.py:312-362 -----------------------------------------------------
    def run(self, x, long_mem, short_mem, id_emb, pos, size_2d, ws, stream, t_long=None):
        """x [N, C(ld)] token-major.  long_mem = (K, V) token-major [>=T, C] with T = t_long;
        short_mem = (K, V) [N, C].  Returns (out [N,C], curr_K, curr_V, global (K,V,T), local (K,V))."""
        p = self.pack()
        N, C = x.shape
        dev = x.device
        h, w = size_2d
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)   # tensors that outlive this call

        # self-attention
        x1 = ws.get('x1', (N, C), dev)
        x1p = ws.get('x1p', (N, C), dev)
        aot_hip.layernorm(x, *p['norm1'], x1, add=pos, out2=x1p, stream=stream)
        qk = ws.get('sa_qk', (N, 2 * C), dev)
        aot_hip.linear(x1p, p['sa_qk_w'], p['sa_qk_b'], qk, stream=stream)
        sv = ws.get('sa_v', (N, C), dev)
        aot_hip.linear(x1, p['sa_v_w'], p['sa_v_b'], sv, stream=stream)
        so = ws.get('sa_o', (N, C), dev)
        self.self_attn.core(qk[:, :C], qk[:, C:], sv, so, N, ws, stream)
        xa = ws.get('xa', (N, C), dev)
        aot_hip.linear(so, p['sa_o_w'], p['sa_o_b'], xa, res=x, stream=stream)

        # long + short term attention
        x2 = new(N, C)                                            # curr_V (normed input, transformer.py:333)
        aot_hip.layernorm(xa, *p['norm2'], x2, stream=stream)
        qc = new(N, C)                                            # curr_Q == curr_K (:331-332)
        aot_hip.linear(x2, p['q_w'], p['q_b'], qc, stream=stream)
        if id_emb is not None:                                    # reference frame: memorise itself (:337-341)
            gk, gv = qc, self.fuse_kv_2d(x2, id_emb, ws, stream)
            lk, lv, t = gk, gv, N
        else:
            gk, gv = long_mem
            lk, lv = short_mem
            t = t_long if t_long is not None else gk.shape[0]
        cat = ws.get('lst_cat', (N, 2 * C), dev)
        self.long_term_attn.core(qc, gk, gv, cat[:, :C], t, ws, stream)
        self.short_term_attn.core(qc, lk, lv, cat[:, C:], size_2d, stream)
        xb = ws.get('xb', (N, C), dev)
        aot_hip.linear(cat, p['lst_w'], p['lst_b'], xb, res=xa, stream=stream)

        # feed-forward: linear1 -> GN(32)+GELU -> dw5x5 -> linear2
        x3 = ws.get('x3', (N, C), dev)
        aot_hip.layernorm(xb, *p['norm3'], x3, stream=stream)
        F1 = self.dim_ff
        f = ws.get('ffn_a', (N, F1), dev)
        aot_hip.linear(x3, p['w1'], p['b1'], f, stream=stream)
        g = ws.get('ffn_b', (N, F1), dev)
        aot_hip.groupnorm(f, *p['gn'], g, 32, ws.get('gn_scratch', (32 * 64 * 2,), dev, torch.float64),
                          ws.get('gn_stats', (64,), dev, torch.float64), act=aot_hip.ACT_GELU, nsplit=32, stream=stream)
        aot_hip.dwconv2d(g, p['dw'], None, f, h, w, F1, h, w, 5, 1, 2, 1, stream=stream)
        out = ws.get('layer_out_%d' % id(self), (N, C), dev)
        aot_hip.linear(f, p['w2'], p['b2'], out, res=xb, stream=stream)
        return out, qc, x2, (gk, gv, t), (lk, lv)

    def fuse_kv_2d(self, v, id_emb, ws, stream):
        p = self.pack()
        N, C = v.shape
        tmp = ws.get('fuse_tmp', (N, C), v.device)
        aot_hip.add(v, id_emb, tmp, stream=stream)
        out = torch.empty(N, C, dtype=torch.float32, device=v.device)
        aot_hip.linear(tmp, p['v_w'], p['v_b'], out, stream=stream)
        return out

    def fuse_key_value_id(self, key, value, id_emb):
        """Reference API (transformer.py:364-367): K unchanged, V <- linear_V(V + id_emb); [N,1,C] tensors."""
        n, b, c = value.shape
        v2 = self.fuse_kv_2d(value.reshape(n * b, c), id_emb.reshape(n * b, c).contiguous(), self._ws(), aot_hip.stream_ptr())
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
        self.layers = nn.ModuleList([
            LongShortTermTransformerBlock(d_model, self_nhead, att_nhead, dim_feedforward, droppath, lt_dropout,
                                          st_dropout, droppath_lst, activation) for _ in range(num_layers)])
        num_norms = (num_layers - 1 if intermediate_norm else 0) + (1 if final_norm else 0)
        self.decoder_norms = nn.ModuleList([nn.LayerNorm(d_model) for _ in range(num_norms)]) if num_norms > 0 else None

    def run(self, x, long_mems, short_mems, id_emb, pos, size_2d, ws, stream, out_cat, t_long=None):
        """Runs the stack; layer outputs (after their decoder norm, transformer.py:124-135) are written to
        column blocks 1.. of ``out_cat`` [N, (L+1)*C] (block 0 = projected encoder feature), which is the
        decoder's concatenated input (models/aot.py:86-92) -- the concat is never a separate copy."""
        C = x.shape[1]
        L = self.num_layers
        outs, mems = [], []
        for i, layer in enumerate(self.layers):
            x, ck, cv, glob, loc = layer.run(x, long_mems[i] if long_mems is not None else None,
                                             short_mems[i] if short_mems is not None else None,
                                             id_emb, pos, size_2d, ws, stream, t_long)
            mems.append((ck, cv, glob, loc))
            is_last = i == L - 1
            norm = None
            if self.decoder_norms is not None:
                if is_last and self.final_norm:
                    norm = self.decoder_norms[-1]
                elif not is_last and self.return_intermediate and self.intermediate_norm:
                    norm = self.decoder_norms[i]
            dst = out_cat[:, (i + 1) * C:(i + 2) * C]
            if norm is not None:
                aot_hip.layernorm(x, norm.weight, norm.bias, dst, stream=stream)
            else:
                dst.copy_(x)
            outs.append(dst)
        return outs, mems
