"""Swin-B trunk, three stages (reference networks/encoders/swin/{build,swin_transformer}.py), for BASELINE config 3
(SwinB-DeAOTL).  Parameter names/shapes follow the reference; compute on the HIP path:

  patch embed  = 4x4/s4 implicit-GEMM conv on the 4-channel-padded image + LayerNorm
  block        = LN -> qkv GEMM -> fused (shifted) window attention -> proj GEMM (+residual)
                 -> LN -> fc1 GEMM with exact-GELU epilogue -> fc2 GEMM (+residual)
  patch merge  = 2x2 gather -> LN(4C) -> bias-free GEMM
"""
import torch
from torch import nn

import aot_hip
from networks.layers.normalization import fold_conv_bn, linear_t


def _ln(m):
    return m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous()


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij')).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer('relative_position_index', rel.sum(-1))     # kept for state_dict parity (:131-147)
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4.):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.drop_path_p = 0.       # stochastic depth (training-time only: models/train_forward.py), set by SwinTransformer
        self._p = None

    def pack(self):
        if self._p is None:
            p = {'n1': _ln(self.norm1), 'n2': _ln(self.norm2),
                 'table': self.attn.relative_position_bias_table.detach().float().contiguous()}
            for k, m in (('qkv', self.attn.qkv), ('proj', self.attn.proj), ('fc1', self.mlp.fc1), ('fc2', self.mlp.fc2)):
                p[k + '_w'], p[k + '_b'] = linear_t(m)
            self._p = p
        return self._p

    def run(self, x, out, H, W, ws, stream, B=1):
        """x [B*H*W, C] -> out [B*H*W, C] (reference SwinTransformerBlock.forward, :262-318); B images stacked along the rows:
        the GEMMs, LayerNorms and the window attention (image = grid z) see one tall matrix."""
        p = self.pack()
        N, C = x.shape
        dev = x.device
        x1 = ws.get('sw_x1', (N, C), dev)
        qkv = ws.get('sw_qkv', (N, 3 * C), dev)
        # (LayerNorm as the prologue of these GEMMs -- aot_layernorm_linear_bf16x6_f32, as in the LSTT -- was measured here too: no faster
        #  (270.2 against 271.4 fps on SwinB-DeAOTL) and one free-running flip off the reference's near-ties: profiles/r06_swin_ln_fuse.txt)
        aot_hip.layernorm(x, *p['n1'], x1, eps=self.norm1.eps, stream=stream)
        aot_hip.linear(x1, p['qkv_w'], p['qkv_b'], qkv, stream=stream)
        a = ws.get('sw_a', (N, C), dev)
        aot_hip.swin_window_attention(qkv, p['qkv_b'], p['table'], a, H, W, C, self.num_heads, self.shift_size, self.attn.scale, B=B,
                                      stream=stream)          # (one launch for the B images: round 6, profiles/r06_swin_batched_attn.txt)
        xa = ws.get('sw_xa', (N, C), dev)
        aot_hip.linear(a, p['proj_w'], p['proj_b'], xa, res=x, stream=stream)
        f = ws.get('sw_f', (N, 4 * C), dev)
        aot_hip.layernorm(xa, *p['n2'], x1, eps=self.norm2.eps, stream=stream)
        aot_hip.linear(x1, p['fc1_w'], p['fc1_b'], f, act=aot_hip.ACT_GELU, stream=stream)
        aot_hip.linear(f, p['fc2_w'], p['fc2_b'], out, res=xa, stream=stream)
        return out


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)
        self._p = None

    def run(self, x, H, W, ws, stream, B=1):
        if self._p is None:
            self._p = (_ln(self.norm), linear_t(self.reduction)[0])
        (g, b), w = self._p
        C = self.dim
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        dev = x.device
        gth = ws.get('sw_merge', (B * H2 * W2, 4 * C), dev)
        for i in range(B):
            aot_hip.patch_merge(x[i * H * W:(i + 1) * H * W], gth[i * H2 * W2:(i + 1) * H2 * W2], H, W, C, stream=stream)
        out = ws.get('sw_merged_%d' % C, (B * H2 * W2, 2 * C), dev)
        aot_hip.layernorm(gth, g, b, gth, eps=self.norm.eps, stream=stream)
        aot_hip.linear(gth, w, None, out, stream=stream)
        return out, H2, W2


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., downsample=None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio)
            for i in range(depth)])
        self.downsample = downsample(dim) if downsample is not None else None


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96):
        super().__init__()
        self.patch_size, self.embed_dim = patch_size, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)


class SwinTransformer(nn.Module):
    def __init__(self, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.,
                 out_indices=(0, 1, 2), drop_path_rate=0.2):
        super().__init__()
        self.num_layers = len(depths) - 1                     # the reference drops the last stage (:560)
        self.embed_dim, self.out_indices = embed_dim, out_indices
        self.patch_embed = PatchEmbed(4, 3, embed_dim)
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                       PatchMerging if i < self.num_layers - 1 else None) for i in range(self.num_layers)])
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        # stochastic-depth decay rule over the blocks of ALL four stages (swin_transformer.py:601-604), the dropped one included
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        for i, layer in enumerate(self.layers):
            for bi, blk in enumerate(layer.blocks):
                blk.drop_path_p = dpr[sum(depths[:i]) + bi]
        for i in out_indices:
            self.add_module('norm%d' % i, nn.LayerNorm(self.num_features[i]))
        self._pe = None

    batched = True       # run() takes B images at once (AOTEngine.encode_ahead)

    def run(self, img, ws, stream):
        """img [B,3,H,W] -> [(feat [B*h*w, C], h, w)] x 3, the B images stacked along the rows.  Sides that are not multiples
        of the 4x4 patch are zero-padded on the right / bottom (PatchEmbed.forward, swin_transformer.py:501-509): the
        implicit-GEMM loader reads taps beyond the image as zeros, so the padding is just the rounded-up output size."""
        B, _, H, W = img.shape
        dev = img.device
        if self._pe is None:
            self._pe = (fold_conv_bn(self.patch_embed.proj, pad_cin=4), _ln(self.patch_embed.norm))
        (pw, pb), (g, b) = self._pe
        x4 = ws.get('img_nhwc4', (B * H * W, 4), dev)
        for i in range(B):
            aot_hip.nchw_to_nhwc(img[i:i + 1], x4[i * H * W:(i + 1) * H * W], 3, H, W, 4, stream=stream)
        h, w = -(-H // 4), -(-W // 4)
        C = self.embed_dim
        x = ws.get('sw_x_%d_0' % C, (B * h * w, C), dev)
        aot_hip.conv2d(x4, pw, pb, x, H, W, 4, h, w, C, 4, 4, 4, 0, 1, B=B, stream=stream)
        aot_hip.layernorm(x, g, b, x, stream=stream)
        feats = []
        for li, layer in enumerate(self.layers):
            C = self.num_features[li]
            for bi, blk in enumerate(layer.blocks):
                out = ws.get('sw_x_%d_%d' % (C, (bi + 1) & 1), (B * h * w, C), dev)
                x = blk.run(x, out, h, w, ws, stream, B=B)
            if li in self.out_indices:
                nm = getattr(self, 'norm%d' % li)
                f = ws.get('sw_stage%d' % li, (B * h * w, C), dev)
                aot_hip.layernorm(x, nm.weight, nm.bias, f, stream=stream)
                feats.append((f, h, w))
            if layer.downsample is not None:
                x, h, w = layer.downsample.run(x, h, w, ws, stream, B=B)
        return feats


def _freeze_stages(net, frozen_stages):
    """reference swin_transformer.py:637-657: frozen_stages >= 0 stops the gradients of the patch embedding, >= 2 those of the
    first frozen_stages - 1 stages -- except the patch-merging layer of the last of them, which stays trainable."""
    if frozen_stages >= 0:
        for p in net.patch_embed.parameters():
            p.requires_grad = False
    if frozen_stages >= 2:
        last = None
        for i in range(frozen_stages - 1):
            last = net.layers[i]
            for blk in last.blocks:
                blk.drop_path_p = 0.
            for p in last.parameters():
                p.requires_grad = False
        if last is not None and last.downsample is not None:
            for p in last.downsample.parameters():
                p.requires_grad = True


def build_swin_model(model_type, freeze_at=0):
    if model_type == 'swin_base':       # reference swin/build.py:11-27
        net = SwinTransformer(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7,
                              out_indices=(0, 1, 2), drop_path_rate=0.3)
        _freeze_stages(net, freeze_at)
        return net
    raise NotImplementedError('Unknown model: %s' % model_type)
