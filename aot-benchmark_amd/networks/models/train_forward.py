"""The training step's DIFFERENTIABLE forward (reference networks/engines/aot_engine.py:33-108 as
trainer.py:460-519 drives it: `loss.backward()` -> clip -> AdamW -> EMA).

The inference path runs fused kernels on packed weights and keeps no activations; a training step needs the autograd
graph.  Here the same networks are written once more over the few differentiable primitives of
`networks/layers/train_ops.py` (every graph node = one C-ABI kernel with a hand-written backward), directly on the
modules' `nn.Parameter`s, so that `loss.backward()` fills `.grad` exactly as it does for the reference:

    MobileNetV2 trunk (FrozenBN folded on the fly; frozen stages record no graph)      mobilenetv2.py:219-224
    LongShortTermTransformerBlock / GatedPropagationModule and their attentions        transformer.py:312-367,582-665
    FPNSegmentationHead, the identity bank, the sine position embedding                fpn.py:34-58, aot.py:50-79
    the engine's frame recurrence: reference frame, optional second self-memorising frame, propagated frames with
    ground-truth / prediction / probability feedback (back-propagation through time comes from autograd)

Scope: every trunk the package has -- MobileNetV2, ResNet-50 / 101, Swin-B (AOT-T/S/B/L, DeAOT-T/S/B/L, R50-/R101-AOTL,
R50-/R101-DeAOTL, SwinB-AOTL, SwinB-DeAOTL; BASELINE config 5 trains R50-DeAOTL).  One sample at a time (the reference batches; every op on this
path is per-sample).  Drop-path / Dropout2d follow the modules' `training` flag with torch's generator (they are
identities in eval mode, which is how the gradient goldens were made)."""
import torch
import torch.nn.functional as F

from networks.layers import train_ops as T
from networks.layers.transformer import DualBranchGPM


# ---- small pieces ------------------------------------------------------------------------------------------------------
def _fold_bn(weight, bn):
    """conv weight [Cout, ...] and FrozenBatchNorm2d (constants) -> (weight * scale, shift)."""
    scale = (bn.weight * (bn.running_var + bn.epsilon).rsqrt()).detach()
    shift = (bn.bias - bn.running_mean * scale).detach()
    return weight * scale.view(-1, *([1] * (weight.dim() - 1))), shift


def _drop_path(x, p, training):
    """DropPath of one sample (basic.py:129-148, batch_dim = 1): the whole branch dropped with probability p."""
    if not training or not p:
        return x
    keep = 1. - p
    return x * ((torch.rand((), device=x.device) < keep).to(x.dtype) / keep)


def _dropout(x, p, training):
    """nn.Dropout (elementwise)."""
    if not training or not p:
        return x
    keep = 1. - p
    return x * ((torch.rand_like(x) < keep).to(x.dtype) / keep)


def _dropout2d(x, p, training):
    """nn.Dropout2d on a token-major map [N, C]: whole channels dropped (basic.py:46,55)."""
    if not training or not p:
        return x
    keep = 1. - p
    return x * ((torch.rand(1, x.shape[1], device=x.device) < keep).to(x.dtype) / keep)


def _no_attn_dropout(m):
    if m.training and m.dropout_p:
        raise NotImplementedError('dropout on the attention weights (TRAIN_LSTT_LT_DROPOUT / ST_DROPOUT, 0 in every reference '
                                  'config) is not part of the differentiable forward')


def _cbr(x, seq, H, W):
    """ConvBNActivation (conv / depthwise conv + FrozenBN + ReLU6), mobilenetv2.py:30-46."""
    conv, bn = seq[0], seq[1]
    w, b = _fold_bn(conv.weight, bn)
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    if conv.groups == 1:
        y, OH, OW = T.conv2d(x, w, b, 1, H, W, s, p, d)
    else:
        y, OH, OW = T.dwconv2d(x, w, 1, H, W, s, p, d)
        y = y + b
    return T.act(y, 'relu6'), OH, OW


def mobilenetv2_features(enc, img):
    """img [1, 3, H, W] -> [(f4, h, w), (f8, h, w), (f16, h, w), (top, h, w)] token-major (mobilenetv2.py:219-224)."""
    _, _, H, W = img.shape
    x = T.to_nhwc(img.float(), 4)
    x, h, w = _cbr(x, enc.features[0], H, W)
    feats = []
    for idx in range(1, 18):
        blk = enc.features[idx]
        y, hh, ww = x, h, w
        j = 0
        if blk.expand:
            y, hh, ww = _cbr(y, blk.conv[0], hh, ww)
            j = 1
        y, hh, ww = _cbr(y, blk.conv[j], hh, ww)
        wpl, bpl = _fold_bn(blk.conv[j + 1].weight, blk.conv[j + 2])
        y, hh, ww = T.conv2d(y, wpl, bpl, 1, hh, ww)
        x = x + y if blk.use_res_connect else y
        h, w = hh, ww
        if idx in (3, 6, 13):
            feats.append((x, h, w))
    x, h, w = _cbr(x, enc.features[18], h, w)
    feats.append((x, h, w))
    return feats


def resnet_features(enc, img):
    """ResNet-50 / 101 trunk (encoders/resnet.py:140-157): img [1, 3, H, W] -> [(f4, h, w), (f8, h, w), (f16, h, w)] token-major.
    conv + FrozenBN folded per call; the stem's max pool has no backward kernel -- with the stem frozen
    (TRAIN_ENCODER_FREEZE_AT >= 1, every reference recipe) nothing asks for one."""
    _, _, H, W = img.shape
    x = T.to_nhwc(img.float(), 4)
    w, b = _fold_bn(enc.conv1.weight, enc.bn1)
    x, h, wd = T.conv2d(x, w, b, 1, H, W, 2, 3, 1)
    x = T.act(x, 'relu')
    x, h, wd = T.maxpool3x3s2(x, h, wd)
    feats = []
    for layer in (enc.layer1, enc.layer2, enc.layer3):
        for blk in layer:
            s, d = blk.stride, blk.dilation
            w1, b1 = _fold_bn(blk.conv1.weight, blk.bn1)
            y = T.act(T.conv2d(x, w1, b1, 1, h, wd)[0], 'relu')
            w2, b2 = _fold_bn(blk.conv2.weight, blk.bn2)
            y, oh, ow = T.conv2d(y, w2, b2, 1, h, wd, s, d, d)
            y = T.act(y, 'relu')
            w3, b3 = _fold_bn(blk.conv3.weight, blk.bn3)
            y = T.conv2d(y, w3, b3, 1, oh, ow)[0]
            if blk.downsample is not None:
                wds, bds = _fold_bn(blk.downsample[0].weight, blk.downsample[1])
                res = T.conv2d(x, wds, bds, 1, h, wd, s, 0, 1)[0]                # (stride 2: im2col of a 1x1 window = every 2nd pixel)
            else:
                res = x
            x, h, wd = T.act(y + res, 'relu'), oh, ow
        feats.append((x, h, wd))
    return feats


def swin_features(enc, img):
    """Swin trunk, three stages (encoders/swin/swin_transformer.py:684-716): img [1, 3, H, W] -> [(f4, h, w), (f8, h, w),
    (f16, h, w)] token-major.  Window partition / cyclic shift / padding are index plumbing (views, roll, pad, permute); the
    arithmetic -- LayerNorm, the qkv / proj / MLP linears, QK^T + relative-position bias (+ shift mask) -> softmax -> PV per
    window and head, GELU -- runs on the differentiable primitives."""
    _, _, H, W = img.shape
    pe = enc.patch_embed
    ps = pe.patch_size
    if H % ps or W % ps:          # sides that are not multiples of the patch: zero-padded right / bottom (:501-509)
        img = F.pad(img.float(), (0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))
    x, h, w = T.conv2d(T.to_nhwc(img.float(), 4), pe.proj.weight, pe.proj.bias, 1, img.shape[2], img.shape[3], ps, 0, 1)
    x = T.layernorm(x, pe.norm.weight, pe.norm.bias)
    feats = []
    for li, layer in enumerate(enc.layers):
        C = x.shape[1]
        ws = layer.blocks[0].window_size
        shift = ws // 2
        hp, wp = -(-h // ws) * ws, -(-w // ws) * ws
        nw = (hp // ws) * (wp // ws)
        # BasicLayer.forward :392-411: region labels of the shifted map -> additive -100 mask between tokens of different regions
        reg = torch.zeros(hp, wp, device=x.device)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                reg[hs, wsl] = cnt
                cnt += 1
        mw = reg.view(hp // ws, ws, wp // ws, ws).permute(0, 2, 1, 3).reshape(nw, ws * ws)
        amask = (mw.unsqueeze(1) != mw.unsqueeze(2)).float() * -100.0                         # [nw, 49, 49]
        for blk in layer.blocks:
            nh, sh, Nt = blk.num_heads, blk.shift_size, ws * ws
            at = blk.attn
            y = T.layernorm(x, blk.norm1.weight, blk.norm1.bias).view(h, w, C)
            y = F.pad(y, (0, 0, 0, wp - w, 0, hp - h))
            if sh:
                y = torch.roll(y, shifts=(-sh, -sh), dims=(0, 1))
            win = y.view(hp // ws, ws, wp // ws, ws, C).permute(0, 2, 1, 3, 4).reshape(nw * Nt, C)
            qkv = T.linear(win, at.qkv.weight, at.qkv.bias).view(nw, Nt, 3, nh, C // nh).permute(2, 0, 3, 1, 4)   # [3, nw, nh, 49, d]
            q, k, v = (t.reshape(nw * nh, Nt, C // nh) for t in (qkv[0], qkv[1], qkv[2]))
            att = T.matmul(q, k.transpose(1, 2), alpha=at.scale)                               # (q * scale) k^T, :181-182
            rpb = at.relative_position_bias_table[at.relative_position_index.view(-1)].view(Nt, Nt, nh).permute(2, 0, 1)
            att = att.view(nw, nh, Nt, Nt) + rpb.unsqueeze(0)
            if sh:
                att = att + amask.unsqueeze(1)
            o = T.matmul(T.softmax_rows(att.reshape(nw * nh, Nt, Nt)), v)                      # [nw*nh, 49, d]
            o = o.view(nw, nh, Nt, C // nh).permute(0, 2, 1, 3).reshape(nw * Nt, C)
            o = T.linear(o, at.proj.weight, at.proj.bias)
            o = o.view(hp // ws, wp // ws, ws, ws, C).permute(0, 2, 1, 3, 4).reshape(hp, wp, C)
            if sh:
                o = torch.roll(o, shifts=(sh, sh), dims=(0, 1))
            x = x + _drop_path(o[:h, :w].reshape(h * w, C), blk.drop_path_p, blk.training)
            f = T.act(T.linear(T.layernorm(x, blk.norm2.weight, blk.norm2.bias), blk.mlp.fc1.weight, blk.mlp.fc1.bias), 'gelu')
            x = x + _drop_path(T.linear(f, blk.mlp.fc2.weight, blk.mlp.fc2.bias), blk.drop_path_p, blk.training)
        if li in enc.out_indices:
            nm = getattr(enc, 'norm%d' % li)
            feats.append((T.layernorm(x, nm.weight, nm.bias), h, w))
        if layer.downsample is not None:                                                       # PatchMerging :338-359
            ds = layer.downsample
            y = x.view(h, w, C)
            if h % 2 or w % 2:
                y = F.pad(y, (0, 0, 0, w % 2, 0, h % 2))
            y = torch.cat([y[0::2, 0::2], y[1::2, 0::2], y[0::2, 1::2], y[1::2, 1::2]], -1)
            h, w = (h + 1) // 2, (w + 1) // 2
            x = T.linear(T.layernorm(y.reshape(h * w, 4 * C), ds.norm.weight, ds.norm.bias), ds.reduction.weight)
    return feats


def encoder_features(enc, img):
    """-> ([f4, f8, f16] shortcuts, (top, h, w) the map the projector reads), aot.py:81-84."""
    kind = type(enc).__name__
    if kind == 'MobileNetV2':
        feats = mobilenetv2_features(enc, img)
        return feats[:3], feats[3]
    if kind == 'ResNet':
        feats = resnet_features(enc, img)
        return feats, feats[2]
    if kind == 'SwinTransformer':
        feats = swin_features(enc, img)
        return feats, feats[2]
    raise NotImplementedError('the differentiable training forward covers the MobileNetV2, ResNet and Swin trunks; got %s' % kind)


def _heads(t, H):
    """[N, H*d] -> [H, N, d] view."""
    n, c = t.shape
    return t.view(n, H, c // H).permute(1, 0, 2)


def _attention(q, k, v, H, scale):
    """softmax((q / scale) k^T) v per head (attention.py:82-117): q [N, C], k / v [T, C] -> [N, C]."""
    n, c = q.shape
    s = T.matmul(_heads(q / scale, H), _heads(k, H).transpose(1, 2))          # [H, N, T]
    p = T.softmax_rows(s)
    o = T.matmul(p, _heads(v, H))                                              # [H, N, d]
    return o.permute(1, 0, 2).reshape(n, c)


def _local_attention(m, q, k, v, size_2d):
    """MultiheadLocalAttentionV2 core (attention.py:308-376): q, k, v [N, C] of the current / previous frame."""
    h, w = size_2d
    H, d, R = m.num_head, m.hidden_dim, m.max_dis
    n, c = q.shape
    W2 = m.window_size * m.window_size
    qh = _heads(q, H)                                                          # UNSCALED q for the relative-position term (:327)
    rel = T.matmul(qh, m.relative_emb_k.weight.view(H, W2, d).transpose(1, 2)) + m.relative_emb_k.bias.view(H, 1, W2)
    dense = T.matmul(_heads(q / m.T, H), _heads(k, H).transpose(1, 2))         # [H, N, N]
    s = T.window_gather(dense, h, w, R, float('-inf')) + rel                   # outside the image: -inf (the reference's -1e8)
    a = T.softmax_rows(s)                                                      # [H, N, 225]
    o = T.matmul(T.window_scatter(a, h, w, R, 0.0), _heads(v, H)) + T.matmul(a, m.relative_emb_v.transpose(1, 2))
    return o.permute(1, 0, 2).reshape(n, c)


def _gated_tail(m, agg, u, size_2d):
    """(agg * u) -> depthwise 5x5 -> projection (attention.py:707-710, 855-860)."""
    h, w = size_2d
    x = agg * u
    x, _, _ = T.dwconv2d(x, m.dw_conv.conv.weight, 1, h, w, 1, 2, 1)
    x = _dropout2d(x, getattr(m.dw_conv, 'dropout_p', 0.), m.training)
    return T.linear(x, m.projection.weight, m.projection.bias)


def _gated_global(m, q, k, v, u, size_2d):
    """GatedPropagation core (attention.py:672-710), single head: q, k [., 128], v [T, E], u [N, E]."""
    s = T.matmul((q / m.T).unsqueeze(0), k.t().unsqueeze(0))                   # [1, N, T]
    p = T.softmax_rows(s)
    agg = T.matmul(p, v.unsqueeze(0))[0]
    return _gated_tail(m, agg, u, size_2d)


def _gated_local(m, q, k, v, u, size_2d):
    """LocalGatedPropagation core (attention.py:814-860), single head."""
    h, w = size_2d
    R = m.max_dis
    W2 = m.window_size * m.window_size
    rel = T.linear(q, m.relative_emb_k.weight.view(W2, -1), m.relative_emb_k.bias).unsqueeze(0)      # unscaled q (:814)
    dense = T.matmul((q / m.T).unsqueeze(0), k.t().unsqueeze(0))               # [1, N, N]
    a = T.softmax_rows(T.window_gather(dense, h, w, R, float('-inf')) + rel)
    agg = T.matmul(T.window_scatter(a, h, w, R, 0.0), v.unsqueeze(0))[0]
    return _gated_tail(m, agg, u, size_2d)


# ---- LSTT block (AOT) --------------------------------------------------------------------------------------------------
def lstt_block(blk, x, long_mem, short_mem, id_emb, pos, size_2d):
    """LongShortTermTransformerBlock.forward (transformer.py:312-362) on token-major x [N, C].  long_mem / short_mem = (K, V);
    id_emb given: the frame memorises itself.  Returns (x, [K, V_normed], [K_g, V_g])."""
    h, w = size_2d
    sa = blk.self_attn
    for m in (sa, blk.long_term_attn, blk.short_term_attn):
        _no_attn_dropout(m)
    dp, tr = blk.droppath_p, blk.training
    x1 = T.layernorm(x, blk.norm1.weight, blk.norm1.bias)
    qk = x1 + pos
    q = T.linear(qk, sa.linear_Q.weight, sa.linear_Q.bias)
    k = T.linear(qk, sa.linear_K.weight, sa.linear_K.bias)
    v = T.linear(x1, sa.linear_V.weight, sa.linear_V.bias)
    o = _attention(q, k, v, sa.num_head, sa.T)
    x = x + _drop_path(T.linear(o, sa.projection.weight, sa.projection.bias), dp, tr)
    x2 = T.layernorm(x, blk.norm2.weight, blk.norm2.bias)
    qc = T.linear(x2, blk.linear_Q.weight, blk.linear_Q.bias)
    kc, vc = qc, x2
    if id_emb is not None:
        kg, vg = fuse_kv(blk, kc, vc, id_emb)
        kl, vl = kg, vg
    else:
        kg, vg = long_mem
        kl, vl = short_mem
    lt = blk.long_term_attn
    a_lt = T.linear(_attention(qc, kg, vg, lt.num_head, lt.T), lt.projection.weight, lt.projection.bias)
    st = blk.short_term_attn
    a_st = T.linear(_local_attention(st, qc, kl, vl, size_2d), st.projection.weight, st.projection.bias)
    x = x + (_drop_path(a_lt + a_st, dp, tr) if blk.droppath_lst else _dropout(a_lt + a_st, blk.lst_dropout_p, tr))     # :350-353
    x3 = T.layernorm(x, blk.norm3.weight, blk.norm3.bias)
    f = T.linear(x3, blk.linear1.weight, blk.linear1.bias)
    f = T.act(T.groupnorm(f, blk.activation.gn.weight, blk.activation.gn.bias, blk.activation.gn.num_groups), 'gelu')
    f, _, _ = T.dwconv2d(f, blk.activation.conv.weight, 1, h, w, 1, 2, 1)
    x = x + _drop_path(T.linear(f, blk.linear2.weight, blk.linear2.bias), dp, tr)
    return x, [kc, vc], [kg, vg]


def fuse_kv(blk, k, v, id_emb):
    """fuse_key_value_id (transformer.py:364-367): K unchanged, V <- linear_V(V + id_emb)."""
    return k, T.linear(v + id_emb, blk.linear_V.weight, blk.linear_V.bias)


# ---- GPM block (DeAOT) -------------------------------------------------------------------------------------------------
def gpm_fuse_id(blk, idv, id_emb):
    """fuse_key_value_id (transformer.py:659-665): ID_V = silu(linear_ID_V([ID_V,] id_emb))."""
    z = id_emb if idv is None else torch.cat([idv, id_emb], 1)
    return T.act(T.linear(z, blk.linear_ID_V.weight, blk.linear_ID_V.bias), 'silu')


def gpm_block(blk, x, x_id, long_mem, short_mem, id_emb, size_2d):
    """GatedPropagationModule.forward (transformer.py:582-657).  Memories = (K, V, ID_V).  Returns (x, x_id, [K, V, ID_V in],
    [K_g, V_g, ID_V_g])."""
    D, E, da = blk.d_model, blk.expand_d_model, blk.d_att * blk.att_nhead
    for m in (blk.self_attn, blk.long_term_attn, blk.short_term_attn):
        _no_attn_dropout(m)
        if m.num_head != 1:
            raise NotImplementedError('the gated propagation of the reference configs is single-head (default_deaot.py)')
    dp, tr = blk.droppath_p, blk.training
    x1 = T.layernorm(x, blk.norm1.weight, blk.norm1.bias)
    qv = T.linear(x1, blk.linear_QV.weight, blk.linear_QV.bias)
    qc = qv[:, :da]
    vc = T.act(qv[:, da:], 'silu')
    kc = qc
    uc = T.linear(x1, blk.linear_U.weight, blk.linear_U.bias)
    if blk.layer_idx == 0:
        u = torch.cat([T.act(uc, 'silu'), torch.ones_like(uc)], 1)
        idvc = None
    else:
        xi = T.layernorm(x_id, blk.id_norm1.weight, blk.id_norm1.bias)
        idvc = xi
        u = T.act(torch.cat([uc, T.linear(xi, blk.linear_ID_U.weight, blk.linear_ID_U.bias)], 1), 'silu')
    if id_emb is not None:
        kg, vg, idvg = kc, vc, gpm_fuse_id(blk, idvc, id_emb)
        kl, vl, idvl = kg, vg, idvg
    else:
        kg, vg, idvg = long_mem
        kl, vl, idvl = short_mem
    lt = _gated_global(blk.long_term_attn, qc, kg, torch.cat([vg, idvg], 1), u, size_2d)
    st = _gated_local(blk.short_term_attn, qc, kl, torch.cat([vl, idvl], 1), u, size_2d)
    y = lt + st
    if blk.droppath_lst:                                                       # :633-638: the two halves draw separately
        ya, yb = _drop_path(y[:, :D], dp, tr), _drop_path(y[:, D:], dp, tr)
    else:
        ya, yb = _dropout(y[:, :D], blk.lst_dropout_p, tr), _dropout(y[:, D:], blk.lst_dropout_p, tr)
    x = x + ya
    x_id = yb if x_id is None else x_id + yb
    z1 = T.layernorm(x, blk.norm2.weight, blk.norm2.bias)
    z2 = T.layernorm(x_id, blk.id_norm2.weight, blk.id_norm2.bias)
    z = torch.cat([z1, z2], 1)
    sa = blk.self_attn
    qk = T.linear(z, sa.linear_QK.weight, sa.linear_QK.bias)
    sv = T.act(torch.cat([T.linear(z1, sa.linear_V1.weight, sa.linear_V1.bias), T.linear(z2, sa.linear_V2.weight, sa.linear_V2.bias)], 1), 'silu')
    su = T.act(torch.cat([T.linear(z1, sa.linear_U1.weight, sa.linear_U1.bias), T.linear(z2, sa.linear_U2.weight, sa.linear_U2.bias)], 1), 'silu')
    y = _gated_global(sa, qk, qk, sv, su, size_2d)
    return x + _drop_path(y[:, :D], dp, tr), x_id + _drop_path(y[:, D:], dp, tr), [kc, vc, idvc], [kg, vg, idvg]


# ---- decoder -----------------------------------------------------------------------------------------------------------
def fpn_decoder(dec, x_in, shortcuts, size_2d):
    """FPNSegmentationHead.forward (fpn.py:34-58): x_in [N16, in_dim], shortcuts = [(f4, h, w), (f8, h, w), (f16, h, w)] ->
    logits [h4*w4, out_dim], h4, w4."""
    (s4, h4, w4), (s8, h8, w8), (s16, h16, w16) = shortcuts

    def cgn(m, x, hh, ww):      # ConvGN (basic.py:75-85) + ReLU
        k = m.conv.kernel_size[0]
        y, _, _ = T.conv2d(x, m.conv.weight, m.conv.bias, 1, hh, ww, 1, k // 2, 1)
        return T.act(T.groupnorm(y, m.gn.weight, m.gn.bias, m.gn.num_groups), 'relu')

    def adapter(m, s, hh, ww):
        return T.conv2d(s, m.weight, m.bias, 1, hh, ww)[0]
    x = cgn(dec.conv_in, x_in, h16, w16) + adapter(dec.adapter_16x, s16, h16, w16)
    x = cgn(dec.conv_16x, x, h16, w16)
    x = T.bilinear(x, 1, h16, w16, h8, w8, dec.align_corners) + adapter(dec.adapter_8x, s8, h8, w8)
    x = cgn(dec.conv_8x, x, h8, w8)
    x = T.bilinear(x, 1, h8, w8, h4, w4, dec.align_corners) + adapter(dec.adapter_4x, s4, h4, w4)
    x = cgn(dec.conv_4x, x, h4, w4)
    return T.conv2d(x, dec.conv_out.weight, dec.conv_out.bias, 1, h4, w4)[0], h4, w4


# ---- one clip's recurrence ---------------------------------------------------------------------------------------------
class ClipGraph:
    """The frame recurrence of one sample (aot_engine.py:188-354 under autograd): memories are graph tensors, the long-term
    bank is the concatenation of the memorised frames."""

    def __init__(self, model, long_term_mem_gap=9999):
        self.m = model
        self.deaot = isinstance(model.LSTT, DualBranchGPM)
        self.gap = long_term_mem_gap
        self.frame_step = 0
        self.last_mem_step = -1
        self.long = None           # per layer: list of per-frame memories (concatenated at use)
        self.short = None
        self.curr = None
        self.pos = None
        self.size_2d = None
        self.feats = None
        self.dec_in = None

    def _encode(self, img):
        self.feats, (top, h, w) = encoder_features(self.m.encoder, img)
        proj = self.m.encoder_projector
        x16 = T.conv2d(top, proj.weight, proj.bias, 1, h, w)[0]
        if self.size_2d is None:
            self.size_2d = (h, w)
            with torch.no_grad():
                pe = self.m.get_pos_emb(torch.zeros(1, x16.shape[1], h, w, device=img.device))
            self.pos = pe[0].permute(1, 2, 0).reshape(h * w, -1).contiguous()
        return x16

    def id_emb(self, one_hot):
        """one-hot or probability map [1, L, H, W] -> identity embedding [N, C] (aot.py:76-79, deaot.py:51-55)."""
        bank = self.m.patch_wise_id_bank
        H, W = one_hot.shape[-2:]
        x = T.to_nhwc(one_hot.float(), (one_hot.shape[1] + 3) // 4 * 4)
        e, _, _ = T.conv2d(x, bank.weight, bank.bias, 1, H, W, bank.stride[0], bank.padding[0], 1)
        if self.deaot:
            e = T.layernorm(e, self.m.id_norm.weight, self.m.id_norm.bias)
        return _dropout(e, self.m.id_dropout_p, self.m.training)

    def _lstt(self, x16, id_emb):
        """LongShortTermTransformer.forward / DualBranchGPM.forward (transformer.py:94-140, 205-255) and the decoder's input
        (aot.py:86-92, fpn.py:34-38).  Returns the per-layer memories this frame would memorise (id_emb given)."""
        stack = self.m.LSTT
        layers = stack.layers
        L = len(layers)
        ref = id_emb is not None
        x = x0 = _dropout(x16, stack.emb_dropout_p, stack.training)
        curr, new_long, outs = [], [], []
        x_id = None
        for i, blk in enumerate(layers):
            long_m = None if ref else [torch.cat(t, 0) for t in zip(*self.long[i])]
            short_m = None if ref else self.short[i]
            if self.deaot:
                x, x_id, c, g = gpm_block(blk, x, x_id, long_m, short_m, id_emb, self.size_2d)
                outs.append(torch.cat([x, x_id], 1))
            else:
                x, c, g = lstt_block(blk, x, long_m, short_m, id_emb, self.pos, self.size_2d)
                outs.append(x)
            curr.append(c)
            new_long.append(g)

        def norm(n, t):
            if self.deaot:
                return T.groupnorm(t, n.gn.weight, n.gn.bias, n.gn.num_groups)
            return T.layernorm(t, n.weight, n.bias)
        if stack.decoder_norms is not None:
            if stack.final_norm:
                outs[-1] = norm(stack.decoder_norms[-1], outs[-1])
            if stack.intermediate_norm:
                for i in range(L - 1):
                    outs[i] = norm(stack.decoder_norms[i], outs[i])
        self.dec_in = torch.cat([x0] + outs, 1) if self.m.decoder.decode_intermediate_input else outs[-1]
        self.curr = curr
        return new_long

    def add_reference_frame(self, img, one_hot, frame_step=None):
        """aot_engine.py:188-251 (also set_prev_frame, :253-289): the frame memorises its own mask."""
        if frame_step is not None:
            self.frame_step = frame_step
        x16 = self._encode(img)
        mems = self._lstt(x16, self.id_emb(one_hot))
        if self.long is None:
            self.long = [[m] for m in mems]
        else:
            for bank, m in zip(self.long, mems):
                bank.append(m)
        self.last_mem_step = self.frame_step
        self.short = mems

    def match_propogate_one_frame(self, img):
        self.frame_step += 1
        self._lstt(self._encode(img), None)

    def decode_logits(self, out_size, obj_num):
        """decode_current_logits (aot_engine.py:356-380): stride-4 logits -> output size, [1, L, H, W]; the channels of unused
        identities are constants (-1e10) that carry no gradient."""
        logits, h4, w4 = fpn_decoder(self.m.decoder, self.dec_in, self.feats, self.size_2d)
        L = logits.shape[1]
        lp = F.pad(logits, (0, (L + 3) // 4 * 4 - L))
        up = T.bilinear(lp, 1, h4, w4, out_size[0], out_size[1], self.m.cfg.MODEL_ALIGN_CORNERS)[:, :L]
        return T.to_nchw(up, out_size[0], out_size[1])

    def update_memory(self, one_hot):
        """update_short_term_memory (aot_engine.py:307-338 / deaot_engine.py:20-56) with the frame's identity embedding."""
        e = self.id_emb(one_hot)
        mems = []
        for blk, c in zip(self.m.LSTT.layers, self.curr):
            if self.deaot:
                mems.append([c[0], c[1], gpm_fuse_id(blk, c[2], e)])
            else:
                mems.append(list(fuse_kv(blk, c[0], c[1], e)))
        self.short = mems
        if self.frame_step - self.last_mem_step >= self.gap:
            for bank, m in zip(self.long, mems):
                bank.append(m)
            self.last_mem_step = self.frame_step


def one_hot(mask, num_classes):
    """utils/image.py:69-74: label map [1, 1, H, W] -> [1, num_classes + 1, H, W] (labels beyond the bank give all zeros)."""
    ids = torch.arange(num_classes + 1, device=mask.device, dtype=mask.dtype).view(1, -1, 1, 1)
    return (mask == ids).float()


def training_forward(engine, all_frames, all_masks, batch_size, obj_nums, step=0, use_prev_pred=False, enable_prev_frame=False,
                     use_prev_prob=False):
    """aot_engine.py:33-108 with an autograd graph: same arguments and return values as AOTEngine.forward."""
    model = engine.AOT
    model._params_touched = True           # a training step follows: the inference path re-packs its weight copies when next used
    bs = int(batch_size)
    T_ = all_frames.shape[0] // bs
    L = model.max_obj_num + 1
    aux_weight = engine.aux_weight * max(engine.aux_step - step, 0.) / engine.aux_step
    n_aux = 2 if enable_prev_frame else 1
    frames = all_frames.view(T_, bs, *all_frames.shape[1:])
    masks = all_masks.view(T_, bs, *all_masks.shape[1:]).float()
    losses = [[None] * bs for _ in range(T_)]
    preds = [[None] * bs for _ in range(T_)]
    for b in range(bs):
        clip = ClipGraph(model, engine.long_term_mem_gap)
        objs = int(obj_nums[b])
        perm = engine.id_shuffle[b] if engine.enable_id_shuffle else None       # identity o is moved to channel perm[o]
        inv = None if perm is None else torch.argsort(perm)

        def ident(m):           # what assign_identity sees (aot_engine.py:168-179): the (shuffled) one-hot / probability map
            oh = m if m.shape[1] == L else one_hot(m, model.max_obj_num)
            return oh if perm is None else oh[:, inv]

        def score(t):
            gt = masks[t, b:b + 1]
            size = tuple(gt.shape[-2:])
            lg = clip.decode_logits(size, objs)
            if perm is not None:
                lg = lg[:, perm]
            lg = torch.cat([lg[:, :objs + 1], torch.full_like(lg[:, objs + 1:], -1e10)], 1)
            scored = [lg[:, :objs + 1].contiguous()]
            label = [gt.view(1, *size)]
            loss = 0
            for fn, wgt in zip(engine.losses, engine.loss_weights):
                loss = loss + wgt * fn(scored, label, step)
            losses[t][b] = loss
            preds[t][b] = lg.detach().argmax(1)
            return torch.softmax(lg, 1) if use_prev_prob else preds[t][b].view(1, 1, *size).float()

        clip.add_reference_frame(frames[0, b:b + 1], ident(masks[0, b:b + 1]), frame_step=0)
        score(0)
        t = 1
        if enable_prev_frame:
            clip.add_reference_frame(frames[1, b:b + 1], ident(masks[1, b:b + 1]), frame_step=1)
            score(1)
            t = 2
        while t < T_:
            clip.match_propogate_one_frame(frames[t, b:b + 1])
            pred = score(t)
            if t < T_ - 1:
                clip.update_memory(ident(pred if use_prev_pred else masks[t, b:b + 1]))
            t += 1
    frame_loss = [torch.cat(l, 0) for l in losses]
    frame_mask = [torch.cat(m, 0) for m in preds]
    aux_loss = torch.cat(frame_loss[:n_aux], 0).mean(0)
    pred_loss = torch.cat(frame_loss[n_aux:], 0).mean(0)
    loss = aux_weight * aux_loss + pred_loss
    return loss, frame_mask, frame_loss, {'image': {}, 'scalar': {}}
