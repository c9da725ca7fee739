"""The training step's DIFFERENTIABLE forward (reference networks/engines/aot_engine.py:33-108 as
trainer.py:460-519 drives it: `loss.backward()` -> clip -> AdamW -> EMA).

The inference path runs fused kernels on packed weights and keeps no activations; a training step needs the autograd
graph.  Here the same networks are written once more over the few differentiable primitives of
`networks/layers/train_ops.py` (every compute node = one C-ABI kernel with a hand-written backward), directly on the
modules' `nn.Parameter`s, so that `loss.backward()` fills `.grad` exactly as it does for the reference:

    MobileNetV2 / ResNet / Swin trunks (FrozenBN folded on the fly; frozen stages record no graph)
    LongShortTermTransformerBlock / GatedPropagationModule and their attentions        transformer.py:312-367,582-665
    FPNSegmentationHead, the identity bank, the sine position embedding                fpn.py:34-58, aot.py:50-79
    the engine's frame recurrence: reference frame, optional second self-memorising frame, propagated frames with
    ground-truth / prediction / probability feedback (back-propagation through time comes from autograd)

Scope: every trunk the package has -- MobileNetV2, ResNet-50 / 101, Swin-B (AOT-T/S/B/L, DeAOT-T/S/B/L, R50-/R101-AOTL,
R50-/R101-DeAOTL, SwinB-AOTL, SwinB-DeAOTL; BASELINE config 5 trains R50-DeAOTL).

The SAMPLES OF A BATCH ARE LANES, as the object groups are on the inference path: every activation is token-major
[B * N, C] with sample b in rows [b * N, (b + 1) * N), and every kernel launch serves the whole batch (the convolution /
normalisation / resize primitives take B maps, the attention products B * heads matrices); only what differs per sample --
the identity permutation, the number of objects the loss looks at -- is sliced per sample.  Drop-path / Dropout2d draw per
sample, as the reference's do along its batch axis, and follow the modules' `training` flag (identities in eval mode,
which is how the gradient goldens were made)."""
import torch
import torch.nn.functional as F

from networks.layers import train_ops as T
from networks.layers.transformer import DualBranchGPM


# ---- small pieces ------------------------------------------------------------------------------------------------------
def _fold_bn(weight, bn):
    """conv weight [Cout, ...] and FrozenBatchNorm2d (constants) -> (weight * scale, shift)."""
    def make():
        scale = (bn.weight * (bn.running_var + bn.epsilon).rsqrt()).detach()
        shift = (bn.bias - bn.running_mean * scale).detach()
        w = weight * scale.view(-1, *([1] * (weight.dim() - 1)))
        w._aot_wkey = (weight.data_ptr(), tuple(weight.shape), 'folded')      # names the weight inside train_ops.weight_cache()
        return w, shift
    # inside a weight_cache() scope (one optimiser step) the folded weight is ONE graph node shared by every frame that uses it
    # (the grad mode is part of the key: a fold first formed under torch.no_grad() has no grad_fn, and handing it to a later
    #  grad-enabled use would silently leave the conv weight without a gradient -- ADVICE r4)
    return T._cached((weight.data_ptr(), id(bn), 'fold', torch.is_grad_enabled()), make)


def _drop_path(x, p, training, B):
    """DropPath (basic.py:129-148, one draw per sample of the batch): a sample's whole branch dropped with probability p."""
    if not training or not p:
        return x
    return (x.view(B, -1, x.shape[1]) * _keep_mask((B, 1, 1), 1. - p, x)).view(x.shape)


def _keep_mask(shape, keep, like):
    """Bernoulli(keep) / keep (two launches: the draw and the rescale)."""
    return torch.empty(shape, dtype=like.dtype, device=like.device).bernoulli_(keep).div_(keep)


def _dropout(x, p, training):
    """nn.Dropout (elementwise)."""
    if not training or not p:
        return x
    return x * _keep_mask(x.shape, 1. - p, x)


def _dropout2d(x, p, training, B):
    """nn.Dropout2d on B token-major maps [B * N, C]: whole channels of a sample dropped (basic.py:46,55)."""
    if not training or not p:
        return x
    return (x.view(B, -1, x.shape[1]) * _keep_mask((B, 1, x.shape[1]), 1. - p, x)).view(x.shape)


def _no_attn_dropout(m):
    if m.training and m.dropout_p:
        raise NotImplementedError('dropout on the attention weights (TRAIN_LSTT_LT_DROPOUT / ST_DROPOUT, 0 in every reference '
                                  'config) is not part of the differentiable forward')


def _maps_nhwc(maps, cpad):
    """[B, C, H, W] planar -> token-major [B * H * W, cpad] (channels >= C zero)."""
    if maps.shape[0] == 1:
        return T.to_nhwc(maps.float(), cpad)
    return torch.cat([T.to_nhwc(maps[b:b + 1].float(), cpad) for b in range(maps.shape[0])], 0)


def _cbr(x, seq, B, H, W):
    """ConvBNActivation (conv / depthwise conv + FrozenBN + ReLU6), mobilenetv2.py:30-46."""
    conv, bn = seq[0], seq[1]
    w, b = _fold_bn(conv.weight, bn)
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    if conv.groups == 1:
        y, OH, OW = T.conv2d(x, w, b, B, H, W, s, p, d)
    else:
        y, OH, OW = T.dwconv2d(x, w, B, H, W, s, p, d)
        y = y + b
    return T.act(y, 'relu6'), OH, OW


def mobilenetv2_features(enc, img):
    """img [B, 3, H, W] -> [(f4, h, w), (f8, h, w), (f16, h, w), (top, h, w)] token-major (mobilenetv2.py:219-224)."""
    B, _, H, W = img.shape
    x = _maps_nhwc(img, 4)
    x, h, w = _cbr(x, enc.features[0], B, H, W)
    feats = []
    for idx in range(1, 18):
        blk = enc.features[idx]
        y, hh, ww = x, h, w
        j = 0
        if blk.expand:
            y, hh, ww = _cbr(y, blk.conv[0], B, hh, ww)
            j = 1
        y, hh, ww = _cbr(y, blk.conv[j], B, hh, ww)
        wpl, bpl = _fold_bn(blk.conv[j + 1].weight, blk.conv[j + 2])
        y, hh, ww = T.conv2d(y, wpl, bpl, B, hh, ww)
        x = x + y if blk.use_res_connect else y
        h, w = hh, ww
        if idx in (3, 6, 13):
            feats.append((x, h, w))
    x, h, w = _cbr(x, enc.features[18], B, h, w)
    feats.append((x, h, w))
    return feats


def resnet_features(enc, img):
    """ResNet-50 / 101 trunk (encoders/resnet.py:140-157): img [B, 3, H, W] -> [(f4, h, w), (f8, h, w), (f16, h, w)] token-major.
    conv + FrozenBN folded per call; the stem's max pool has no backward kernel -- with the stem frozen
    (TRAIN_ENCODER_FREEZE_AT >= 1, every reference recipe) nothing asks for one."""
    B, _, H, W = img.shape
    x = _maps_nhwc(img, 4)
    w, b = _fold_bn(enc.conv1.weight, enc.bn1)
    x, h, wd = T.conv2d(x, w, b, B, H, W, 2, 3, 1)
    x = T.act(x, 'relu')
    pooled = [T.maxpool3x3s2(x[i * h * wd:(i + 1) * h * wd], h, wd) for i in range(B)]
    x, h, wd = (pooled[0][0] if B == 1 else torch.cat([q[0] for q in pooled], 0)), pooled[0][1], pooled[0][2]
    feats = []
    for layer in (enc.layer1, enc.layer2, enc.layer3):
        for blk in layer:
            s, d = blk.stride, blk.dilation
            w1, b1 = _fold_bn(blk.conv1.weight, blk.bn1)
            y = T.act(T.conv2d(x, w1, b1, B, h, wd)[0], 'relu')
            w2, b2 = _fold_bn(blk.conv2.weight, blk.bn2)
            y, oh, ow = T.conv2d(y, w2, b2, B, h, wd, s, d, d)
            y = T.act(y, 'relu')
            w3, b3 = _fold_bn(blk.conv3.weight, blk.bn3)
            y = T.conv2d(y, w3, b3, B, oh, ow)[0]
            if blk.downsample is not None:
                wds, bds = _fold_bn(blk.downsample[0].weight, blk.downsample[1])
                res = T.conv2d(x, wds, bds, B, h, wd, s, 0, 1)[0]                # (stride 2: im2col of a 1x1 window = every 2nd pixel)
            else:
                res = x
            x, h, wd = T.act(y + res, 'relu'), oh, ow
        feats.append((x, h, wd))
    return feats


def swin_features(enc, img):
    """Swin trunk, three stages (encoders/swin/swin_transformer.py:684-716): img [B, 3, H, W] -> [(f4, h, w), (f8, h, w),
    (f16, h, w)] token-major.  Window partition / cyclic shift / padding are index plumbing (views, roll, pad, permute); the
    arithmetic -- LayerNorm, the qkv / proj / MLP linears, QK^T + relative-position bias (+ shift mask) -> softmax -> PV per
    window and head, GELU -- runs on the differentiable primitives."""
    B, _, H, W = img.shape
    pe = enc.patch_embed
    ps = pe.patch_size
    if H % ps or W % ps:          # sides that are not multiples of the patch: zero-padded right / bottom (:501-509)
        img = F.pad(img.float(), (0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))
    x, h, w = T.conv2d(_maps_nhwc(img, 4), pe.proj.weight, pe.proj.bias, B, img.shape[2], img.shape[3], ps, 0, 1)
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
            y = T.layernorm(x, blk.norm1.weight, blk.norm1.bias).view(B, h, w, C)
            y = F.pad(y, (0, 0, 0, wp - w, 0, hp - h))
            if sh:
                y = torch.roll(y, shifts=(-sh, -sh), dims=(1, 2))
            win = y.view(B, hp // ws, ws, wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nw * Nt, C)
            qkv = T.linear(win, at.qkv.weight, at.qkv.bias).view(B * nw, Nt, 3, nh, C // nh).permute(2, 0, 3, 1, 4)   # [3, B nw, nh, 49, d]
            q, k, v = (t.reshape(B * nw * nh, Nt, C // nh) for t in (qkv[0], qkv[1], qkv[2]))
            att = T.matmul(q, k.transpose(1, 2), alpha=at.scale)                               # (q * scale) k^T, :181-182
            rpb = at.relative_position_bias_table[at.relative_position_index.view(-1)].view(Nt, Nt, nh).permute(2, 0, 1)
            att = att.view(B, nw, nh, Nt, Nt) + rpb.view(1, 1, nh, Nt, Nt)
            if sh:
                att = att + amask.view(1, nw, 1, Nt, Nt)
            o = T.matmul(T.softmax_rows(att.reshape(B * nw * nh, Nt, Nt)), v)                  # [B nw nh, 49, d]
            o = o.view(B * nw, nh, Nt, C // nh).permute(0, 2, 1, 3).reshape(B * nw * Nt, C)
            o = T.linear(o, at.proj.weight, at.proj.bias)
            o = o.view(B, hp // ws, wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, hp, wp, C)
            if sh:
                o = torch.roll(o, shifts=(sh, sh), dims=(1, 2))
            x = x + _drop_path(o[:, :h, :w].reshape(B * h * w, C), blk.drop_path_p, blk.training, B)
            f = T.act(T.linear(T.layernorm(x, blk.norm2.weight, blk.norm2.bias), blk.mlp.fc1.weight, blk.mlp.fc1.bias), 'gelu')
            x = x + _drop_path(T.linear(f, blk.mlp.fc2.weight, blk.mlp.fc2.bias), blk.drop_path_p, blk.training, B)
        if li in enc.out_indices:
            nm = getattr(enc, 'norm%d' % li)
            feats.append((T.layernorm(x, nm.weight, nm.bias), h, w))
        if layer.downsample is not None:                                                       # PatchMerging :338-359
            ds = layer.downsample
            y = x.view(B, h, w, C)
            if h % 2 or w % 2:
                y = F.pad(y, (0, 0, 0, w % 2, 0, h % 2))
            y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
            h, w = (h + 1) // 2, (w + 1) // 2
            x = T.linear(T.layernorm(y.reshape(B * h * w, 4 * C), ds.norm.weight, ds.norm.bias), ds.reduction.weight)
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


def _heads(t, H, B):
    """[B * n, H * d] -> [B * H, n, d] view (a matrix per sample and head)."""
    n, c = t.shape[0] // B, t.shape[1]
    return t.view(B, n, H, c // H).permute(0, 2, 1, 3).reshape(B * H, n, c // H)


def _unheads(o, H, B):
    """[B * H, n, d] -> [B * n, H * d]."""
    n, d = o.shape[1], o.shape[2]
    return o.view(B, H, n, d).permute(0, 2, 1, 3).reshape(B * n, H * d)


def _per_sample(p, B):
    """a per-head parameter [H, a, b] as the [B * H, a, b] operand of a batched product (the gradient sums over the samples)."""
    return p if B == 1 else p.unsqueeze(0).expand(B, *p.shape).reshape(B * p.shape[0], *p.shape[1:])


def _attention(q, k, v, H, scale, B):
    """softmax((q / scale) k^T) v per sample and head (attention.py:82-117): q [B * N, C], k / v [B * T, C] -> [B * N, C]."""
    s = T.matmul(_heads(q / scale, H, B), _heads(k, H, B).transpose(1, 2))       # [B H, N, T]
    return _unheads(T.matmul(T.softmax_rows(s), _heads(v, H, B)), H, B)


def _local_attention(m, q, k, v, size_2d, B):
    """MultiheadLocalAttentionV2 core (attention.py:308-376): q, k, v [B * N, C] of the current / previous frame."""
    h, w = size_2d
    H, d, R = m.num_head, m.hidden_dim, m.max_dis
    W2 = m.window_size * m.window_size
    qh = _heads(q, H, B)                                                        # UNSCALED q for the relative-position term (:327)
    rel = T.matmul(qh, _per_sample(m.relative_emb_k.weight.view(H, W2, d).transpose(1, 2), B))
    rel = (rel.view(B, H, -1, W2) + m.relative_emb_k.bias.view(1, H, 1, W2)).view(B * H, -1, W2)
    dense = T.matmul(_heads(q / m.T, H, B), _heads(k, H, B).transpose(1, 2))     # [B H, N, N]
    s = T.window_gather(dense, h, w, R, float('-inf')) + rel                   # outside the image: -inf (the reference's -1e8)
    a = T.softmax_rows(s)                                                      # [B H, N, 225]
    o = T.matmul(T.window_scatter(a, h, w, R, 0.0), _heads(v, H, B)) + T.matmul(a, _per_sample(m.relative_emb_v.transpose(1, 2), B))
    return _unheads(o, H, B)


def _gated_tail(m, agg, u, size_2d, B):
    """(agg * u) -> depthwise 5x5 -> projection (attention.py:707-710, 855-860)."""
    h, w = size_2d
    x = agg * u
    x, _, _ = T.dwconv2d(x, m.dw_conv.conv.weight, B, h, w, 1, 2, 1)
    x = _dropout2d(x, m.dw_conv.dropout_p, m.training, B)
    return T.linear(x, m.projection.weight, m.projection.bias)


def _gated_global(m, q, k, v, u, size_2d, B):
    """GatedPropagation core (attention.py:672-710), single head: q [B * N, 128], k [B * T, 128], v [B * T, E], u [B * N, E]."""
    s = T.matmul((q / m.T).view(B, -1, q.shape[1]), k.view(B, -1, k.shape[1]).transpose(1, 2))       # [B, N, T]
    agg = T.matmul(T.softmax_rows(s), v.view(B, -1, v.shape[1])).reshape(-1, v.shape[1])
    return _gated_tail(m, agg, u, size_2d, B)


def _gated_local(m, q, k, v, u, size_2d, B):
    """LocalGatedPropagation core (attention.py:814-860), single head."""
    h, w = size_2d
    R = m.max_dis
    W2 = m.window_size * m.window_size
    rel = T.linear(q, m.relative_emb_k.weight.view(W2, -1), m.relative_emb_k.bias).view(B, -1, W2)      # unscaled q (:814)
    dense = T.matmul((q / m.T).view(B, -1, q.shape[1]), k.view(B, -1, k.shape[1]).transpose(1, 2))     # [B, N, N]
    a = T.softmax_rows(T.window_gather(dense, h, w, R, float('-inf')) + rel)
    agg = T.matmul(T.window_scatter(a, h, w, R, 0.0), v.view(B, -1, v.shape[1])).reshape(-1, v.shape[1])
    return _gated_tail(m, agg, u, size_2d, B)


# ---- LSTT block (AOT) --------------------------------------------------------------------------------------------------
def lstt_block(blk, x, long_mem, short_mem, id_emb, pos, size_2d, B):
    """LongShortTermTransformerBlock.forward (transformer.py:312-362) on token-major x [B * N, C].  long_mem / short_mem = (K, V);
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
    o = _attention(q, k, v, sa.num_head, sa.T, B)
    x = x + _drop_path(T.linear(o, sa.projection.weight, sa.projection.bias), dp, tr, B)
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
    a_lt = T.linear(_attention(qc, kg, vg, lt.num_head, lt.T, B), lt.projection.weight, lt.projection.bias)
    st = blk.short_term_attn
    a_st = T.linear(_local_attention(st, qc, kl, vl, size_2d, B), st.projection.weight, st.projection.bias)
    x = x + (_drop_path(a_lt + a_st, dp, tr, B) if blk.droppath_lst else _dropout(a_lt + a_st, blk.lst_dropout_p, tr))     # :350-353
    x3 = T.layernorm(x, blk.norm3.weight, blk.norm3.bias)
    f = T.linear(x3, blk.linear1.weight, blk.linear1.bias)
    f = T.act(T.groupnorm(f, blk.activation.gn.weight, blk.activation.gn.bias, blk.activation.gn.num_groups, B), 'gelu')
    f, _, _ = T.dwconv2d(f, blk.activation.conv.weight, B, h, w, 1, 2, 1)
    x = x + _drop_path(T.linear(f, blk.linear2.weight, blk.linear2.bias), dp, tr, B)
    return x, [kc, vc], [kg, vg]


def fuse_kv(blk, k, v, id_emb):
    """fuse_key_value_id (transformer.py:364-367): K unchanged, V <- linear_V(V + id_emb)."""
    return k, T.linear(v + id_emb, blk.linear_V.weight, blk.linear_V.bias)


# ---- GPM block (DeAOT) -------------------------------------------------------------------------------------------------
def gpm_fuse_id(blk, idv, id_emb):
    """fuse_key_value_id (transformer.py:659-665): ID_V = silu(linear_ID_V([ID_V,] id_emb))."""
    z = id_emb if idv is None else torch.cat([idv, id_emb], 1)
    return T.act(T.linear(z, blk.linear_ID_V.weight, blk.linear_ID_V.bias), 'silu')


def gpm_block(blk, x, x_id, long_mem, short_mem, id_emb, size_2d, B):
    """GatedPropagationModule.forward (transformer.py:582-657).  Memories = (K, V, ID_V).  Returns (x, x_id, [K, V, ID_V in],
    [K_g, V_g, ID_V_g])."""
    D, da = blk.d_model, blk.d_att * blk.att_nhead
    for m in (blk.self_attn, blk.long_term_attn, blk.short_term_attn):
        _no_attn_dropout(m)
        if m.num_head != 1:
            raise NotImplementedError('the gated propagation of the reference configs is single-head (default_deaot.py)')
    dp, tr = blk.droppath_p, blk.training
    x1 = T.layernorm(x, blk.norm1.weight, blk.norm1.bias)
    qv = T.linear(x1, blk.linear_QV.weight, blk.linear_QV.bias)
    qc, vraw = qv.split([da, qv.shape[1] - da], 1)        # (one split node: its backward is ONE concatenation, not two zero-fills + copies + an add)
    vc = T.act(vraw, 'silu')
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
    lt = _gated_global(blk.long_term_attn, qc, kg, torch.cat([vg, idvg], 1), u, size_2d, B)
    st = _gated_local(blk.short_term_attn, qc, kl, torch.cat([vl, idvl], 1), u, size_2d, B)
    ya, yb = (lt + st).split([D, lt.shape[1] - D], 1)
    if blk.droppath_lst:                                                       # :633-638: the two halves draw separately
        ya, yb = _drop_path(ya, dp, tr, B), _drop_path(yb, dp, tr, B)
    else:
        ya, yb = _dropout(ya, blk.lst_dropout_p, tr), _dropout(yb, blk.lst_dropout_p, tr)
    x = x + ya
    x_id = yb if x_id is None else x_id + yb
    z1 = T.layernorm(x, blk.norm2.weight, blk.norm2.bias)
    z2 = T.layernorm(x_id, blk.id_norm2.weight, blk.id_norm2.bias)
    z = torch.cat([z1, z2], 1)
    sa = blk.self_attn
    qk = T.linear(z, sa.linear_QK.weight, sa.linear_QK.bias)
    sv = T.act(torch.cat([T.linear(z1, sa.linear_V1.weight, sa.linear_V1.bias), T.linear(z2, sa.linear_V2.weight, sa.linear_V2.bias)], 1), 'silu')
    su = T.act(torch.cat([T.linear(z1, sa.linear_U1.weight, sa.linear_U1.bias), T.linear(z2, sa.linear_U2.weight, sa.linear_U2.bias)], 1), 'silu')
    ya, yb = _gated_global(sa, qk, qk, sv, su, size_2d, B).split([D, D], 1)
    return x + _drop_path(ya, dp, tr, B), x_id + _drop_path(yb, dp, tr, B), [kc, vc, idvc], [kg, vg, idvg]


# ---- decoder -----------------------------------------------------------------------------------------------------------
def fpn_decoder(dec, x_in, shortcuts, size_2d, B):
    """FPNSegmentationHead.forward (fpn.py:34-58): x_in [B * N16, in_dim], shortcuts = [(f4, h, w), (f8, h, w), (f16, h, w)] ->
    logits [B * h4 * w4, out_dim], h4, w4."""
    (s4, h4, w4), (s8, h8, w8), (s16, h16, w16) = shortcuts

    def cgn(m, x, hh, ww):      # ConvGN (basic.py:75-85) + ReLU
        k = m.conv.kernel_size[0]
        y, _, _ = T.conv2d(x, m.conv.weight, m.conv.bias, B, hh, ww, 1, k // 2, 1)
        return T.act(T.groupnorm(y, m.gn.weight, m.gn.bias, m.gn.num_groups, B), 'relu')

    def adapter(m, s, hh, ww):
        return T.conv2d(s, m.weight, m.bias, B, hh, ww)[0]
    x = cgn(dec.conv_in, x_in, h16, w16) + adapter(dec.adapter_16x, s16, h16, w16)
    x = cgn(dec.conv_16x, x, h16, w16)
    x = T.bilinear(x, B, h16, w16, h8, w8, dec.align_corners) + adapter(dec.adapter_8x, s8, h8, w8)
    x = cgn(dec.conv_8x, x, h8, w8)
    x = T.bilinear(x, B, h8, w8, h4, w4, dec.align_corners) + adapter(dec.adapter_4x, s4, h4, w4)
    x = cgn(dec.conv_4x, x, h4, w4)
    return T.conv2d(x, dec.conv_out.weight, dec.conv_out.bias, B, h4, w4)[0], h4, w4


# ---- the clips of a batch, frame by frame ----------------------------------------------------------------------------------
class ClipGraph:
    """The frame recurrence of B clips at once (aot_engine.py:188-354 under autograd): memories are graph tensors, the
    long-term bank of a sample is the concatenation of its memorised frames."""

    def __init__(self, model, batch, long_term_mem_gap=9999, freeze_id=False):
        self.m = model
        self.B = int(batch)
        self.freeze_id = bool(freeze_id)      # aot_engine.py:46,176-177: with prediction feedback the identity bank gets no gradient
        self.deaot = isinstance(model.LSTT, DualBranchGPM)
        self.gap = long_term_mem_gap
        self.frame_step = 0
        self.last_mem_step = -1
        self.long = None           # per layer: list over the memorised frames of per-frame memories ([B * N, .] each)
        self.short = None
        self.curr = None
        self.pos = None
        self.size_2d = None
        self.feats = None
        self.dec_in = None
        self._pre = None           # encode_all(): per frame ([f4, f8, f16], projected 16x map, h, w)

    def encode_all(self, frames_all, n_frames):
        """The encoder over ALL frames of the step as one batch of n_frames * B lanes (time-major), like the reference's
        offline_encoder (aot_engine.py:147-166): the trunk does not depend on the memory, and one pass over ten maps costs a fifth
        of the launches of five passes over two.  _encode(t) then hands out frame t's rows (unbind: one stack in backward)."""
        B = self.B
        feats, (top, h, w) = encoder_features(self.m.encoder, frames_all)
        proj = self.m.encoder_projector
        x16 = T.conv2d(top, proj.weight, proj.bias, n_frames * B, h, w)[0]
        per = [[(ft, hh, ww) for ft in f.view(n_frames, B * hh * ww, f.shape[1]).unbind(0)] for f, hh, ww in feats]
        x16s = x16.view(n_frames, B * h * w, x16.shape[1]).unbind(0)
        self._pre = [([per[k][t] for k in range(len(per))], x16s[t], h, w) for t in range(n_frames)]

    def _encode(self, img, t=None):
        if self._pre is not None and t is not None:
            self.feats, x16, h, w = self._pre[t]
        else:
            self.feats, (top, h, w) = encoder_features(self.m.encoder, img)
            proj = self.m.encoder_projector
            x16 = T.conv2d(top, proj.weight, proj.bias, self.B, h, w)[0]
        if self.size_2d is None:
            self.size_2d = (h, w)
            with torch.no_grad():
                pe = self.m.get_pos_emb(torch.zeros(1, x16.shape[1], h, w, device=img.device))
            self.pos = pe[0].permute(1, 2, 0).reshape(h * w, -1).repeat(self.B, 1).contiguous()
        return x16

    def id_emb(self, one_hot):
        """one-hot or probability maps [B, L, H, W] -> identity embedding [B * N, C] (aot.py:76-79, deaot.py:51-55)."""
        bank = self.m.patch_wise_id_bank
        H, W = one_hot.shape[-2:]
        x = _maps_nhwc(one_hot, (one_hot.shape[1] + 3) // 4 * 4)
        e, _, _ = T.conv2d(x, bank.weight, bank.bias, self.B, H, W, bank.stride[0], bank.padding[0], 1)
        if self.deaot:
            e = T.layernorm(e, self.m.id_norm.weight, self.m.id_norm.bias)
        e = _dropout(e, self.m.id_dropout_p, self.m.training)
        # assign_identity (aot_engine.py:176-177): `if self.training and self.freeze_id: id_emb = id_emb.detach()` -- no gradient
        # into patch_wise_id_bank / id_norm, and none through a probability feedback into the earlier frames
        return e.detach() if self.freeze_id else e

    def _bank(self, i):
        """The long-term memory of layer i: per component, the samples' memorised frames side by side ([B * T, .])."""
        B = self.B
        return [torch.cat([t.view(B, -1, t.shape[1]) for t in comp], 1).reshape(-1, comp[0].shape[1]) for comp in zip(*self.long[i])]

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
            long_m = None if ref else self._bank(i)
            short_m = None if ref else self.short[i]
            if self.deaot:
                x, x_id, c, g = gpm_block(blk, x, x_id, long_m, short_m, id_emb, self.size_2d, self.B)
                outs.append(torch.cat([x, x_id], 1))
            else:
                x, c, g = lstt_block(blk, x, long_m, short_m, id_emb, self.pos, self.size_2d, self.B)
                outs.append(x)
            curr.append(c)
            new_long.append(g)

        def norm(n, t):
            if self.deaot:
                return T.groupnorm(t, n.gn.weight, n.gn.bias, n.gn.num_groups, self.B)
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

    def add_reference_frame(self, img, one_hot, frame_step=None, t=None):
        """aot_engine.py:188-251 (also set_prev_frame, :253-289): the frame memorises its own mask."""
        if frame_step is not None:
            self.frame_step = frame_step
        x16 = self._encode(img, t)
        mems = self._lstt(x16, self.id_emb(one_hot))
        if self.long is None:
            self.long = [[m] for m in mems]
        else:
            for bank, m in zip(self.long, mems):
                bank.append(m)
        self.last_mem_step = self.frame_step
        self.short = mems

    def match_propogate_one_frame(self, img, t=None):
        self.frame_step += 1
        self._lstt(self._encode(img, t), None)

    def decode_logits(self, out_size):
        """decode_current_logits (aot_engine.py:356-380): stride-4 logits -> output size; token-major [B * H * W, L]."""
        logits, h4, w4 = fpn_decoder(self.m.decoder, self.dec_in, self.feats, self.size_2d, self.B)
        L = logits.shape[1]
        lp = F.pad(logits, (0, (L + 3) // 4 * 4 - L))
        return T.bilinear(lp, self.B, h4, w4, out_size[0], out_size[1], self.m.cfg.MODEL_ALIGN_CORNERS)[:, :L]

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
    size = tuple(masks.shape[-2:])
    HW = size[0] * size[1]
    losses = [[None] * bs for _ in range(T_)]
    preds = [[None] * bs for _ in range(T_)]
    if getattr(engine, 'short_term_mem_skip', 1) != 1:
        raise NotImplementedError('training_forward keeps one short-term frame (short_term_mem_skip = 1, every reference recipe); '
                                  'aot_engine.py:329-332 would read short_term_memories_list[-skip:][0]')
    clip = ClipGraph(model, bs, engine.long_term_mem_gap, freeze_id=bool(engine.training and use_prev_pred))
    objs = [int(n) for n in obj_nums]
    # identity o of sample b is moved to channel perm[b][o] (trainer.py:457; reversed on the logits, aot_engine.py:364-367)
    perms = engine.id_shuffle if engine.enable_id_shuffle else [None] * bs
    invs = [None if p is None else torch.argsort(p) for p in perms]

    def ident(maps):
        """what assign_identity sees (aot_engine.py:168-179): per sample the (shuffled) one-hot / probability map -> [bs, L, H, W]"""
        out = []
        for b, m in enumerate(maps):
            oh = m if m.shape[1] == L else one_hot(m, model.max_obj_num)
            out.append(oh if invs[b] is None else oh[:, invs[b]])
        return torch.cat(out, 0)

    def score(t):
        """generate_loss_mask (aot_engine.py:398-430) of frame t: per sample, shuffled identities back in place, the channels of
        unused identities at -1e10 (constants: no gradient), loss on the first objs + 1 channels, prediction / probabilities."""
        lg_all = clip.decode_logits(size)
        feedback = []
        for b in range(bs):
            gt = masks[t, b:b + 1]
            lg = lg_all[b * HW:(b + 1) * HW]
            if perms[b] is not None:       # channel t <- the channel identity t was moved to (aot_engine.py:364-367): a column gather
                lg = T.permute_cols(lg, perms[b])
            lg = torch.cat([lg[:, :objs[b] + 1], lg.new_full((HW, L - objs[b] - 1), -1e10)], 1)
            scored = [T.to_nchw(lg[:, :objs[b] + 1], *size)]
            label = [gt.view(1, *size)]
            loss = 0
            for fn, wgt in zip(engine.losses, engine.loss_weights):
                loss = loss + wgt * fn(scored, label, step)
            losses[t][b] = loss
            preds[t][b] = lg.detach().argmax(1).view(1, *size)
            feedback.append(T.to_nchw(T.softmax_rows(lg), *size) if use_prev_prob else preds[t][b].view(1, 1, *size).float())
        return feedback

    truth = lambda t: [masks[t, b:b + 1] for b in range(bs)]
    # the auxiliary decodes record no graph once their weight has faded to 0 (aot_engine.py:55-58,66-69: `grad_state`): no wasted
    # decoder backward, and a non-finite auxiliary loss cannot reach the gradients through 0 * nan
    aux_grad = torch.no_grad if aux_weight == 0 else torch.enable_grad
    clip.encode_all(all_frames, T_)
    clip.add_reference_frame(frames[0], ident(truth(0)), frame_step=0, t=0)
    with aux_grad():
        score(0)
    t = 1
    if enable_prev_frame:
        clip.add_reference_frame(frames[1], ident(truth(1)), frame_step=1, t=1)
        with aux_grad():
            score(1)
        t = 2
    while t < T_:
        clip.match_propogate_one_frame(frames[t], t)
        pred = score(t)
        if t < T_ - 1:
            clip.update_memory(ident(pred if use_prev_pred else truth(t)))
        t += 1
    frame_loss = [torch.cat(l, 0) for l in losses]
    frame_mask = [torch.cat(m, 0) for m in preds]
    aux_loss = torch.cat(frame_loss[:n_aux], 0).mean(0)
    pred_loss = torch.cat(frame_loss[n_aux:], 0).mean(0)
    loss = aux_weight * aux_loss + pred_loss
    return loss, frame_mask, frame_loss, {'image': {}, 'scalar': {}}
