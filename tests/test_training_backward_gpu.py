"""The model's backward pass on the device: each differentiable primitive of networks/layers/train_ops.py --
forward value and every gradient -- against torch's own autograd of the same op in fp64 (tests/train_stand_ins.py), then the
whole training step (AOTEngine.forward with autograd on -> loss.backward() -> clip -> AdamW) against the REAL reference's
gradients and updated parameters (tests/golden/train_grads.npz).  Run on the MI355X box."""
import os

import numpy as np
import pytest
import torch

import train_stand_ins as S
from common import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import aot_hip
    aot_hip.load()
    from networks.layers import train_ops
    return train_ops


def _check(name, ours, ref, inputs, tol=2e-5):
    """ours / ref: callables on the list of leaf tensors (fp32 device / fp64 device copies) returning one tensor.  Compares the
    value and, under one random cotangent, the gradient of every leaf that requires one."""
    a = [t.detach().clone().requires_grad_(t.requires_grad) for t in inputs]
    b = [t.detach().double().clone().requires_grad_(t.requires_grad) for t in inputs]
    ya, yb = ours(a), ref(b)
    assert ya.shape == yb.shape, '%s: shape %s vs %s' % (name, tuple(ya.shape), tuple(yb.shape))
    fin = torch.isfinite(yb)
    assert torch.equal(torch.isfinite(ya), fin)
    scale = float(yb.detach()[fin].abs().max()) + 1e-30
    err = float((ya.detach().double() - yb.detach())[fin].abs().max()) / scale
    assert err <= tol, '%s: forward differs by %.2e of the largest entry' % (name, err)
    ct = torch.randn(yb.shape, dtype=torch.float64, device=yb.device, generator=torch.Generator(device=yb.device).manual_seed(7))
    ct = torch.where(fin, ct, torch.zeros_like(ct))
    ya.backward(ct.float())
    yb.backward(ct)
    worst = err
    for i, (p, q) in enumerate(zip(a, b)):
        if not p.requires_grad:
            continue
        assert p.grad is not None, '%s: no gradient for input %d' % (name, i)
        s = float(q.grad.abs().max()) + 1e-30
        e = float((p.grad.double() - q.grad).abs().max()) / s
        assert e <= tol, '%s: gradient of input %d differs by %.2e of its largest entry' % (name, i, e)
        worst = max(worst, e)
    return worst


def _r(*shape, grad=True, seed=0, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device='cuda', generator=g) * scale).requires_grad_(grad)


def test_matmul_on_strided_views(T):
    """aot_matmul_strided_f32 forward and both operand gradients + the bias gradient: contiguous, transposed, head-split
    (permuted) and broadcast (stride 0) operands, alpha != 1 -- the shapes nn.Linear, QK^T, PV and the relative-position
    products take in models/train_forward.py."""
    x, w, b = _r(437, 256), _r(96, 256, seed=1), _r(96, seed=2)
    _check('linear', lambda t: T.linear(*t), lambda t: S.linear(*t), [x, w, b])
    q, k = _r(437, 256, seed=3), _r(1311, 256, seed=4)
    heads = lambda t: t.view(t.shape[0], 8, 32).permute(1, 0, 2)
    _check('qk^t', lambda t: T.matmul(heads(t[0]), heads(t[1]).transpose(1, 2), alpha=0.25),
           lambda t: S.matmul(heads(t[0]), heads(t[1]).transpose(1, 2), alpha=0.25), [q, k])
    p, v = _r(8, 437, 225, seed=5), _r(8, 32, 225, seed=6)
    _check('attn x rel_v', lambda t: T.matmul(t[0], t[1].transpose(1, 2)), lambda t: S.matmul(t[0], t[1].transpose(1, 2)), [p, v])
    a1, b1 = _r(1, 33, 70, seed=7), _r(1, 70, 5, seed=8)
    _check('small odd', lambda t: T.matmul(t[0], t[1]), lambda t: S.matmul(t[0], t[1]), [a1, b1])


def test_copy2d_pad_makes_any_view_dense(T):
    """aot_copy2d_pad_f32 (train_ops._dense): row-major slices, transposed views, broadcasts and ragged extents become the dense
    zero-padded matrix the GEMM kernels take, bit for bit; a matrix that already is one is passed through untouched."""
    g = torch.Generator(device='cuda').manual_seed(5)
    base = torch.randn(301, 517, device='cuda', generator=g)
    views = {'row-major': base, 'column slice': base[:, 13:400], 'row slice': base[7:300:3], 'transposed': base.t(),
             'transposed slice': base[5:205, 11:76].t(), 'broadcast row': base[3:4].expand(77, 517), 'scalar': base[2:3, 4:5].expand(40, 90)}
    for name, v in views.items():
        R, C = v.shape
        for rpad, cpad in ((R, C), (R + 19, C), (R, -(-C // 32) * 32), (-(-R // 64) * 64, -(-C // 32) * 32 + 32)):
            d = T._dense(v, rpad, cpad)
            ref = torch.zeros(rpad, cpad, device='cuda')
            ref[:R, :C] = v
            assert d.shape == (rpad, cpad) and d.is_contiguous() and torch.equal(d, ref), '%s -> [%d, %d]' % (name, rpad, cpad)
    assert T._dense(base, 301, 517).data_ptr() == base.data_ptr()


def test_matmul_few_large_matrices_on_the_tile_kernels(T):
    """The gated propagation's products, one matrix per sample (QK^T with the keys as a transposed view, PV, a reduction length that
    is no multiple of 32) and their gradients: the LDS-direct GEMM path of matmul()."""
    B, N, Tk = 2, 900, 2 * 900 + 37
    q, k = _r(B, N, 128), _r(B, Tk, 128, seed=1)
    _check('gated qk^t', lambda t: T.matmul(t[0], t[1].transpose(1, 2)), lambda t: S.matmul(t[0], t[1].transpose(1, 2)), [q, k], tol=3e-5)
    p, v = _r(B, N, Tk, seed=2, scale=0.05), _r(B, Tk, 1024, seed=3)
    _check('gated pv', lambda t: T.matmul(t[0], t[1]), lambda t: S.matmul(t[0], t[1]), [p, v], tol=3e-5)
    x, w, b = _r(1, 1674, 3468, seed=4, scale=0.1), _r(256, 3468, seed=5, scale=0.1), _r(256, seed=6)
    _check('id bank as a matmul', lambda t: T.matmul(t[0], t[1].t().unsqueeze(0), t[2]), lambda t: S.matmul(t[0], t[1].t().unsqueeze(0), t[2]),
           [x, w, b], tol=3e-5)
    x, w, b = _r(27378, 128, seed=7), _r(11, 128, seed=8), _r(11, seed=9)       # the decoder's conv_out: narrow output, its weight
    _check('narrow output', lambda t: T.linear(*t), lambda t: S.linear(*t), [x, w, b], tol=3e-5)      # gradient a long reduction


@pytest.mark.parametrize('M,K,N', [(437, 256, 96), (5000, 1152, 128), (3000, 256, 1024), (1674, 1024, 256), (2100, 64, 48),
                                   (437, 100, 96), (40, 256, 96), (2000, 256, 11)])
def test_linear_on_the_tile_kernels(T, M, K, N):
    """nn.Linear: forward, dgrad and wgrad on the LDS-direct fp32 GEMM kernels of the inference path (split-K wgrad with the row
    count zero-padded to the granule; column-sum bias gradient) where the shapes allow, the strided general kernel elsewhere
    (K not a multiple of 32, few rows, a narrow output)."""
    x, w, b = _r(M, K), _r(N, K, seed=1, scale=0.2), _r(N, seed=2)
    _check('linear %dx%dx%d' % (M, K, N), lambda t: T.linear(*t), lambda t: S.linear(*t), [x, w, b], tol=3e-5)
    xs = _r(M, K + 32, seed=3)                      # a column slice of a wider buffer as the input
    _check('linear on a slice', lambda t: T.linear(t[0][:, 16:16 + K], t[1], None), lambda t: S.linear(t[0][:, 16:16 + K], t[1], None),
           [xs, w], tol=3e-5)


@pytest.mark.parametrize('geom', [
    # (H, W, Cin of the map, Cin of the weight, Cout, K, stride, pad, dil)
    (33, 41, 4, 3, 32, 3, 2, 1, 1),         # stem: image padded 3 -> 4 channels
    (17, 21, 64, 64, 48, 3, 1, 1, 1),       # decoder 3x3
    (129, 161, 12, 11, 256, 17, 16, 8, 1),  # identity bank, align-corners geometry
    (128, 160, 12, 11, 256, 16, 16, 0, 1),  # identity bank, the other geometry
    (9, 11, 96, 96, 40, 1, 1, 0, 1),        # 1x1
])
def test_conv2d_im2col(T, geom):
    H, W, cmap, cw, cout, K, s, p, d = geom
    x = _r(H * W, cmap)
    if cmap != cw:
        x = x.detach().clone()
        x[:, cw:] = 0
        x.requires_grad_(True)
    w, b = _r(cout, cw, K, K, seed=1, scale=0.1), _r(cout, seed=2)
    _check('conv2d', lambda t: T.conv2d(t[0], t[1], t[2], 1, H, W, s, p, d)[0], lambda t: S.conv2d(t[0], t[1], t[2], 1, H, W, s, p, d)[0],
           [x, w, b])


@pytest.mark.parametrize('geom', [(17, 21, 64, 3, 1, 1, 1), (33, 41, 96, 3, 2, 1, 1), (9, 11, 192, 3, 1, 2, 2), (30, 38, 1024, 5, 1, 2, 1)])
def test_dwconv2d(T, geom):
    H, W, C, K, s, p, d = geom
    x, w = _r(H * W, C), _r(C, 1, K, K, seed=1)
    _check('dwconv2d', lambda t: T.dwconv2d(t[0], t[1], 1, H, W, s, p, d)[0], lambda t: S.dwconv2d(t[0], t[1], 1, H, W, s, p, d)[0], [x, w])


@pytest.mark.parametrize('kind', ['relu', 'relu6', 'gelu', 'silu'])
def test_activations(T, kind):
    x = _r(301, 77, scale=4.0)
    _check(kind, lambda t: T.act(t[0], kind), lambda t: S.act(t[0], kind), [x])


def test_layernorm_and_groupnorm(T):
    x, g, b = _r(437, 256, scale=3.0), _r(256, seed=1), _r(256, seed=2)
    _check('layernorm', lambda t: T.layernorm(*t), lambda t: S.layernorm(*t), [x, g, b])
    for C, G, rows in ((1024, 32, 437), (256, 8, 1700), (512, 2, 437), (128, 8, 6000)):
        x, g, b = _r(rows, C, scale=2.0, seed=G), _r(C, seed=1), _r(C, seed=2)
        _check('groupnorm %d/%d' % (C, G), lambda t: T.groupnorm(t[0], t[1], t[2], G), lambda t: S.groupnorm(t[0], t[1], t[2], G), [x, g, b],
               tol=5e-5)


def test_softmax_rows_with_masked_entries(T):
    x = _r(8, 437, 225, scale=3.0).detach()
    x[:, :, ::7] = float('-inf')
    x[:, 5, :] = x[:, 5, :].clamp(max=-1.0)
    x.requires_grad_(True)
    _check('softmax_rows', lambda t: T.softmax_rows(t[0]), lambda t: S.softmax_rows(t[0]), [x])
    y = _r(1, 437, 1311, scale=5.0)
    _check('softmax_rows long', lambda t: T.softmax_rows(t[0]), lambda t: S.softmax_rows(t[0]), [y])


@pytest.mark.parametrize('geom', [(9, 11, 17, 21, 256, True), (17, 21, 33, 41, 128, True), (8, 10, 16, 20, 128, False),
                                  (33, 41, 129, 161, 12, True), (32, 40, 128, 160, 12, False)])
def test_bilinear(T, geom):
    IH, IW, OH, OW, C, align = geom
    x = _r(IH * IW, C)
    _check('bilinear', lambda t: T.bilinear(t[0], 1, IH, IW, OH, OW, align), lambda t: S.bilinear(t[0], 1, IH, IW, OH, OW, align), [x])


def test_window_gather_and_scatter(T):
    h, w, R = 9, 11, 7
    d = _r(8, h * w, h * w)
    _check('window_gather', lambda t: T.window_gather(t[0], h, w, R, float('-inf')), lambda t: S.window_gather(t[0], h, w, R, float('-inf')), [d])
    a = _r(8, h * w, 225, seed=1)
    _check('window_scatter', lambda t: T.window_scatter(t[0], h, w, R, 0.0), lambda t: S.window_scatter(t[0], h, w, R, 0.0), [a])
    h, w = 19, 23                                     # a map larger than the window in both directions
    d = _r(1, h * w, h * w, seed=2)
    _check('window_gather big', lambda t: T.window_gather(t[0], h, w, R, 0.0), lambda t: S.window_gather(t[0], h, w, R, 0.0), [d])


def test_batched_maps(T):
    """B = 2 maps per launch (the samples of a training batch are lanes): convolution through im2col / col2im, depthwise
    convolution, GroupNorm (statistics per sample and group), bilinear resize -- value and every gradient."""
    B, H, W = 2, 17, 21
    x, w, b = _r(B * H * W, 64), _r(48, 64, 3, 3, seed=1, scale=0.1), _r(48, seed=2)
    _check('conv2d B=2', lambda t: T.conv2d(t[0], t[1], t[2], B, H, W, 1, 1, 1)[0], lambda t: S.conv2d(t[0], t[1], t[2], B, H, W, 1, 1, 1)[0],
           [x, w, b])
    x, w = _r(B * H * W, 96), _r(96, 1, 5, 5, seed=3)
    _check('dwconv2d B=2', lambda t: T.dwconv2d(t[0], t[1], B, H, W, 1, 2, 1)[0], lambda t: S.dwconv2d(t[0], t[1], B, H, W, 1, 2, 1)[0], [x, w])
    x, g, bt = _r(B * 437, 256, scale=2.0, seed=4), _r(256, seed=5), _r(256, seed=6)
    x = (x.detach() + torch.arange(B, device='cuda').repeat_interleave(437).view(-1, 1) * 3.0).requires_grad_(True)   # distinct statistics
    _check('groupnorm B=2', lambda t: T.groupnorm(t[0], t[1], t[2], 8, B), lambda t: S.groupnorm(t[0], t[1], t[2], 8, B), [x, g, bt], tol=5e-5)
    x = _r(B * 9 * 11, 128, seed=7)
    _check('bilinear B=2', lambda t: T.bilinear(t[0], B, 9, 11, 17, 21, True), lambda t: S.bilinear(t[0], B, 9, 11, 17, 21, True), [x])
    d = _r(B * 8, 9 * 11, 9 * 11, seed=8)
    _check('window_gather G=16', lambda t: T.window_gather(t[0], 9, 11, 7, 0.0), lambda t: S.window_gather(t[0], 9, 11, 7, 0.0), [d])


def test_layout_changes(T):
    x = _r(33 * 41, 12)
    _check('to_nchw', lambda t: T.to_nchw(t[0], 33, 41), lambda t: S.to_nchw(t[0], 33, 41), [x])
    m = _r(1, 11, 33, 41, seed=1)
    _check('to_nhwc', lambda t: T.to_nhwc(t[0], 12), lambda t: S.to_nhwc(t[0], 12), [m])


def _train_engine(case, train_mode=False):
    from common import TRAIN_CFG, TRAIN_FWD_CASES, synth_model_state, train_batch
    from networks.engines import build_engine
    c = TRAIN_FWD_CASES[case]
    cfg, model, _ = synth_model_state(c['model'], cfg_overrides=TRAIN_CFG)
    model = model.cuda()
    model.train(train_mode)
    engine = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP)
    engine.train(train_mode)
    frames, masks, objs, perms = train_batch(case)
    engine.restart_engine(len(objs), perms is not None)
    if perms is not None:
        engine.id_shuffle = [p.cuda() for p in perms]
    kw = dict(step=c['step'], use_prev_pred=c.get('use_prev_pred', False), enable_prev_frame=c.get('enable_prev_frame', False),
              use_prev_prob=c.get('use_prev_prob', False))
    return c, cfg, model, engine, frames.cuda(), masks.cuda(), objs, kw


@pytest.mark.parametrize('case', ['tf_aott', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_training_step_matches_reference(T, case):
    """One training step on the device against the REAL reference (train_grads.npz): AOTEngine.forward with autograd on (every
    graph node a HIP kernel), `loss.backward()`, the gradient clip, one AdamW step over the reference's parameter groups.
      * loss within 1e-5 relative; the gradient of each of the 105 / 108 trainable parameters: L2 norm within 0.2 %, 64 sampled
        entries within 0.5 % of the rms entry, five small tensors in full; no gradient where the reference has none;
      * clip_grad_norm's total norm within 1e-4 relative;
      * the UPDATED parameters: on every sampled entry whose clipped reference gradient is above 1e-5 (where AdamW's first step,
        -lr * g / (|g| + 1e-8), is not at the mercy of the gradient's last bits) new - old within 1e-5 x the parameter's rms of
        the reference's; on the others within the step's bound lr * (1 + wd * |p|)."""
    from common import check_grads_against_golden, grad_sample_index
    from utils.learning import get_trainable_params
    from utils.optim import AdamW
    c, cfg, model, engine, frames, masks, objs, kw = _train_engine(case)
    g = np.load(os.path.join(GOLD, 'train_grads.npz'))
    model.zero_grad()
    loss, pred, frame_loss, _ = engine(frames, masks, len(objs), objs, **kw)
    assert loss.requires_grad and len(pred) == len(frame_loss) == c['frames']
    np.testing.assert_allclose(float(loss.detach()), float(g[case + '.loss']), rtol=1e-5)
    loss.backward()
    torch.cuda.synchronize()
    worst = check_grads_against_golden(case, {k: p.grad for k, p in model.named_parameters()}, g)
    lr, wd, clip = (float(v) for v in g[case + '.step.lr_wd_clip'])
    groups = get_trainable_params(model, lr, wd, use_frozen_bn=cfg.MODEL_FREEZE_BN, exclusive_wd_dict={},
                                  no_wd_keys=['absolute_pos_embed', 'relative_position_bias_table', 'relative_emb_v', 'conv_out'])
    names = [str(n) for n in g[case + '.names']]
    assert [gr['name'] for gr in groups if gr['params'][0].grad is not None] == names
    assert [gr['weight_decay'] for gr in groups if gr['name'] in set(names)] == pytest.approx(g[case + '.step.wd'].tolist())
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt = AdamW(groups, lr=lr, weight_decay=wd)
    total, scale = opt.clip_grad_norm(clip)
    assert total == pytest.approx(float(g[case + '.step.total_norm']), rel=1e-4)
    opt.step(grad_scale=scale)
    torch.cuda.synchronize()
    after = dict(model.named_parameters())
    worst_upd = 0.0
    for i, k in enumerate(names):
        p0 = before[k].double().flatten()
        d = (after[k].detach().double().flatten() - p0).cpu()
        idx = grad_sample_index(d.numel())
        ref_d = torch.from_numpy(g[case + '.step.dsample'][i][:idx.numel()])
        ref_g = torch.from_numpy(g[case + '.sample'][i][:idx.numel()]).double() * min(1.0, clip / (float(g[case + '.step.total_norm']) + 1e-6))
        rms = float(p0.norm()) / max(1.0, p0.numel()) ** 0.5
        firm = ref_g.abs() >= 1e-5
        err = (d[idx] - ref_d).abs()
        if firm.any():
            e = float(err[firm].max())
            assert e <= 1e-5 * max(rms, 1e-3), '%s: updated entries differ by %g (parameter rms %g)' % (k, e, rms)
            worst_upd = max(worst_upd, e / max(rms, 1e-3))
        bound = lr * (1.0 + float(g[case + '.step.wd'][i]) * p0[idx].abs().cpu()) * 1.001
        assert bool((d[idx].abs() <= bound).all()), '%s: a step larger than lr (1 + wd |p|)' % k
        if float(g[case + '.norm'][i]) / max(1.0, p0.numel()) ** 0.5 * min(1.0, clip / float(g[case + '.step.total_norm'])) >= 1e-6:
            # (a gradient that is rounding noise -- a key bias -- moves its parameter by noise too: no norm to compare)
            assert abs(float(d.norm()) - float(g[case + '.step.dnorm'][i])) <= 0.02 * float(g[case + '.step.dnorm'][i]) + 1e-9
    print('training step %s: loss %.6f; worst sampled gradient error %.2e of the rms entry; worst updated entry %.2e of the parameter rms'
          % (case, float(loss.detach()), worst, worst_upd))


@pytest.mark.parametrize('case', ['tf_aott', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_training_steps_reduce_the_loss(T, case):
    """Six steps of the trainer's loop in train mode (drop-path / Dropout2d drawing, trainer.py:460-519: forward -> backward ->
    clip 5.0 -> AdamW -> EMA) on one batch: the loss of the deterministic network -- measured by the OTHER form of
    AOTEngine.forward, the fused inference kernels under no_grad, which re-pack the updated weights -- goes down, and the EMA
    shadow follows the parameters."""
    from utils.ema import ExponentialMovingAverage, get_param_buffer_for_ema
    from utils.learning import get_trainable_params
    from utils.optim import AdamW
    c, cfg, model, engine, frames, masks, objs, kw = _train_engine(case, train_mode=True)

    def eval_loss():
        model.eval()
        with torch.no_grad():
            l = float(engine(frames, masks, len(objs), objs, **kw)[0])
        model.train()
        return l
    l0 = eval_loss()
    opt = AdamW(get_trainable_params(model, 2e-4, 0.07, use_frozen_bn=cfg.MODEL_FREEZE_BN, no_wd_keys=['relative_emb_v', 'conv_out']),
                lr=2e-4, weight_decay=0.07)
    ema_params = get_param_buffer_for_ema(model, update_buffer=False)
    ema = ExponentialMovingAverage(ema_params, decay=0.99)
    import time
    seen = []
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(6):
        opt.zero_grad()
        loss = engine(frames, masks, len(objs), objs, **kw)[0]
        loss.backward()
        _, scale = opt.clip_grad_norm(5.0)
        opt.step(grad_scale=scale)
        ema.update(ema_params)
        seen.append(float(loss.detach()))
    torch.cuda.synchronize()
    print('  %.1f ms per training step (batch %d x %d frames at %dx%d, forward + backward + clip + AdamW + EMA)'
          % ((time.time() - t0) / 6 * 1e3, len(objs), c['frames'], *c['size']))
    l1 = eval_loss()
    print('six training steps %s: eval loss %.4f -> %.4f (train-mode losses %s)' % (case, l0, l1, ' '.join('%.3f' % v for v in seen)))
    assert l1 < l0 - 0.02, 'the loss did not go down: %.4f -> %.4f' % (l0, l1)
    assert all(np.isfinite(seen))
    moved = max(float((s - p.detach()).abs().max()) for s, p in zip(ema.shadow_params, ema_params))
    assert 0 < moved < 6 * 2e-4 * 1.1


@pytest.mark.parametrize('case', ['tf_aott_prev', 'tf_deaott_prob'])
def test_prediction_feedback_freezes_the_identity_bank(T, case):
    """aot_engine.py:46,176-177: with `use_prev_pred` a TRAINING engine detaches every identity embedding (`freeze_id`), so the
    identity bank (and DeAOT's id_norm) receive no gradient and nothing flows back through a fed-back probability map; in eval
    mode -- how the gradient goldens were made -- the reference does not detach, and neither does this engine."""
    got = {}
    for mode in (True, False):
        c, cfg, model, engine, frames, masks, objs, kw = _train_engine(case, train_mode=mode)
        assert kw['use_prev_pred']
        model.zero_grad()
        loss = engine(frames, masks, len(objs), objs, **kw)[0]
        loss.backward()
        g = model.patch_wise_id_bank.weight.grad
        got[mode] = None if g is None else float(g.abs().max())
        if mode:
            assert g is None or float(g.abs().max()) == 0.0, 'the identity bank got a gradient in train mode with use_prev_pred'
            if hasattr(model, 'id_norm'):
                assert model.id_norm.weight.grad is None or float(model.id_norm.weight.grad.abs().max()) == 0.0
            assert float(model.decoder.conv_out.weight.grad.abs().max()) > 0
        else:
            assert g is not None and float(g.abs().max()) > 0
    print('identity-bank gradient max: train mode %s, eval mode %.3g' % (got[True], got[False]))


# ---- round 4: bf16 matrix-core products, flat training state, the data-parallel step ---------------------------------------------
@pytest.mark.parametrize('M,K,N', [(437, 256, 96), (1800, 1024, 256), (7000, 1152, 128), (5000, 64, 512), (1674, 512, 1024)])
def test_gemm_bf16_is_the_rounded_product(T, M, K, N):
    """aot_pack_bf16_f32 + aot_conv2d_bf16_f32 (precision 'bf16' of the training path): exactly the product of the operands
    ROUNDED TO NEAREST EVEN to bf16 -- torch's own .bfloat16() -- accumulated in fp32: against that product in fp64 the result
    differs by fp32 summation order only; against the unrounded fp32 product by the bf16 rounding (~2^-9 relative per operand)."""
    import aot_hip
    g = torch.Generator(device='cuda').manual_seed(M + K + N)
    a = torch.randn(M, K, device='cuda', generator=g) * 1.7
    w = torch.randn(K, N, device='cuda', generator=g) * 0.3
    bias = torch.randn(N, device='cuda', generator=g)
    a[3, 5], w[7, 2] = 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20          # ties / above-ties of the rounding
    out = aot_hip.gemm_bf16(a, w, bias)
    ref = a.bfloat16().double() @ w.bfloat16().double() + bias.double()
    scale = float(ref.abs().max())
    err = float((out.double() - ref).abs().max()) / scale
    assert err < 2e-6, 'bf16 product differs from the rounded-operand product by %.2e' % err
    full = a.double() @ w.double() + bias.double()
    rel = float((out.double() - full).norm() / full.norm())
    assert 1e-4 < rel < 1e-2, 'bf16 rounding error %.2e is not at the bf16 level' % rel


@pytest.mark.parametrize('M,K,N,ks', [(256, 4096, 256, 4), (200, 137 * 32 * 3, 130, 3), (64, 32 * 15, 64, 15), (512, 960 * 32, 64, 15)])
def test_gemm_bf16_split_k(T, M, K, N, ks):
    """The split-K form of aot_conv2d_bf16_f32 (weight gradients: K is the row count): the same rounded-operand product, the
    k-slices summed in order by the reduce pass; ragged M / N, bias added once."""
    import aot_hip
    g = torch.Generator(device='cuda').manual_seed(M + K + N)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) * 0.1
    bias = torch.randn(N, device='cuda', generator=g)
    out = aot_hip.gemm_bf16_packed(a, aot_hip.pack_bf16(w), N, bias, ks=ks)
    one = aot_hip.gemm_bf16_packed(a, aot_hip.pack_bf16(w), N, bias)
    ref = a.bfloat16().double() @ w.bfloat16().double() + bias.double()
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) / scale < 2e-6
    assert float((out - one).abs().max()) / scale < 2e-6
    with pytest.raises(aot_hip.AotHipError):
        aot_hip.gemm_bf16_packed(a, aot_hip.pack_bf16(w), N, bias, ks=7 if (K // 32) % 7 else 11)


def test_linear_bf16_forward_and_dgrad(T):
    """nn.Linear under train_ops.matmul_precision('bf16'): forward, dgrad and (split-K) wgrad on the bf16 matrix cores -- each
    exactly the product of its operands rounded to bf16, fp32 accumulation (the mode is recorded at forward time and used by
    backward wherever it runs); bias gradient in fp32."""
    x, w, b = _r(1800, 512, seed=1), _r(256, 512, seed=2, scale=0.05), _r(256, seed=3)
    with T.matmul_precision('bf16'):
        y = T.linear(x, w, b)
    ct = torch.randn_like(y)
    y.backward(ct)                                       # outside the context: the Function carries its mode
    yr = x.detach().bfloat16().double() @ w.detach().bfloat16().double().t() + b.detach().double()
    assert float((y.detach().double() - yr).abs().max()) / float(yr.abs().max()) < 2e-6
    dxr = ct.bfloat16().double() @ w.detach().bfloat16().double()
    assert float((x.grad.double() - dxr).abs().max()) / float(dxr.abs().max()) < 2e-6
    dwr = ct.bfloat16().double().t() @ x.detach().bfloat16().double()            # split-K over the 1800 rows (padded to a multiple of 32 ks)
    assert float((w.grad.double() - dwr).abs().max()) / float(dwr.abs().max()) < 2e-6
    assert float((b.grad.double() - ct.double().sum(0)).abs().max()) / float(ct.double().sum(0).abs().max()) < 2e-5


def test_flat_train_state_step_equals_per_tensor_adamw(T):
    """utils/flat_state.py::FlatTrainState.step (three launches over flat buffers: aot_sumsq_flat_f64, aot_adamw_flat_f32 with the
    clip factor taken from the device, aot_ema_update_f32) against the per-tensor path the reference-pinned step test uses
    (utils.optim.AdamW.clip_grad_norm + step, utils.ema): same parameters after three steps with per-tensor lr / weight decay, a
    tensor that never gets a gradient (skipped like `p.grad is None`) and a clip that binds."""
    from utils.ema import ExponentialMovingAverage
    from utils.flat_state import FlatTrainState
    from utils.optim import AdamW
    torch.manual_seed(3)
    shapes = [(64, 33), (33,), (5, 7, 3, 3), (1,), (1000, 257)]

    def make():
        g = torch.Generator().manual_seed(11)
        return [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes] + [torch.nn.Parameter(torch.ones(9).cuda())]
    pa, pb = make(), make()
    hyper = [(2e-3, 0.07), (1e-3, 0.0), (2e-4, 0.03), (5e-3, 0.0), (1e-3, 0.07), (1e-3, 0.07)]
    ga = [{'params': [p], 'lr': lr, 'weight_decay': wd, 'name': 'p%d' % i} for i, (p, (lr, wd)) in enumerate(zip(pa, hyper))]
    gb = [{'params': [p], 'lr': lr, 'weight_decay': wd, 'name': 'p%d' % i} for i, (p, (lr, wd)) in enumerate(zip(pb, hyper))]
    opt = AdamW(ga, lr=1e-3, weight_decay=0.07)
    ema = ExponentialMovingAverage(pa, decay=0.99)
    st = FlatTrainState(gb, ema=True, ema_decay=0.99)
    for it in range(3):
        gg = torch.Generator().manual_seed(100 + it)
        grads = [torch.randn(*s, generator=gg).cuda() * (30.0 if it == 1 else 0.01) for s in shapes]     # step 1: the clip binds
        opt.zero_grad()
        st.zero_grad()
        for p, q, g in zip(pa, pb, grads):                 # the last tensor never gets a gradient
            p.grad = g.clone()
            (q * g).sum().backward()
        total, scale = opt.clip_grad_norm(5.0)
        opt.step(grad_scale=scale)
        ema.update(pa)
        st.average()
        st.step(max_norm=5.0)
        assert st.grad_norm() == pytest.approx(total, rel=1e-6)
        assert pb[-1].grad is None
    for i, (p, q) in enumerate(zip(pa, pb)):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-7 * max(1.0, float(p.detach().abs().max())), 'tensor %d' % i
        assert float((ema.shadow_params[i] - st.shadow_of(q)).abs().max()) <= 1e-7 * max(1.0, float(p.detach().abs().max()))
    assert torch.equal(pb[-1].detach(), torch.ones(9, device='cuda')), 'a tensor without a gradient was decayed'


@pytest.mark.parametrize('case', ['tf_deaott_prob', 'tf_r50_deaotl'])
def test_train_step_bf16_close_to_fp32(T, case):
    """The training step with the conv / linear products of forward and dgrad on the bf16 matrix cores (precision 'bf16': BASELINE
    config 5) against the fp32 step that is pinned on the reference: loss within 1e-2 relative, the gradient of every tensor with
    a non-negligible norm within 8 % in L2 and at a cosine above 0.995 (bf16 carries 8 significand bits: ~0.4 % per product,
    accumulated over the depth of the graph), the total norm within 2 %."""
    c, cfg, model, engine, frames, masks, objs, kw = _train_engine(case)
    out = {}
    for prec in ('f32', 'bf16'):
        model.zero_grad()
        with T.matmul_precision(prec):
            loss = engine(frames, masks, len(objs), objs, **kw)[0]
            loss.backward()
        out[prec] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    l32, g32 = out['f32']
    l16, g16 = out['bf16']
    assert abs(l16 - l32) <= 1e-2 * abs(l32), 'loss %.5f (bf16) vs %.5f (fp32)' % (l16, l32)
    assert set(g16) == set(g32)
    tot32 = sum(float(g.double().pow(2).sum()) for g in g32.values()) ** 0.5
    tot16 = sum(float(g.double().pow(2).sum()) for g in g16.values()) ** 0.5
    assert abs(tot16 - tot32) <= 0.02 * tot32
    worst, worst_cos = 0.0, 1.0
    for k, g in g32.items():
        n = float(g.norm())
        if n < 1e-3 * tot32:
            continue
        d = float((g16[k] - g).norm()) / n
        cos = float((g16[k] * g).sum() / (g16[k].norm() * g.norm()))
        worst, worst_cos = max(worst, d), min(worst_cos, cos)
        assert d < 0.08 and cos > 0.995, '%s: bf16 gradient off by %.3f (cosine %.4f)' % (k, d, cos)
    print('bf16 step %s: loss %.5f vs %.5f; total norm %.4f vs %.4f; worst tensor %.3f of its norm, cosine %.5f'
          % (case, l16, l32, tot16, tot32, worst, worst_cos))


def test_train_step_object_nccl_world1(T):
    """networks/managers/trainer.py::TrainStep over a one-rank RCCL group (what the builder's box can run; tools/dev/train_ddp.py is
    the N-rank launcher): schedule -> forward -> backward with the buckets leaving from the hooks -> average -> clip + AdamW + EMA;
    the loss goes down over six steps in both precisions."""
    import torch.distributed as dist
    import importlib
    from networks.engines import build_engine
    from networks.managers.trainer import TrainStep
    from networks.models import build_vos_model
    from utils.synth import synth_clip, synth_state_dict
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        created = True
    try:
        for prec in ('f32', 'bf16'):
            cfg = importlib.import_module('configs.pre_ytb_dav').EngineConfig('test', 'deaott')      # BASELINE config 5's stage
            model = build_vos_model(cfg.MODEL_VOS, cfg)
            model.load_state_dict(synth_state_dict(model.state_dict()))
            model = model.cuda().train()
            engine = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0,
                                  long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP).train()
            step = TrainStep(cfg, model, engine, precision=prec, bucket_mb=4.0)
            fr, mk, objs = [], [], []
            for b in range(2):
                f, m, o, _ = synth_clip(60 + b, 4, (129, 161), (129, 161), 2 + b, device='cuda')
                fr.append(torch.cat(f, 0)); mk.append(m.expand(4, -1, -1, -1)); objs.append(2 + b)
            frames = torch.stack(fr, 1).reshape(8, 3, 129, 161).contiguous()
            masks = torch.stack(mk, 1).reshape(8, 1, 129, 161).contiguous().float()
            losses = [float(step(frames, masks, objs, i)[0]) for i in range(6)]
            assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, '%s: %s' % (prec, losses)
            assert len(step.state.buckets) >= 2
            print('TrainStep %s: losses %s, %d buckets' % (prec, ' '.join('%.3f' % v for v in losses), len(step.state.buckets)))
            step.state.close()
    finally:
        if created:
            dist.destroy_process_group()
