"""Host logic of the training-side slice: learning-rate schedule, parameter groups, EMA decay rule and the bucketed
gradient all-reduce -- against goldens of the REAL reference (tests/golden/training.json, make_golden.make_training).  No GPU."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import GOLD, model_cfg


def _gold():
    with open(os.path.join(GOLD, 'training.json')) as f:
        return json.load(f)


def test_lr_schedule_matches_reference():
    """adjust_learning_rate (utils/learning.py:4-46): warm-up, polynomial / cosine decay, restarts, encoder ratio, frozen
    groups -- the returned base rate and every group's lr / weight_decay at 12 iterations of 3 configurations."""
    from utils.learning import adjust_learning_rate
    names = ['encoder.layer1.0.conv1.weight', 'LSTT.layers.0.norm1.weight', 'patch_wise_id_bank.weight', 'decoder.conv_out.bias']

    class Opt:
        def __init__(self):
            self.param_groups = [{'name': n, 'lr': 0., 'weight_decay': 0.07} for n in names]
    for cfg in _gold()['schedule']:
        for itr, row in zip((0, 1, 25, 49, 50, 51, 299, 300, 301, 500, 899, 999), cfg['rows']):
            o = Opt()
            now = adjust_learning_rate(o, 2e-4, itr=itr, **cfg['kw'])
            got = [now] + [g['lr'] for g in o.param_groups] + [g['weight_decay'] for g in o.param_groups]
            assert got == pytest.approx(row, rel=1e-12, abs=0), (cfg['kw'], itr)


def test_param_groups_match_reference():
    """get_trainable_params (utils/learning.py:49-90) on R50-AOTL: the same trainable tensors in the same order (encoder frozen
    up to TRAIN_ENCODER_FREEZE_AT, FrozenBN statistics are buffers) with the same weight decay -- exclusive overrides, 1-D
    exemptions with and without frozen BN, exempted keys."""
    from networks.models import build_vos_model
    from utils.learning import get_trainable_params
    cfg = model_cfg('r50_aotl')
    net = build_vos_model(cfg.MODEL_VOS, cfg)
    for use_frozen, ref in zip((True, False), _gold()['param_groups']):
        gs = get_trainable_params(net, 2e-4, 0.07, use_frozen_bn=use_frozen, exclusive_wd_dict={'relative_emb_k': 0.001, 'norm': 0.01},
                                  no_wd_keys=['pos_emb', 'mask_token'])
        assert [[g['name'], g['weight_decay']] for g in gs] == ref
        assert all(g['lr'] == 2e-4 and len(g['params']) == 1 for g in gs)


def test_ema_decay_rule_matches_reference():
    """min(decay, (1 + n) / (10 + n)) with n counted from 1 (utils/ema.py:57-62)."""
    from utils.ema import ExponentialMovingAverage
    ema = ExponentialMovingAverage([torch.nn.Parameter(torch.zeros(3))], decay=0.999)
    got = []
    for _ in _gold()['ema_decays']:
        got.append(ema.current_decay())
        ema.num_updates += 1
    assert got == pytest.approx(_gold()['ema_decays'], rel=1e-15)
    with pytest.raises(ValueError):
        ExponentialMovingAverage([], decay=1.5)


def _allreduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from utils.dist_grad import BucketedAllReduce
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300, 7), (5,), (64, 64), (1000,), (3, 3, 3))]
    params[3].requires_grad_(False)
    g = torch.Generator().manual_seed(100 + rank)
    for i, p in enumerate(params):
        if p.requires_grad and not (i == 1 and rank == 1):        # rank 1 has no gradient for tensor 1: counts as zeros
            p.grad = torch.randn(p.shape, generator=g)
    red = BucketedAllReduce(params, bucket_mb=0.01)               # 2621 floats per bucket: several buckets, one oversized tensor
    nb = len(red.buckets)
    red.average()
    q.put((rank, nb, [None if p.grad is None else p.grad.clone() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    """utils/dist_grad.BucketedAllReduce (the DDP gradient averaging of trainer.py:59-74): world 2 over gloo -- every rank ends
    with the mean of the ranks' gradients, a missing gradient counts as zeros, frozen tensors are left alone."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = ((300, 7), (5,), (64, 64), (1000,), (3, 3, 3))
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    want = []
    for i, shp in enumerate(shapes):
        if i == 3:
            want.append(None)
            continue
        a = torch.randn(shp, generator=gens[0])
        b = torch.randn(shp, generator=gens[1]) if i != 1 else torch.zeros(shp)
        want.append((a + b) / 2)
    assert res[0][1] >= 3
    for rank, _, grads in res:
        for got, w in zip(grads, want):
            if w is None:
                assert got is None
            else:
                assert torch.allclose(got, w, rtol=0, atol=1e-7)


@pytest.mark.parametrize('case', ['tf_aott', 'tf_aott_prev', 'tf_aott_shuffle', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_training_forward_orchestration_vs_reference(case, monkeypatch):
    """AOTEngine.forward's own logic -- frame order, which map is fed back, identity shuffle and its reversal, per-sample
    slicing, loss combination -- checked on CPU against the REAL reference's training engine (train_forward.npz) with the
    per-frame device stages stood in for by the oracle.  (The same forward on the real kernels: test_training_gpu.py.)"""
    import types

    import aot_hip
    from common import GOLD, TRAIN_CFG, TRAIN_FWD_CASES, model_cfg, synth_model_state, train_batch
    from networks.engines.aot_engine import AOTEngine
    from oracle.aot_oracle import OracleEngine, OracleModel, ce_topk_loss, soft_jaccard_loss
    c = TRAIN_FWD_CASES[case]
    g = np.load(os.path.join(GOLD, 'train_forward.npz'))
    _, _, sd = synth_model_state(c['model'])
    om = OracleModel(c['model'], sd)
    cfg = model_cfg(c['model'])

    class Staged(AOTEngine):
        def _restart_clip(self):
            super()._restart_clip()
            self._o = OracleEngine(om, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP)

        def add_reference_frame(self, img=None, mask=None, frame_step=-1, obj_nums=None, img_embs=None):
            if obj_nums is not None:
                self.obj_nums = obj_nums
            self._o.frame_step = self.frame_step
            self._o.add_reference_frame(img, mask, self.obj_nums)

        def match_propogate_one_frame(self, img=None, img_embs=None):
            self.frame_step += 1
            self._o.match_propogate_one_frame(img)

        def update_short_term_memory(self, curr_mask, curr_id_emb=None, skip_long_term_update=False):
            self._o.update_memory(curr_mask)

        def decode_current_logits(self, output_size=None):
            lg = self._o.decode_current_logits(output_size)
            self.pred_id_logits = self._o.pred_id_logits
            return lg

    def fuse_probs(logits, flips, new_label=None, want_aug_labels=True, want_prob=False, stream=None):
        return logits.argmax(1, keepdim=True).float(), None, torch.softmax(logits, 1) if want_prob else None
    monkeypatch.setattr(aot_hip, 'fuse_probs', fuse_probs)
    stub = types.SimpleNamespace(cfg=cfg, max_obj_num=10, parameters=lambda: iter([torch.zeros(1)]))
    eng = Staged(stub, gpu_id=0, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP)
    mining = TRAIN_CFG['TRAIN_HARD_MINING_RATIO'] * TRAIN_CFG['TRAIN_TOTAL_STEPS']
    eng.losses = [lambda lg, lb, step: ce_topk_loss(lg[0], lb[0], step, TRAIN_CFG['TRAIN_TOP_K_PERCENT_PIXELS'], mining),
                  lambda lg, lb, step: soft_jaccard_loss(lg[0], lb[0])]
    eng.loss_weights = [0.5, 0.5]
    eng.aux_weight = TRAIN_CFG['TRAIN_AUX_LOSS_WEIGHT']
    eng.aux_step = TRAIN_CFG['TRAIN_TOTAL_STEPS'] * TRAIN_CFG['TRAIN_AUX_LOSS_RATIO'] + 1e-5
    frames, masks, objs, perms = train_batch(case)
    eng.restart_engine(len(objs), perms is not None)
    if perms is not None:
        eng.id_shuffle = perms
    with torch.no_grad():
        loss, pred, frame_loss, _ = eng(frames, masks, len(objs), objs, step=c['step'],
                                        use_prev_pred=c.get('use_prev_pred', False),
                                        enable_prev_frame=c.get('enable_prev_frame', False),
                                        use_prev_prob=c.get('use_prev_prob', False))
    ref = g[case + '.masks']
    got = torch.stack(pred).numpy()
    ties = np.unpackbits(g[case + '.ties'])[:ref.size].reshape(ref.shape).astype(bool)
    assert int(((got != ref) & ~ties).sum()) == 0
    np.testing.assert_allclose(torch.stack(frame_loss).numpy(), g[case + '.frame_loss'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(loss), float(g[case + '.loss']), rtol=1e-4)


def test_pytorch_iou_matches_reference():
    """utils.metric.pytorch_iou on the 4-D maps the trainer passes (trainer.py:507-509: the reduction then runs over
    objects x rows, per image column) and on the documented 3-D maps; empty samples skipped, an object-free batch scores 1."""
    from utils.metric import pytorch_iou
    gi = torch.Generator().manual_seed(1)
    pred = torch.randint(0, 5, (3, 1, 20, 30), generator=gi)
    tgt = torch.randint(0, 5, (3, 1, 20, 30), generator=gi)
    for objs, r4, r3 in _gold()['pytorch_iou']:
        assert float(pytorch_iou(pred, tgt, objs)) == pytest.approx(r4, abs=1e-7)
        assert float(pytorch_iou(pred[:, 0], tgt[:, 0], objs)) == pytest.approx(r3, abs=1e-7)


def _toy():
    torch.manual_seed(5)
    return torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2, bias=False))


def _named_groups(net):
    return [{'params': [p], 'lr': 2e-4, 'weight_decay': 0.07 if p.dim() > 1 else 0., 'name': n} for n, p in net.named_parameters()]


def test_training_checkpoint_interop_with_reference(tmp_path):
    """utils.checkpoint / utils.optim.AdamW state dicts against a checkpoint written by the REAL reference's save_network
    with torch.optim.AdamW (tests/golden/ref_train_ckpt.pth): it resumes here (weights, moments, step counts, group
    settings), what is saved here resumes into torch.optim.AdamW, files are pruned to max_keep like the reference's, and the
    by-name resume (v2) survives reordered / new / vanished parameter groups."""
    from utils.checkpoint import load_network_and_optimizer, load_network_and_optimizer_v2, save_network
    from utils.optim import AdamW
    ref = torch.load(os.path.join(GOLD, 'ref_train_ckpt.pth'), map_location='cpu', weights_only=True)
    net = _toy()
    with torch.no_grad():
        for p in net.parameters():
            p.zero_()
    opt = AdamW(_named_groups(net), lr=1e-3, weight_decay=0.5)
    net, opt, removed = load_network_and_optimizer(net, opt, os.path.join(GOLD, 'ref_train_ckpt.pth'))
    assert removed == []
    for k, v in net.state_dict().items():
        assert torch.equal(v, ref['state_dict'][k])
    for i, g in enumerate(opt.param_groups):
        rg, rs = ref['optimizer']['param_groups'][i], ref['optimizer']['state'][i]
        assert g['name'] == rg['name'] and g['lr'] == rg['lr'] and g['weight_decay'] == rg['weight_decay']
        assert tuple(g['betas']) == tuple(rg['betas']) and g['eps'] == rg['eps']
        st = opt.state[g['params'][0]]
        assert st['step'] == 3 and torch.equal(st['exp_avg'], rs['exp_avg']) and torch.equal(st['exp_avg_sq'], rs['exp_avg_sq'])
    # written here -> resumes into torch.optim.AdamW (the reference's optimiser), pruned like the reference's files
    for step in (1, 2, 3):
        save_network(net, opt, step, str(tmp_path / 'ckpt'), max_keep=2)
    assert sorted(os.listdir(tmp_path / 'ckpt')) == _gold()['ckpt_kept']
    mine = torch.load(str(tmp_path / 'ckpt' / 'save_step_3.pth'), map_location='cpu', weights_only=True)
    net2 = _toy()
    topt = torch.optim.AdamW(_named_groups(net2), lr=1e-3, weight_decay=0.5)
    topt.load_state_dict(mine['optimizer'])
    for i, g in enumerate(topt.param_groups):
        rs = ref['optimizer']['state'][i]
        st = topt.state[g['params'][0]]
        assert float(st['step']) == 3.0 and torch.equal(st['exp_avg'], rs['exp_avg']) and g['lr'] == 2e-4
        assert g['name'] == ref['optimizer']['param_groups'][i]['name']
    # by name: groups reversed, one group the checkpoint does not know, one it knows dropped
    net3 = _toy()
    extra = torch.nn.Parameter(torch.ones(2))
    groups = _named_groups(net3)[::-1][:-1] + [{'params': [extra], 'lr': 5e-4, 'weight_decay': 0., 'name': 'extra'}]
    opt3 = AdamW(groups, lr=1e-3)
    load_network_and_optimizer_v2(net3, opt3, os.path.join(GOLD, 'ref_train_ckpt.pth'))
    names = [g['name'] for g in ref['optimizer']['param_groups']]
    for g in opt3.param_groups:
        p = g['params'][0]
        if g['name'] == 'extra':
            assert p not in opt3.state and g['lr'] == 5e-4
        else:
            rs = ref['optimizer']['state'][names.index(g['name'])]
            assert torch.equal(opt3.state[p]['exp_avg'], rs['exp_avg']) and opt3.state[p]['step'] == 3
    with pytest.raises(ValueError):
        AdamW(_named_groups(_toy())[:2], lr=1e-3).load_state_dict(ref['optimizer'])


def test_average_meter_and_zip_match_reference(tmp_path):
    """utils.meters.AverageMeter against the reference class on a sequence with weights and a reset (training.json), and
    utils.eval.zip_folder's archive layout (names start at the zipped folder's own name, utils/eval.py:5-13)."""
    import zipfile
    from utils.eval import zip_folder
    from utils.meters import AverageMeter
    m = AverageMeter(momentum=0.9)
    for i, (v, row) in enumerate(zip([1.0, 3.0, 2.0, 8.0, 5.0, 4.0], _gold()['meter'])):
        if i == 4:
            m.reset()
        m.update(v, n=1 + i % 2)
        assert [m.val, m.avg, m.moving_avg, m.count, m.long_count] == pytest.approx(row, rel=1e-12)
    root = tmp_path / 'eval' / 'Annotations'
    (root / 'seq_a').mkdir(parents=True)
    (root / 'seq_b').mkdir()
    for f in ('seq_a/00000.png', 'seq_a/00001.png', 'seq_b/00000.png'):
        (root / f).write_bytes(b'x')
    zip_folder(str(root), str(tmp_path / 'out.zip'))
    with zipfile.ZipFile(str(tmp_path / 'out.zip')) as z:
        assert sorted(z.namelist()) == ['Annotations/seq_a/00000.png', 'Annotations/seq_a/00001.png', 'Annotations/seq_b/00000.png']


def test_random_helpers_match_reference_streams():
    """utils.math under fixed CPU seeds draws what the reference's functions draw (training.json): generate_permute_matrix
    (background kept / not kept) and truncated_normal_; the engine's index form of the shuffle converts to the same matrices."""
    from utils.math import generate_permute_matrix, permutation_to_matrix, truncated_normal_
    gold = _gold()
    for keep in (True, False):
        torch.manual_seed(3)
        m = generate_permute_matrix(11, 4, keep, device=torch.device('cpu'))
        assert m.shape == (4, 11, 11) and torch.equal(m.sum(1), torch.ones(4, 11)) and torch.equal(m.sum(2), torch.ones(4, 11))
        assert m.argmax(-1).tolist() == gold['permute_cols'][str(keep)]
        assert all(torch.equal(permutation_to_matrix(m[i].argmax(-1)), m[i]) for i in range(4))
    torch.manual_seed(5)
    t = truncated_normal_(torch.zeros(3, 7), 0.1, 0.5)
    assert torch.allclose(t, torch.tensor(gold['trunc_normal']), atol=0, rtol=0) and (t - 0.1).abs().max() < 1.0


@pytest.mark.parametrize('case', ['tf_aott', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_training_graph_glue_vs_reference_gradients(case, monkeypatch):
    """The differentiable training forward (networks/models/train_forward.py, reached through AOTEngine.forward with autograd
    on) against the REAL reference's `loss.backward()` (tests/golden/train_grads.npz): loss and the gradient of every trainable
    parameter.  On CPU the graph's primitives are plain-torch stand-ins (tests/train_stand_ins.py) and the losses the
    oracle's, so this pins the GLUE -- weight layouts, which tensor feeds which op, memories carried through time, the
    shuffle and its reversal, what is frozen.  The same graph on the HIP kernels: tests/test_training_gpu.py."""
    import train_stand_ins
    from common import GOLD, TRAIN_CFG, TRAIN_FWD_CASES, check_grads_against_golden, synth_model_state, train_batch
    from networks.engines import build_engine
    from oracle.aot_oracle import ce_topk_loss, soft_jaccard_loss
    train_stand_ins.install(monkeypatch)
    c = TRAIN_FWD_CASES[case]
    g = np.load(os.path.join(GOLD, 'train_grads.npz'))
    cfg, model, _ = synth_model_state(c['model'], cfg_overrides=TRAIN_CFG)
    model.eval()                                                   # how the fixture was made: regularisers off
    eng = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP)
    eng.eval()        # the gradient goldens were made with engine.eval(): no freeze_id detach (aot_engine.py:176-177 needs self.training)
    mining = TRAIN_CFG['TRAIN_HARD_MINING_RATIO'] * TRAIN_CFG['TRAIN_TOTAL_STEPS']
    eng.losses = [lambda lg, lb, step: ce_topk_loss(lg[0], lb[0], step, TRAIN_CFG['TRAIN_TOP_K_PERCENT_PIXELS'], mining),
                  lambda lg, lb, step: soft_jaccard_loss(lg[0], lb[0])]
    eng.loss_weights = [0.5, 0.5]
    eng.aux_weight = TRAIN_CFG['TRAIN_AUX_LOSS_WEIGHT']
    eng.aux_step = TRAIN_CFG['TRAIN_TOTAL_STEPS'] * TRAIN_CFG['TRAIN_AUX_LOSS_RATIO'] + 1e-5
    frames, masks, objs, perms = train_batch(case)
    eng.restart_engine(len(objs), perms is not None)
    if perms is not None:
        eng.id_shuffle = perms
    model.zero_grad()
    loss, pred, frame_loss, _ = eng(frames, masks, len(objs), objs, step=c['step'], use_prev_pred=c.get('use_prev_pred', False),
                                    enable_prev_frame=c.get('enable_prev_frame', False),
                                    use_prev_prob=c.get('use_prev_prob', False))
    np.testing.assert_allclose(float(loss.detach()), float(g[case + '.loss']), rtol=1e-4)
    assert len(pred) == len(frame_loss) == c['frames'] and pred[0].shape == (len(objs), *c['size'])
    loss.backward()
    worst = check_grads_against_golden(case, {k: p.grad for k, p in model.named_parameters()}, g)
    print('training graph %s: loss %.6f, worst sampled gradient error %.2e of the rms entry' % (case, float(loss.detach()), worst))


def _ddp_worker(rank, world, port, case, q):
    """One rank of a data-parallel training step: its own sample of the batch through the differentiable forward (stand-in
    primitives on CPU), backward, bucketed all-reduce average of the MODEL's gradients."""
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    import train_stand_ins
    from common import TRAIN_CFG, TRAIN_FWD_CASES, synth_model_state, train_batch
    from networks.engines import build_engine
    from networks.layers import train_ops
    from oracle.aot_oracle import ce_topk_loss, soft_jaccard_loss
    from utils.dist_grad import BucketedAllReduce

    class Patch:
        def setattr(self, obj, name, value):
            setattr(obj, name, value)
    train_stand_ins.install(Patch())
    c = TRAIN_FWD_CASES[case]
    cfg, model, _ = synth_model_state(c['model'], cfg_overrides=TRAIN_CFG)
    model.eval()
    eng = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP)
    eng.eval()        # the gradient goldens were made with engine.eval(): no freeze_id detach (aot_engine.py:176-177 needs self.training)
    mining = TRAIN_CFG['TRAIN_HARD_MINING_RATIO'] * TRAIN_CFG['TRAIN_TOTAL_STEPS']
    eng.losses = [lambda lg, lb, step: ce_topk_loss(lg[0], lb[0], step, TRAIN_CFG['TRAIN_TOP_K_PERCENT_PIXELS'], mining),
                  lambda lg, lb, step: soft_jaccard_loss(lg[0], lb[0])]
    eng.loss_weights = [0.5, 0.5]
    eng.aux_weight = TRAIN_CFG['TRAIN_AUX_LOSS_WEIGHT']
    eng.aux_step = TRAIN_CFG['TRAIN_TOTAL_STEPS'] * TRAIN_CFG['TRAIN_AUX_LOSS_RATIO'] + 1e-5
    frames, masks, objs, perms = train_batch(case)
    bs = len(objs)
    assert bs == world
    mine = slice(rank, None, bs)                                   # time-major batch: this rank's sample of every frame
    eng.restart_engine(1, perms is not None)
    if perms is not None:
        eng.id_shuffle = [perms[rank]]
    loss, _, _, _ = eng(frames[mine], masks[mine], 1, [objs[rank]], step=c['step'], use_prev_pred=c.get('use_prev_pred', False),
                        enable_prev_frame=c.get('enable_prev_frame', False), use_prev_prob=c.get('use_prev_prob', False))
    loss.backward()
    BucketedAllReduce(list(model.parameters()), bucket_mb=1.0).average()
    if rank == 0:
        q.put({k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None})   # (by value)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('case', ['tf_aott', 'tf_deaott_prob'])
def test_data_parallel_step_averages_model_gradients_gloo_world2(case):
    """The DDP step of trainer.py:59-74 over gloo, world 2, with REAL model gradients: each rank runs one sample of the batch
    through the differentiable forward and backward, the bucketed all-reduce averages the parameters' gradients -- and the
    result is the reference's full-batch `loss.backward()` (train_grads.npz): the batch loss is the mean of per-sample
    losses, so averaging per-rank gradients is exactly what the reference's DistributedDataParallel computes."""
    import socket
    from common import GOLD, check_grads_against_golden
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    grads = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # (a parameter no rank has a gradient for -- LSTT.mask_token, unused -- comes out of the all-reduce as zeros)
    check_grads_against_golden(case, {k: torch.from_numpy(v) for k, v in grads.items() if v.any()},
                               np.load(os.path.join(GOLD, 'train_grads.npz')))


def test_training_regularisers_draw_per_sample():
    """The training-time regularisers of the differentiable forward (reference basic.py:46-55,129-148): DropPath drops a SAMPLE's
    whole branch with probability p and rescales the kept ones by 1 / (1 - p); Dropout2d drops whole channels of a sample; both
    are identities in eval mode; the stochastic-depth rates follow the reference's rules (constant, or growing linearly over the
    layers with TRAIN_LSTT_DROPPATH_SCALING; Swin: linspace(0, 0.3) over the blocks of all four stages, frozen blocks 0)."""
    from networks.models import train_forward as tf
    from networks.layers.transformer import _droppath_rate
    torch.manual_seed(0)
    B, N, C = 64, 5, 8
    x = torch.ones(B * N, C)
    assert tf._drop_path(x, 0.3, False, B) is x and tf._dropout2d(x, 0.1, False, B) is x and tf._drop_path(x, 0.0, True, B) is x
    y = tf._drop_path(x, 0.25, True, B).view(B, N * C)
    kept = y[:, 0] != 0
    assert bool(((y == 0).all(1) | (y == 1 / 0.75).all(1)).all()), 'a sample is kept or dropped as a whole'
    assert 0.55 < kept.float().mean() < 0.95
    z = tf._dropout2d(x, 0.5, True, B).view(B, N, C)
    assert bool((z == z[:, :1]).all()), 'a channel of a sample is kept or dropped over the whole map'
    assert set(z.unique().tolist()) == {0.0, 2.0} and not bool((z[0] == z[1]).all() and (z[1] == z[2]).all())
    assert [_droppath_rate(0.1, i, 3, False) for i in range(3)] == [0.1, 0.1, 0.1]
    assert [_droppath_rate(0.2, i, 3, True) for i in range(3)] == [0.0, 0.1, 0.2] and _droppath_rate(0.2, 0, 1, True) == 0
    from common import model_cfg
    from networks.models import build_vos_model
    cfg = model_cfg('swinb_deaotl')
    enc = build_vos_model(cfg.MODEL_VOS, cfg).encoder
    rates = [blk.drop_path_p for layer in enc.layers for blk in layer.blocks]
    want = torch.linspace(0, 0.3, 24).tolist()[:22]
    frozen = sum(len(l.blocks) for l in enc.layers[:max(0, cfg.TRAIN_ENCODER_FREEZE_AT - 1)]) if cfg.TRAIN_ENCODER_FREEZE_AT >= 2 else 0
    assert rates[:frozen] == [0.0] * frozen and rates[frozen:] == pytest.approx(want[frozen:])


def _flat_worker(rank, world, port, q):
    """One rank of FlatTrainState over gloo: a small MLP (plain torch modules: the state is backend-agnostic up to step())."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from utils.flat_state import FlatTrainState
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 8))
    unused = torch.nn.Parameter(torch.ones(5))                 # no rank ever uses it: must stay `grad is None`
    one_sided = torch.nn.Parameter(torch.ones(8))              # only rank 1 uses it: both ranks get the averaged gradient
    groups = [{'params': [unused], 'name': 'unused'}, {'params': [one_sided], 'name': 'one_sided'}]
    groups += [{'params': [p], 'lr': 1e-3, 'weight_decay': 0.0, 'name': k} for k, p in net.named_parameters()]
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    st = FlatTrainState(groups, bucket_mb=512 * 4 / 2 ** 20, group=None)       # 512 elements: one layer per bucket
    assert all(torch.equal(before[k], p.detach()) for k, p in net.named_parameters())   # re-pointing keeps the values
    assert all(p.data_ptr() >= st.flat_p.data_ptr() and p.data_ptr() < st.flat_p.data_ptr() + 4 * st.total for p in net.parameters())
    x = torch.randn(world, 4, 16, generator=torch.Generator().manual_seed(5))
    st.zero_grad()
    out = net(x[rank])
    loss = out.pow(2).mean() + ((out * one_sided).sum() * 0.01 if rank == 1 else 0.0)
    loss.backward()
    in_bwd, order = st.launched_in_backward, list(st.launch_order)
    st.average()
    res = {'rank': rank, 'in_bwd': in_bwd, 'order': order, 'nbuckets': len(st.buckets),
           'grads': {k: p.grad.numpy().copy() for k, p in net.named_parameters()},          # (by value: the worker exits)
           'unused_none': unused.grad is None, 'one_sided': None if one_sided.grad is None else one_sided.grad.numpy().copy(),
           'views': all(p.grad.data_ptr() >= st.flat_g.data_ptr() and p.grad.data_ptr() < st.flat_g.data_ptr() + 4 * st.total
                        for p in net.parameters())}
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_folded_weight_cache_respects_grad_mode():
    """inside a weight_cache() scope the folded conv weight is cached per step -- a fold first formed under
    torch.no_grad() (the aux decodes when their weight is 0) must not be what a later grad-enabled use gets."""
    from networks.layers import train_ops as T
    from networks.models.train_forward import _fold_bn

    class BN:          # FrozenBatchNorm2d's constants
        weight, bias = torch.ones(4), torch.zeros(4)
        running_mean, running_var, epsilon = torch.zeros(4), torch.ones(4), 1e-5
    w = torch.nn.Parameter(torch.randn(4, 3, 1, 1))
    with T.weight_cache():
        with torch.no_grad():
            a, _ = _fold_bn(w, BN)
        b, _ = _fold_bn(w, BN)
        c, _ = _fold_bn(w, BN)
    assert not a.requires_grad and b.requires_grad and b is c
    b.sum().backward()
    assert w.grad is not None and float(w.grad.abs().sum()) > 0


def _flat_order_worker(rank, world, port, q):
    """the one-sided tensor ALONE in a middle bucket; rank 0 never touches it, rank 1 does.  Both ranks must
    issue the same collective sequence (bucket sizes differ, so a different order is a wire-level mismatch), replicas must start
    from rank 0's values although every rank seeds its own, and the optimiser state must survive a named_state round trip."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from utils.flat_state import FlatTrainState
    torch.manual_seed(100 + rank)                              # different initial weights per rank: the constructor must fix that
    head = torch.nn.Linear(32, 8)
    trunk = torch.nn.Linear(16, 32)
    one_sided = torch.nn.Parameter(torch.ones(600))            # registered BETWEEN trunk and head: a bucket of its own in the middle
    groups = [{'params': [p], 'name': 'trunk.' + k} for k, p in trunk.named_parameters()]
    groups += [{'params': [one_sided], 'name': 'one_sided'}]
    groups += [{'params': [p], 'name': 'head.' + k} for k, p in head.named_parameters()]
    st = FlatTrainState(groups, bucket_mb=256 * 4 / 2 ** 20, group=None)       # 256 elements per bucket
    w0 = [p.detach().clone() for p in list(trunk.parameters()) + list(head.parameters())]
    gathered = [None] * world
    dist.all_gather_object(gathered, [w.numpy().tolist() for w in w0])
    same_start = gathered[0] == gathered[1]
    mid = [i for i, b in enumerate(st.buckets) if b['members'] == [st._index[id(one_sided)]]]
    x = torch.randn(world, 4, 16, generator=torch.Generator().manual_seed(7))
    st.zero_grad()
    out = head(torch.relu(trunk(x[rank])))
    loss = out.pow(2).mean() + (one_sided.sum() * 0.01 if rank == 1 else 0.0)
    loss.backward()
    in_bwd = list(st.launch_order)
    st.average()
    named = st.named_state()
    st2_moments = {k: {'step': 3, 'exp_avg': v['exp_avg'] + 1.0, 'exp_avg_sq': v['exp_avg_sq'] + 2.0} for k, v in named.items()}
    st.load_named_state(st2_moments, ema_updates=5)
    back = st.named_state()
    round_trip = all(back[k]['step'] == 3 and torch.equal(back[k]['exp_avg'], st2_moments[k]['exp_avg']) and
                     torch.equal(back[k]['exp_avg_sq'], st2_moments[k]['exp_avg_sq']) for k in named) and st.ema_updates == 5
    q.put({'rank': rank, 'same_start': same_start, 'mid': mid, 'nbuckets': len(st.buckets), 'in_bwd': in_bwd,
           'order': list(st.launch_order), 'one_sided': one_sided.grad.numpy().copy(), 'round_trip': round_trip})
    dist.barrier()
    dist.destroy_process_group()


def test_flat_train_state_rank_independent_collective_order_gloo_world2():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_order_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0['same_start'] and r1['same_start'], 'the constructor did not broadcast rank 0 parameters'
    assert len(r0['mid']) == 1 and 0 < r0['mid'][0] < r0['nbuckets'] - 1, 'the one-sided tensor is not alone in a middle bucket: %s' % r0
    assert r0['order'] == r1['order'] == list(range(r0['nbuckets'])), 'ranks issued different collective sequences: %s / %s' % (r0['order'], r1['order'])
    m = r0['mid'][0]
    # rank 0 never completes the middle bucket in backward: it (and everything behind it) waits for average(); rank 1 runs through
    assert r0['in_bwd'] == list(range(m)) and r1['in_bwd'] == list(range(r1['nbuckets']))
    assert np.allclose(r0['one_sided'], 0.005) and np.allclose(r1['one_sided'], 0.005)      # the average of (0, 0.01)
    assert r0['round_trip'] and r1['round_trip']


def test_flat_train_state_overlaps_and_averages_gloo_world2():
    """utils/flat_state.py over gloo, world 2 (trainer.py:59-74's DistributedDataParallel): parameters and gradients are views of
    the flat buffers; a bucket's all-reduce is issued from the post-accumulate-grad hooks WHILE backward runs, first the bucket of
    the last layers; the averaged gradients equal the full-batch gradients; a tensor no rank used keeps `grad is None` (torch's
    AdamW skips it), one that a single rank used gets the averaged gradient on both (find_unused_parameters = True)."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r['rank'])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the same computation in one process on the whole batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 8))
    one_sided = torch.nn.Parameter(torch.ones(8))
    x = torch.randn(2, 4, 16, generator=torch.Generator().manual_seed(5))
    o0, o1 = net(x[0]), net(x[1])
    (0.5 * (o0.pow(2).mean() + o1.pow(2).mean() + (o1 * one_sided).sum() * 0.01)).backward()
    for r in res:
        assert r['views'] and r['unused_none']
        assert r['nbuckets'] >= 3 and r['in_bwd'] >= r['nbuckets'] - 1, 'buckets were not issued from the backward hooks: %s' % r
        assert r['order'][0] == 0 and r['order'] == sorted(r['order'])            # last layers first, in production order
        for k, p in net.named_parameters():
            assert np.allclose(r['grads'][k], p.grad.numpy(), atol=1e-7), k
        assert np.allclose(r['one_sided'], one_sided.grad.numpy(), atol=1e-8)
