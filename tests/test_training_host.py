"""Host logic of the training-side slice: learning-rate schedule, parameter groups, EMA decay rule and the bucketed
gradient all-reduce -- against goldens of the REAL reference (tests/golden/training.json, make_golden.make_training).  No GPU."""
import json
import os

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
