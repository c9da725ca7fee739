"""Flat training state: the data-parallel step of the reference's trainer (networks/managers/trainer.py:59-74 DistributedDataParallel,
:116-118 AdamW, :501-503 clip_grad_norm_ + step, :516-517 EMA) re-designed around four flat fp32 buffers.

The reference wraps the engine in DistributedDataParallel (bucketed all-reduce overlapped with backward) and then walks the
parameters three more times per step (gradient norm, AdamW with one parameter group PER TENSOR -- utils/learning.py builds them
so --, EMA): ~700 small launches for R50-DeAOTL.  Here every trainable tensor is a range of

    params | grads | exp_avg | exp_avg_sq | (rank 0) EMA shadow          one buffer each, tensors in REVERSE registration order

and `p.data` / `p.grad` are views into them, so that

  * backward accumulates straight into the bucket memory (no pack / unpack copies); a bucket is a contiguous range of the
    gradient buffer, and the post-accumulate-grad hook of its last tensor issues ONE asynchronous all-reduce for the range while
    backward is still running (RCCL on its own stream; xGMI rings are bound per link, so few large collectives);
  * the gradient norm is one launch (aot_sumsq_flat_f64), AdamW one launch with a per-tensor table of (lr, weight decay, bias
    corrections) and the clip factor taken from the device-resident norm (aot_adamw_flat_f32: no host synchronisation in a
    step), the EMA one launch (aot_ema_update_f32);
  * tensors that received no gradient on ANY rank are skipped by the optimiser exactly as torch.optim.AdamW skips
    `p.grad is None` (DDP's find_unused_parameters = True semantics, trainer.py:72): the hooks record which tensors were touched
    and one small MAX all-reduce agrees on the set (distributed runs only: its result comes back to the host, ONE small
    synchronisation per step in front of the optimiser launches; a single process never waits);
  * every rank issues the SAME sequence of collectives whatever it touched (round 5): bucket i goes out only after
    buckets 0 .. i-1 -- from the hooks while backward runs when its turn has come, otherwise in average() -- and the touched-set
    all-reduce comes last, after every bucket.  (Issuing a bucket as soon as ITS members had fired let two ranks that used
    different tensors order their collectives differently: mismatched sizes on the wire.)
  * construction broadcasts rank 0's parameters (what DistributedDataParallel does, trainer.py:59-74): replicas start identical
    whatever seed each rank built its model with.

The collective layer is torch.distributed (`nccl` = RCCL on the GPU box, `gloo` in the CPU tests); the optimiser kernels need
the device (no CPU fallback: `step()` raises on CPU tensors, `average()` alone is backend-agnostic).
"""
import math

import torch
import torch.distributed as dist


class FlatTrainState:
    def __init__(self, param_groups, betas=(0.9, 0.999), eps=1e-8, bucket_mb=32.0, group=None, ema_decay=None, ema=False):
        """param_groups: utils.learning.get_trainable_params(...) -- dicts with 'params' (one tensor each), 'lr', 'weight_decay',
        'name'.  The tensors are re-pointed into the flat parameter buffer (values kept)."""
        self.groups = []
        for g in param_groups:
            for p in g['params']:
                if p.requires_grad:
                    self.groups.append({'param': p, 'lr': g.get('lr', 1e-3), 'weight_decay': g.get('weight_decay', 0.0),
                                        'name': g.get('name', ''), 'step': 0})
        if not self.groups:
            raise ValueError('no trainable parameters')
        self.groups.reverse()                 # the order backward produces the gradients in: the first bucket fills first
        self.betas, self.eps, self.group = betas, eps, group
        dev = self.groups[0]['param'].device
        offs, n = [], 0
        for g in self.groups:
            offs.append(n)
            n += (g['param'].numel() + 3) // 4 * 4          # 16-byte aligned ranges (vector loads, collective alignment)
        self.offsets, self.total = offs + [n], n
        mk = lambda: torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq = mk(), mk(), mk(), mk()
        for g, o in zip(self.groups, offs):
            p = g['param']
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            g['grad_view'] = self.flat_g[o:o + p.numel()].view_as(p)
        if self.distributed() and dist.get_world_size(group) > 1:
            # rank 0's values everywhere (DistributedDataParallel's constructor does the same): one broadcast of the flat buffer
            dist.broadcast(self.flat_p, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.shadow = self.flat_p.clone() if ema else None
        self.ema_decay, self.ema_updates = ema_decay, 0
        # buckets: contiguous ranges of the gradient buffer, cut at tensor boundaries
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets, start, members = [], 0, []
        for i, g in enumerate(self.groups):
            members.append(i)
            if self.offsets[i + 1] - start >= cap or i == len(self.groups) - 1:
                self.buckets.append({'range': (start, self.offsets[i + 1]), 'members': members, 'pending': 0, 'work': None,
                                     'ready': False})
                start, members = self.offsets[i + 1], []
        self._bucket_of = {}
        for b in self.buckets:
            for i in b['members']:
                self._bucket_of[i] = b
        self._index = {id(g['param']): i for i, g in enumerate(self.groups)}
        self._touched = torch.zeros(len(self.groups), dtype=torch.int32)          # host
        self._touched_dev = torch.zeros(len(self.groups), dtype=torch.int32, device=dev)
        self._touch_work = None
        self.launch_order = []                 # bucket indices in the order their all-reduce was issued (tests read it)
        self._next_bucket = 0
        self.launched_in_backward = 0
        self._in_backward = False
        self._hooks = [g['param'].register_post_accumulate_grad_hook(self._ready) for g in self.groups]
        self._seg_off = torch.tensor(self.offsets, dtype=torch.int64, device=dev)
        # the per-tensor table goes to the device asynchronously and the host never waits for a step: two pinned staging buffers
        # used in turn, each guarded by the event of its last copy (a host that runs a whole step ahead of the device must not
        # overwrite a table whose copy has not executed yet)
        self._hyp_hosts = [torch.zeros(len(self.groups), 4, dtype=torch.float32, pin_memory=dev.type == 'cuda') for _ in range(2)]
        self._hyp_events = [None, None]
        self._hyp_turn = 0
        self._hyp = torch.zeros(len(self.groups), 4, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._scratch = (torch.zeros(1024, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))

    # ---- optimizer.param_groups look-alike: adjust_learning_rate() writes 'lr' / 'weight_decay' by 'name' -------------------
    @property
    def param_groups(self):
        return self.groups

    def world(self):
        return dist.get_world_size(self.group) if self.distributed() else 1

    @staticmethod
    def distributed():
        """A process group exists: the buckets go through it even when it has one rank (the N = 1 run of tools/dev/train_ddp.py
        exercises the same RCCL calls an 8-rank job issues)."""
        return dist.is_available() and dist.is_initialized()

    # ---- one step ---------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        """One fill; every tensor's .grad is (again) its range of the gradient buffer, so autograd accumulates in place."""
        self.flat_g.zero_()
        for g in self.groups:
            g['param'].grad = g['grad_view']
        for b in self.buckets:
            b['pending'], b['work'], b['ready'] = len(b['members']), None, False
        self._next_bucket = 0                  # buckets go out strictly in index order: the same order on every rank
        self._touched.zero_()
        self._touch_work = None
        self.launch_order, self.launched_in_backward, self._in_backward = [], 0, True

    def _ready(self, p):
        i = self._index[id(p)]
        if p.grad is not self.groups[i]['grad_view']:       # someone replaced .grad (zero_grad(set_to_none)): fold it back in
            self.groups[i]['grad_view'].copy_(p.grad)
            p.grad = self.groups[i]['grad_view']
        self._touched[i] = 1
        b = self._bucket_of[i]
        b['pending'] -= 1
        if b['pending'] == 0:
            b['ready'] = True
            # rank-independent order: a complete bucket waits for its predecessors (a rank that never touches a tensor of bucket
            # j < i issues j in average(), and i after it -- exactly what a rank that touched everything does)
            while self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket]['ready']:
                self._launch(self.buckets[self._next_bucket])
                self._next_bucket += 1

    def _launch(self, b):
        if b['work'] is not None or not self.distributed():
            b['work'] = b['work'] or True
            return
        lo, hi = b['range']
        self.launch_order.append(self.buckets.index(b))
        if self._in_backward:
            self.launched_in_backward += 1
        b['work'] = dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def average(self):
        """After backward: issues what the hooks could not (buckets holding a tensor without a gradient), waits, divides by the
        world size, and agrees on the set of tensors that got a gradient on any rank."""
        self._in_backward = False
        world = self.world()
        if self.distributed():
            for b in self.buckets[self._next_bucket:]:           # what the hooks could not issue, in index order
                self._launch(b)
            self._next_bucket = len(self.buckets)
            self._touched_dev.copy_(self._touched)
            self._touch_work = dist.all_reduce(self._touched_dev, op=dist.ReduceOp.MAX, group=self.group, async_op=True)   # always last
            for b in self.buckets:
                b['work'].wait()
            if world > 1:
                self.flat_g.mul_(1.0 / world)
            self._touch_work.wait()
            self._touched.copy_(self._touched_dev)
        for g, t in zip(self.groups, self._touched.tolist()):
            if not t:
                g['param'].grad = None          # what the reference's optimizer sees for an unused tensor

    def step(self, max_norm=0.0):
        """clip_grad_norm_(max_norm) + AdamW + (if built with ema=True) the EMA update: three launches, no host sync.
        Returns the device scalar holding sum(grad^2) (sqrt = the total norm before clipping)."""
        import aot_hip
        b1, b2 = self.betas
        touched = self._touched.tolist()
        turn = self._hyp_turn
        self._hyp_turn = turn ^ 1
        if self._hyp_events[turn] is not None:
            self._hyp_events[turn].synchronize()          # (the copy issued two steps ago: done long since, unless the host ran ahead)
        host = self._hyp_hosts[turn]
        for i, (g, t) in enumerate(zip(self.groups, touched)):
            if t:
                g['step'] += 1
                st = g['step']
                host[i, 0] = g['lr']
                host[i, 1] = g['weight_decay']
                host[i, 2] = 1.0 - b1 ** st
                host[i, 3] = math.sqrt(1.0 - b2 ** st)
            else:
                host[i, 0] = -1.0
        self._hyp.copy_(host, non_blocking=True)
        if self._hyp.is_cuda:
            self._hyp_events[turn] = torch.cuda.Event()
            self._hyp_events[turn].record()
        with torch.no_grad():
            aot_hip.sumsq_flat(self.flat_g, self._scratch, self._sumsq)
            aot_hip.adamw_flat(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self._seg_off, self._hyp, b1, b2, self.eps,
                               sumsq=self._sumsq, max_norm=max_norm)
            if self.shadow is not None:
                n = self.ema_updates + 1
                decay = min(self.ema_decay, (1 + n) / (10 + n)) if self.ema_decay is not None else 0.0      # utils/ema.py:57-62
                self.ema_updates = n
                aot_hip.ema_update(self.shadow, self.flat_p, 1.0 - decay)
        return self._sumsq

    def grad_norm(self):
        return float(self._sumsq.item()) ** 0.5

    # ---- views for checkpoints / tests ---------------------------------------------------------------------------------------
    def named_state(self):
        """{name: {'step', 'exp_avg', 'exp_avg_sq'}} as views (the per-tensor layout torch.optim.AdamW checkpoints use)."""
        out = {}
        for g, o in zip(self.groups, self.offsets):
            p = g['param']
            out[g['name']] = {'step': g['step'], 'exp_avg': self.exp_avg[o:o + p.numel()].view_as(p),
                              'exp_avg_sq': self.exp_avg_sq[o:o + p.numel()].view_as(p)}
        return out

    def load_named_state(self, state, ema_updates=None):
        """Resume: the inverse of named_state() -- {name: {'step', 'exp_avg', 'exp_avg_sq'}} in the per-tensor layout of a
        torch.optim.AdamW checkpoint (the reference resumes its optimiser and ema.num_updates, trainer.py:170-215); tensors are
        copied into the flat moment buffers, names this state does not hold are ignored, missing names keep zero moments."""
        for g, o in zip(self.groups, self.offsets):
            ent = state.get(g['name'])
            if ent is None:
                continue
            p = g['param']
            g['step'] = int(ent['step'])
            self.exp_avg[o:o + p.numel()].view_as(p).copy_(torch.as_tensor(ent['exp_avg']).to(self.exp_avg.device))
            self.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(torch.as_tensor(ent['exp_avg_sq']).to(self.exp_avg_sq.device))
        if ema_updates is not None:
            self.ema_updates = int(ema_updates)
        return self

    def shadow_of(self, p):
        i = self._index[id(p)]
        o = self.offsets[i]
        return self.shadow[o:o + p.numel()].view_as(p)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
