"""Optimiser of the trainer (reference networks/managers/trainer.py:116-118,501-503): AdamW over named parameter groups and the
global gradient-norm clip, with the element-wise work as device kernels (csrc/train_ops.hip)."""
import torch

import aot_hip


class AdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, no amsgrad) over the groups produced by
    utils.learning.get_trainable_params ({'params', 'lr', 'weight_decay', 'name'})."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.param_groups = []
        for g in params:
            g = dict(g) if isinstance(g, dict) else {'params': [g]}
            g.setdefault('lr', lr)
            g.setdefault('weight_decay', weight_decay)
            g.setdefault('betas', betas)
            g.setdefault('eps', eps)
            g.setdefault('name', '')
            self.param_groups.append(g)
        self.state = {}

    def zero_grad(self):
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    def clip_grad_norm(self, max_norm):
        """clip_grad_norm_(parameters, max_norm): returns (total L2 norm, factor the step multiplies the gradients by)."""
        grads = [p.grad for g in self.param_groups for p in g['params'] if p.grad is not None]
        if not grads:
            return 0.0, 1.0
        acc = torch.zeros(1, dtype=torch.float64, device=grads[0].device)
        # one launch over all gradients laid end to end (a launch per tensor was ~5 % of a training step's kernel time)
        aot_hip.sumsq_accum(torch.cat([gr.reshape(-1) for gr in grads]) if len(grads) > 1 else grads[0].contiguous(), acc)
        total = float(acc.item()) ** 0.5
        return total, min(1.0, max_norm / (total + 1e-6))

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        for g in self.param_groups:
            b1, b2 = g['betas']
            for p in g['params']:
                if p.grad is None:
                    continue
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}
                st['step'] += 1
                aot_hip.adamw_step(p, p.grad.contiguous(), st['exp_avg'], st['exp_avg_sq'], g['lr'], g['weight_decay'], b1, b2,
                                   g['eps'], st['step'], grad_scale)

    # ---- checkpointing: the layout of torch.optim.AdamW's state dict, so that checkpoints written by the reference's
    # trainer (utils/checkpoint.py:124-160) resume here and the other way round ------------------------------------------
    _TORCH_GROUP_KEYS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                             decoupled_weight_decay=True)

    def state_dict(self):
        index, groups = {}, []
        for g in self.param_groups:
            out = {k: v for k, v in g.items() if k != 'params'}
            for k, v in self._TORCH_GROUP_KEYS.items():
                out.setdefault(k, v)
            out['params'] = [index.setdefault(id(p), len(index)) for p in g['params']]
            groups.append(out)
        state = {}
        for p, st in self.state.items():
            state[index[id(p)]] = {'step': torch.tensor(float(st['step'])), 'exp_avg': st['exp_avg'],
                                   'exp_avg_sq': st['exp_avg_sq']}
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        saved = sd['param_groups']
        if len(saved) != len(self.param_groups) or any(len(a['params']) != len(b['params'])
                                                       for a, b in zip(saved, self.param_groups)):
            raise ValueError('loaded state dict has different parameter groups')
        by_index = {}
        for g_saved, g in zip(saved, self.param_groups):
            for i, p in zip(g_saved['params'], g['params']):
                by_index[i] = p
            for k, v in g_saved.items():
                if k != 'params' and k not in self._TORCH_GROUP_KEYS:
                    g[k] = tuple(v) if k == 'betas' else v
        self.state = {}
        for i, st in sd['state'].items():
            p = by_index[int(i)]
            self.state[p] = {'step': int(float(st['step'])),
                             'exp_avg': st['exp_avg'].to(device=p.device, dtype=p.dtype).clone(),
                             'exp_avg_sq': st['exp_avg_sq'].to(device=p.device, dtype=p.dtype).clone()}

