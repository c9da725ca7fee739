"""CPU census of the training graph's autograd nodes (which torch glue surrounds the C-ABI primitives): builds one of the golden
training cases with the plain-torch stand-ins of tests/train_stand_ins.py, runs the forward and counts the backward nodes by type
and -- for the view / index nodes, whose backward is a zero-fill plus an add on the GPU -- by the source line that made them.
    python tools/dev/cpu_graph_census.py [tf_r50_deaotl]"""
import collections
import os
import sys
import traceback

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, 'aot-benchmark_amd'), os.path.join(R, 'tests'), R]
import torch

case = sys.argv[1] if len(sys.argv) > 1 else 'tf_r50_deaotl'
import train_stand_ins
from common import TRAIN_CFG, TRAIN_FWD_CASES, synth_model_state, train_batch
from networks.engines import build_engine
from oracle.aot_oracle import ce_topk_loss, soft_jaccard_loss


class Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


train_stand_ins.install(Patch())
c = TRAIN_FWD_CASES[case]
cfg, model, _ = synth_model_state(c['model'], cfg_overrides=TRAIN_CFG)
model.train()
eng = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP)
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

# source line of every view / index op that records a graph node
sites = collections.Counter()
WATCH = ('__getitem__', 'narrow', 'select', 'chunk', 'split', 'unbind', 'index_select', '__mul__', '__rmul__', '__truediv__', '__add__',
         '__radd__', '__sub__', '__rsub__', 'contiguous', 'expand', 'sum', 'mean', 'mul', 'add', 'div')
orig = {n: getattr(torch.Tensor, n) for n in WATCH}


def wrap(name):
    def f(self, *a, **k):
        out = orig[name](self, *a, **k)
        if torch.is_grad_enabled() and isinstance(self, torch.Tensor) and self.requires_grad:
            for fr in reversed(traceback.extract_stack()[:-1]):
                if 'aot-benchmark_amd' in fr.filename:
                    sites[(name, os.path.basename(fr.filename), fr.lineno, fr.line.strip()[:90])] += 1
                    break
        return out
    return f


for n in WATCH:
    setattr(torch.Tensor, n, wrap(n))
loss, pred, frame_loss, _ = eng(frames, masks, len(objs), objs, step=c['step'], use_prev_pred=c.get('use_prev_pred', False),
                                enable_prev_frame=c.get('enable_prev_frame', False), use_prev_prob=c.get('use_prev_prob', False))
for n in WATCH:
    setattr(torch.Tensor, n, orig[n])
loss = loss.mean()
seen, stack, cnt = set(), [loss.grad_fn], collections.Counter()
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    cnt[type(fn).__name__] += 1
    stack.extend(f for f, _ in fn.next_functions)
print('%s: %d autograd nodes' % (case, len(seen)))
for k, v in cnt.most_common(40):
    print('  %-40s %d' % (k, v))
print('view / index ops on tensors that require grad, by source line:')
for (name, f, ln, src), v in sites.most_common(int(os.environ.get('TOP', '70'))):
    print('  %4d  %-12s %s:%d  %s' % (v, name, f, ln, src))

# which backward nodes zero-fill / add / copy (each is a launch of its own on the GPU)
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU]) as prof:
    loss.backward()
evs = prof.events()
by_parent = collections.Counter()
for e in evs:
    if e.name in ('aten::fill_', 'aten::zero_', 'aten::add_', 'aten::add', 'aten::copy_', 'aten::mul', 'aten::sum', 'aten::cat', 'aten::div'):
        p = e.cpu_parent
        while p is not None and not p.name.startswith('autograd::engine::evaluate_function'):
            p = p.cpu_parent
        by_parent[(e.name, p.name.split(': ')[-1] if p is not None else '-')] += 1
print('backward: elementwise launches by autograd node:')
for (op, node), v in by_parent.most_common(45):
    print('  %5d  %-12s %s' % (v, op, node))
