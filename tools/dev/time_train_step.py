"""One training step (forward with the autograd graph -> backward -> clip -> AdamW -> EMA) at the recipe's size:
    python tools/dev/time_train_step.py [model=r50_deaotl] [batch=2] [frames=5] [size=465]
configs/default.py: DATA_RANDOMCROP 465 x 465, DATA_SEQ_LEN 5, TRAIN_BATCH_SIZE 16 over 8 GPUs -> 2 per GPU.  Synthetic clips, random
weights; prints ms per step, frames / s and the peak memory."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from common import TRAIN_CFG, synth_model_state
from networks.engines import build_engine
from utils.ema import ExponentialMovingAverage, get_param_buffer_for_ema
from utils.learning import get_trainable_params
from utils.optim import AdamW
from utils.synth import synth_clip
name = sys.argv[1] if len(sys.argv) > 1 else 'r50_deaotl'
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 5
S = int(sys.argv[4]) if len(sys.argv) > 4 else 465
cfg, model, _ = synth_model_state(name, cfg_overrides=TRAIN_CFG)
if not cfg.MODEL_ALIGN_CORNERS:
    S = S // 16 * 16
model = model.cuda().train()
engine = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP).train()
frames, masks, objs = [], [], []
for b in range(bs):
    f, m, o, _ = synth_clip(40 + b, T, (S, S), (S, S), 3 + b, device='cuda')
    fr = torch.cat(f, 0) if isinstance(f, (list, tuple)) else f
    frames.append(fr)
    # (the label of every frame: the first-frame label -- a timing run only needs plausible maps)
    masks.append(m.expand(T, -1, -1, -1))
    objs.append(3 + b)
all_frames = torch.stack(frames, 1).reshape(T * bs, 3, S, S).contiguous()
all_masks = torch.stack(masks, 1).reshape(T * bs, 1, S, S).contiguous().float()
opt = AdamW(get_trainable_params(model, 2e-4, 0.07, use_frozen_bn=cfg.MODEL_FREEZE_BN, no_wd_keys=['relative_emb_v', 'conv_out']), lr=2e-4,
            weight_decay=0.07)
ema_params = get_param_buffer_for_ema(model, update_buffer=False)
ema = ExponentialMovingAverage(ema_params, decay=0.99)


def step(i):
    engine.restart_engine(bs, True)
    opt.zero_grad()
    loss = engine(all_frames, all_masks, bs, objs, step=i)[0]
    loss.backward()
    _, scale = opt.clip_grad_norm(5.0)
    opt.step(grad_scale=scale)
    ema.update(ema_params)
    return float(loss.detach())


l0 = step(0)
torch.cuda.synchronize()
t0 = time.time()
ls = [step(i + 1) for i in range(3)]
torch.cuda.synchronize()
dt = (time.time() - t0) / 3
print('%s, batch %d x %d frames at %dx%d: %.0f ms per training step = %.1f frames/s on one MI355X; peak memory %.1f GiB; losses %.3f -> %s'
      % (name, bs, T, S, S, dt * 1e3, bs * T / dt, torch.cuda.max_memory_allocated() / 2**30, l0, ' '.join('%.3f' % v for v in ls)))
