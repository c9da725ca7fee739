"""The training step of the reference's trainer (networks/managers/trainer.py:356-519: per-step schedule -> forward ->
backward -> clip -> AdamW -> EMA, under DistributedDataParallel :59-74) as one object -- `TrainStep` -- over the flat training
state of utils/flat_state.py.  What is NOT here: data loading, logging, checkpoint cadence (the loop around the step).

One process per GPU; `torch.distributed` (`nccl` = RCCL over xGMI) carries the gradient buckets, issued from the
post-accumulate-grad hooks while backward runs.  `precision='bf16'` runs the conv / linear products of forward, dgrad and wgrad
on the bf16 matrix cores (operands rounded to bf16 in the kernel, fp32 accumulation, fp32 master weights and optimiser state; no
loss scale: bf16 has fp32's exponent range) -- what `--amp` / BASELINE config 5 ask of the reference (trainer.py:123-125,
460-487, there fp16 autocast + GradScaler).
"""
import torch

from utils.flat_state import FlatTrainState
from utils.learning import adjust_learning_rate, get_trainable_params


class TrainStep:
    def __init__(self, cfg, model, engine, group=None, precision='f32', bucket_mb=32.0, ema=True):
        if precision not in ('f32', 'bf16'):
            raise ValueError("precision is 'f32' or 'bf16'")
        self.cfg, self.model, self.engine, self.precision = cfg, model, engine, precision
        use_frozen_bn = cfg.MODEL_FREEZE_BN and 'swin' not in cfg.MODEL_ENCODER          # trainer.py:79-90
        groups = get_trainable_params(model, base_lr=cfg.TRAIN_LR, weight_decay=cfg.TRAIN_WEIGHT_DECAY, use_frozen_bn=use_frozen_bn,
                                      exclusive_wd_dict=cfg.TRAIN_WEIGHT_DECAY_EXCLUSIVE, no_wd_keys=cfg.TRAIN_WEIGHT_DECAY_EXEMPTION)
        total = float(cfg.TRAIN_TOTAL_STEPS)
        self.state = FlatTrainState(groups, bucket_mb=bucket_mb, group=group, ema=ema,
                                    ema_decay=1. - 1. / (total * cfg.TRAIN_EMA_RATIO))              # trainer.py:94-98
        self.start_seq = int(cfg.TRAIN_SEQ_TRAINING_START_RATIO * cfg.TRAIN_TOTAL_STEPS)
        self.lr = None

    def schedule(self, step):
        """trainer.py:407-428: prediction feedback (and the frozen identity bank) in the second part of the recipe, the LR of the step."""
        cfg = self.cfg
        use_prev_pred = step >= self.start_seq
        if step % cfg.TRAIN_LR_UPDATE_STEP == 0 or self.lr is None:
            self.lr = adjust_learning_rate(self.state, base_lr=cfg.TRAIN_LR, p=cfg.TRAIN_LR_POWER, itr=step, max_itr=cfg.TRAIN_TOTAL_STEPS,
                                           restart=cfg.TRAIN_LR_RESTART, warm_up_steps=cfg.TRAIN_LR_WARM_UP_RATIO * cfg.TRAIN_TOTAL_STEPS,
                                           is_cosine_decay=cfg.TRAIN_LR_COSINE_DECAY, min_lr=cfg.TRAIN_LR_MIN,
                                           encoder_lr_ratio=cfg.TRAIN_LR_ENCODER_RATIO,
                                           freeze_params=cfg.TRAIN_SEQ_TRAINING_FREEZE_PARAMS if use_prev_pred else [])
        return use_prev_pred

    def __call__(self, all_frames, all_masks, obj_nums, step, enable_prev_frame=None):
        """One step on this rank's share of the batch (time-major [T * bs, ...], trainer.py:452-455).  Returns the loss tensor
        (device; reading it is the caller's synchronisation) and the per-frame masks / losses."""
        from networks.layers import train_ops
        cfg = self.cfg
        use_prev_pred = self.schedule(step)
        bs = len(obj_nums)
        self.engine.restart_engine(bs, True)
        self.state.zero_grad()
        with train_ops.matmul_precision(self.precision), train_ops.weight_cache():
            loss, masks, losses, _ = self.engine(all_frames, all_masks, bs, obj_nums, step=step, use_prev_pred=use_prev_pred,
                                                 enable_prev_frame=cfg.TRAIN_ENABLE_PREV_FRAME if enable_prev_frame is None
                                                 else enable_prev_frame, use_prev_prob=cfg.MODEL_USE_PREV_PROB)
            loss = torch.mean(loss)
            loss.backward()                    # gradient buckets leave while this runs
        self.state.average()
        self.state.step(max_norm=cfg.TRAIN_CLIP_GRAD_NORM)
        return loss.detach(), masks, losses
