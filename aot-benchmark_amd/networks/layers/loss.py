"""Segmentation losses of the training path (reference networks/layers/loss.py:119-188) on the device.

Same classes, constructor arguments and call contract as the reference -- ``loss(list of [1, obj+1, H, W] logits, list of
[1, H, W] labels, step) -> [len] tensor`` (aot_engine.py:398-419) -- with forward and backward as HIP kernels
(csrc/train_ops.hip): per-pixel cross entropy, radix select of the hard-example set, fp64 class sums for the Jaccard ratio.
The samples of a batch carry different numbers of objects, hence one launch group per sample, as in the reference's loop.
(``SoftJaccordLoss`` keeps the reference's spelling.)"""
import torch
from torch import nn

import aot_hip


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, top_k):
        logits = logits.contiguous().float()
        label = label.reshape(logits.shape[0], -1).float().contiguous()
        loss, saved = aot_hip.ce_loss(logits, label, top_k)
        ctx.save_for_backward(logits, label, saved[0], *([saved[1]] if saved[1] is not None else [saved[2]]))
        ctx.top_k = top_k
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, label, loss_px, extra = ctx.saved_tensors
        if ctx.top_k > 0:
            gscale = (gout.float() / float(ctx.top_k)).contiguous()
            grad = aot_hip.ce_loss_bwd(logits, label, (loss_px, extra, None), gscale)
        else:
            gscale = (gout.float() / extra).contiguous()
            grad = aot_hip.ce_loss_bwd(logits, label, (loss_px, None, extra), gscale)
        return grad, None, None


class _JaccardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, eps):
        logits = logits.contiguous().float()
        label = label.reshape(logits.shape[0], -1).float().contiguous()
        loss, sums = aot_hip.soft_jaccard(logits, label, eps)
        ctx.save_for_backward(logits, label, sums)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, label, sums = ctx.saved_tensors
        return aot_hip.soft_jaccard_bwd(logits, label, sums, gout.float().contiguous(), ctx.eps), None, None


class CrossEntropyLoss(nn.Module):
    """reference loss.py:137-188: cross entropy, optionally over the hardest top-k share of the pixels only, the share
    annealed from all pixels to ``top_k_percent_pixels`` over ``hard_example_mining_step`` steps."""

    def __init__(self, top_k_percent_pixels=None, hard_example_mining_step=100000):
        super().__init__()
        self.top_k_percent_pixels = top_k_percent_pixels
        if top_k_percent_pixels is not None:
            assert 0 < top_k_percent_pixels < 1
        self.hard_example_mining_step = hard_example_mining_step + 1e-5

    def top_k_pixels(self, num_pixels, step):
        """How many pixels enter the mean at `step` (loss.py:171-177); 0 = all valid pixels."""
        if self.top_k_percent_pixels is None:
            return 0
        if self.hard_example_mining_step == 0:
            return int(self.top_k_percent_pixels * num_pixels)
        ratio = min(1.0, step / float(self.hard_example_mining_step))
        return int((ratio * self.top_k_percent_pixels + (1.0 - ratio)) * num_pixels)

    def forward(self, dic_tmp, y, step):
        total = []
        for pred_logits, gts in zip(dic_tmp, y):
            k = self.top_k_pixels(float(pred_logits.size(2) * pred_logits.size(3)), step)
            total.append(_CEFn.apply(pred_logits, gts, k))
        return torch.cat(total, dim=0)


class SoftJaccordLoss(nn.Module):
    """reference loss.py:119-137: 1 - soft IoU per class present in the sample (tversky_loss with alpha = beta = 1), averaged."""

    def __init__(self, ignore_index=255):
        super().__init__()
        if ignore_index != 255:
            raise NotImplementedError('the kernels take 255 as the ignore label (the only value the reference uses)')
        self.ignore_index = ignore_index

    def forward(self, tmp_dic, label_dic, step=None):
        return torch.cat([_JaccardFn.apply(pred, label, 1e-6) for pred, label in zip(tmp_dic, label_dic)], dim=0)
