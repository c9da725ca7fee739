"""Gradient averaging across the data-parallel ranks (what DistributedDataParallel does for the reference's trainer,
trainer.py:59-74): gradients are packed into a few large buckets and every bucket is ONE all_reduce -- xGMI rings are bound per
link, so few large collectives beat one per tensor -- issued asynchronously in reverse parameter order (the order backward
produces them) and unpacked after the last wait.  Backend-agnostic (`nccl` = RCCL on the GPU box, `gloo` in the CPU tests)."""
import torch
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, params, bucket_mb=64.0, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > cap:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def average(self):
        """All-reduces every gradient (missing ones count as zeros) and divides by the world size, in place."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        works = []
        for i, bucket in enumerate(self.buckets):
            dev, dt = bucket[0].device, bucket[0].dtype
            total = sum(p.numel() for p in bucket)
            if self._flat[i] is None or self._flat[i].numel() != total or self._flat[i].device != dev:
                self._flat[i] = torch.empty(total, dtype=dt, device=dev)
            flat, o = self._flat[i], 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[o:o + n].zero_()
                else:
                    flat[o:o + n].copy_(p.grad.reshape(-1))
                o += n
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for i, bucket in enumerate(self.buckets):
            works[i].wait()
            flat, o = self._flat[i], 0
            flat.div_(world)
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[o:o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
