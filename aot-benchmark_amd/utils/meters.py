"""Running statistics of the trainer's log lines (reference utils/meters.py:4-33)."""


class AverageMeter:
    """Three views of a logged scalar: `val` the last sample, `avg` the weighted mean since reset(), `moving_avg` an
    exponential average whose momentum warms up as 1 - 1/n towards `momentum` -- n (`long_count`) counts every sample ever
    seen, reset() does not touch it."""

    def __init__(self, momentum=0.999):
        self.momentum, self.long_count, self.moving_avg = momentum, 0, 0
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        seen = self.long_count
        keep = 0.0 if seen == 0 else min(self.momentum, 1.0 - 1.0 / seen)      # the first sample replaces the initial 0
        self.moving_avg = val if seen == 0 else keep * self.moving_avg + (1 - keep) * val
        self.val, self.sum, self.count, self.long_count = val, self.sum + val * n, self.count + n, seen + n
        self.avg = self.sum / self.count
