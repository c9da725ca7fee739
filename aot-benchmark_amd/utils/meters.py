"""Running statistics of the trainer's log lines (reference utils/meters.py:4-33)."""


class AverageMeter:
    """Last value, mean since reset(), and a moving average whose momentum warms up as 1 - 1/n to `momentum` (the count it
    warms up on survives reset())."""

    def __init__(self, momentum=0.999):
        self.momentum = momentum
        self.long_count = 0
        self.moving_avg = 0
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        if self.long_count == 0:
            self.moving_avg = val
        else:
            m = min(self.momentum, 1. - 1. / self.long_count)
            self.moving_avg = self.moving_avg * m + val * (1 - m)
        self.val = val
        self.sum += val * n
        self.count += n
        self.long_count += n
        self.avg = self.sum / self.count
