"""Exponential moving average of the parameters (reference utils/ema.py) with the update as a device kernel."""
import torch

import aot_hip


def get_param_buffer_for_ema(model, update_buffer=False, required_buffers=('running_mean', 'running_var')):
    out = [p for p in model.parameters() if p.requires_grad]
    if update_buffer:
        out += [b for k, b in model.named_buffers() if any(r in k for r in required_buffers)]
    return out


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.collected_params = []

    def current_decay(self):
        """Decay of the NEXT update: min(decay, (1 + n) / (10 + n)) with n counting from 1 (utils/ema.py:57-62)."""
        if self.num_updates is None:
            return self.decay
        n = self.num_updates + 1
        return min(self.decay, (1 + n) / (10 + n))

    def update(self, parameters):
        decay = self.current_decay()
        if self.num_updates is not None:
            self.num_updates += 1
        omd = 1.0 - decay
        with torch.no_grad():
            for s, p in zip(self.shadow_params, parameters):
                aot_hip.ema_update(s, p.detach().contiguous(), omd)

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, parameters):
            p.data.copy_(s.data)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)
        del self.collected_params
