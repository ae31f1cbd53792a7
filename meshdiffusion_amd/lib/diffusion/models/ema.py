"""Exponential moving average of parameters -- mirror of the reference's
lib/diffusion/models/ema.py (update :32-51, copy_to :53-64, store/restore :66-89,
state_dict :91-97).  Same state-dict format {decay, num_updates, shadow_params}.
"""
import torch

from .... import hip_ops as ops


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def _effective_decay(self):
        d = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            d = min(d, (1 + self.num_updates) / (10 + self.num_updates))
        return d

    def update(self, parameters):
        one_minus = 1.0 - self._effective_decay()
        live = [p for p in parameters if p.requires_grad]
        with torch.no_grad():
            if live and live[0].is_cuda:
                # s -= (1-d) * (s - p), batched over all tensors (one multi-tensor launch each)
                diff = torch._foreach_sub(self.shadow_params, live)
                torch._foreach_mul_(diff, one_minus)
                torch._foreach_sub_(self.shadow_params, diff)
            else:
                for s, p in zip(self.shadow_params, live):
                    s.sub_(one_minus * (s - p))

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, [q for q in parameters if q.requires_grad]):
            p.data.copy_(s.data)
        # `p.data.copy_` does not bump `p._version`: the packed-weight caches of the HIP layers (keyed on
        # data_ptr + _version + PARAM_EPOCH) would keep serving the previous weights
        ops.bump_param_epoch()

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)
        ops.bump_param_epoch()

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        self.shadow_params = state_dict["shadow_params"]
