"""Host-side EMA of the parameters, the object ``step_fn`` keeps in ``state['ema']`` (reference
``models/ema.py:4-85``: same constructor, methods, ``state_dict`` keys and decay warm-up).  The update runs
as two multi-tensor (``torch._foreach``) launches over all ~200 parameter tensors instead of one small
launch per tensor; it is plain PyTorch plumbing, not part of the HIP path."""

from __future__ import annotations

import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def update(self, parameters):
        """shadow -= (1 - decay) (shadow - param), decay = min(decay, (1 + n) / (10 + n))  (ema.py:34-42)."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            live = [p.detach() for p in parameters if p.requires_grad]
            diff = torch._foreach_sub(self.shadow_params, live)
            torch._foreach_mul_(diff, 1.0 - decay)          # same operation order as the reference: bit-identical
            torch._foreach_sub_(self.shadow_params, diff)

    def copy_to(self, parameters):
        with torch.no_grad():
            torch._foreach_copy_([p.data for p in parameters if p.requires_grad], self.shadow_params)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            torch._foreach_copy_([p.data for p in parameters], [c.data for c in self.collected_params])

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        self.shadow_params = state_dict['shadow_params']
