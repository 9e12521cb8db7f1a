"""EMA of the parameters, the object ``step_fn`` keeps in ``state['ema']`` (reference ``models/ema.py:4-85``: same
constructor, methods, ``state_dict`` keys and decay warm-up).  The shadow parameters are views of one flat buffer; on
the GPU the update is one pass of ``dnag_optim_step`` (``include/dnag.h``) with the reference's operation order
(``s -= (1 - d) (s - p)``, no fused multiply-add: bit-identical to ``models/ema.py:39-42`` for the same parameters),
either on its own (``update``) or folded into the optimiser's pass (``losses.FusedAdam.step(ema=...)``)."""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .flat import chain_view, flatten_, FlatCache


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def next_one_minus_decay(self) -> float:
        """Advance ``num_updates`` and return ``1 - decay`` of this update, in Python doubles (ema.py:34-38)."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return 1.0 - decay

    def flat_shadow(self, like=None) -> torch.Tensor:
        """The shadow parameters as one flat buffer (re-seated as views of it on first use)."""
        if like is not None and self.shadow_params and self.shadow_params[0].device != like[0].device:
            self.shadow_params = [s.to(like[0].device) for s in self.shadow_params]
        if getattr(self, "_flat_cache", None) is None:
            self._flat_cache = FlatCache()
        return self._flat_cache.get(self.shadow_params, flatten_)

    def update(self, parameters):
        """shadow -= (1 - decay) (shadow - param), decay = min(decay, (1 + n) / (10 + n))  (ema.py:24-42)."""
        live = [p for p in parameters if p.requires_grad]
        omd = self.next_one_minus_decay()
        flat = chain_view([p.data for p in live])
        if flat is not None and flat.is_cuda:
            lib = _lib.load()
            a = _lib.OptimArgsC()
            a.n, a.flags = flat.numel(), _lib.OPTIM_EMA
            shadow = self.flat_shadow(live)
            if shadow.numel() != flat.numel():
                raise RuntimeError(f"EMA shadow holds {shadow.numel()} values, the parameters {flat.numel()}")
            a.params, a.shadow, a.ema_one_minus_decay = _lib.ptr(flat), _lib.ptr(shadow), omd
            a.stream = _lib.current_stream(flat.device)
            with torch.cuda.device(flat.device):
                _lib.check(lib.dnag_optim_step(C.byref(a)), "dnag_optim_step")
            torch.autograd.graph.increment_version(self.shadow_params)     # raw-pointer write
            return
        with torch.no_grad():        # parameters that are not one flat device buffer (host-side use): multi-tensor plumbing
            diff = torch._foreach_sub(self.shadow_params, [p.detach() for p in live])
            torch._foreach_mul_(diff, omd)                  # same operation order as the reference: bit-identical
            torch._foreach_sub_(self.shadow_params, diff)

    def copy_to(self, parameters):
        # in place on the parameters themselves (not on .data): their versions change, so the packed kernel-layout
        # weights of the model are rebuilt before the next forward / sampling call
        with torch.no_grad():
            torch._foreach_copy_([p for p in parameters if p.requires_grad], self.shadow_params)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            torch._foreach_copy_(list(parameters), [c.data for c in self.collected_params])

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        self.shadow_params = state_dict['shadow_params']
