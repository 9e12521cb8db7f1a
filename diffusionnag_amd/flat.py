"""Flat fp32 views of parameter lists.  The training step (``dnag_cate_dsm_step``) and the optimiser side
(``dnag_optim_step``) work on ONE contiguous buffer in ``state_dict()`` order; the ``nn.Parameter`` objects the
reference's code iterates over (``model.parameters()``, ``ema.shadow_params``, ``p.grad``) are kept as views into it."""

from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def chain_view(tensors: Sequence[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
    """If the tensors are contiguous fp32 pieces lying back to back in one storage, the flat ``[n]`` view over all of
    them; else None."""
    if not tensors or any(t is None for t in tensors):
        return None
    first = tensors[0]
    if first.dtype != torch.float32:
        return None
    ptr = first.data_ptr()
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != ptr or t.device != first.device:
            return None
        ptr += 4 * t.numel()
    n = (ptr - first.data_ptr()) // 4
    if first.untyped_storage().data_ptr() + first.untyped_storage().nbytes() < ptr:
        return None
    return torch.as_strided(first.detach(), (n,), (1,))


def flatten_(tensors: List[torch.Tensor]) -> torch.Tensor:
    """Re-seat the tensors' ``.data`` as views of one new flat buffer (values kept) and return it.  A no-op returning
    the existing flat view when they already are."""
    flat = chain_view([t.data for t in tensors])
    if flat is not None:
        return flat
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors]).contiguous()
    o = 0
    for t in tensors:
        n = t.numel()
        t.data = flat[o:o + n].view(t.shape)
        o += n
    return flat


class FlatCache:
    """Remembers the flat view of a fixed list of tensors and re-validates it with two pointer comparisons (first tensor at
    the buffer's start, last tensor ending at its end) instead of a walk over every tensor: the training step asks for
    the same views four times per step, and the walk was a quarter of the host's time per step."""

    def __init__(self):
        self.flat = None
        self.first = self.last = None

    def get(self, tensors: List[torch.Tensor], make=chain_view) -> Optional[torch.Tensor]:
        f = self.flat
        if f is not None and tensors and tensors[0] is self.first and tensors[-1] is self.last:
            a, b = tensors[0].data, tensors[-1].data
            if a.data_ptr() == f.data_ptr() and b.data_ptr() + 4 * b.numel() == f.data_ptr() + 4 * f.numel():
                return f
        f = make(tensors if make is flatten_ else [t.data for t in tensors])
        self.flat = f
        self.first, self.last = (tensors[0], tensors[-1]) if (f is not None and tensors) else (None, None)
        return f
