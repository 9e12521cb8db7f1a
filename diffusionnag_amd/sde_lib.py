"""Host mirror of the reference's ``sde_lib`` for the sampling path.

Only scalar schedules live here; every per-element operation of the discretisation
(``f = sqrt(alpha) x - x``, ``rev_f = f - G^2 score``, reference ``sde_lib.py:119-125``,
``:198-211``) runs inside the HIP exit kernel.  The tables are built with torch CPU fp32
ops in the same order as the reference so that the kernel consumes bit-identical scalars.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _lib


class VPSDE:
    """Variance-preserving SDE, reference ``sde_lib.py:145-211`` (same constructor and attributes)."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.N = N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)      # :158
        self.alphas = 1. - self.discrete_betas                                   # :159
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def marginal_prob(self, x, t):
        """``sde_lib.py:179-188``."""
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_mean_coeff[:, None, None]) * x
        std = torch.sqrt(1. - torch.exp(2. * log_mean_coeff))
        return mean, std

    def prior_sampling(self, shape):
        """``sde_lib.py:190-191``: drawn on the CPU generator, like the reference."""
        return torch.randn(*shape)

    # ---------------------------------------------------------------------------------------
    def schedule_table(self, eps: float, n_steps: int | None = None) -> np.ndarray:
        """Per-step scalars of the PC loop as a host ``[n_steps][8]`` fp32 table (``include/dnag.h``).

        Row i holds, for ``t_i = linspace(T, eps, N)[i]`` (``sampling.py:437``):
        ``999 t`` (``models/utils.py:93``), the marginal std, ``alphas[k]``, ``sqrt(alphas[k])``,
        ``G = sqrt(discrete_betas[k])``, ``G*G`` (``sde_lib.py:122`` squares the rounded root) and
        ``t``, with ``k = (t (N-1) / T).long()`` (``sde_lib.py:200``, ``sampling.py:238``).
        """
        ts = torch.linspace(self.T, eps, self.N)
        S = self.N if n_steps is None else int(n_steps)
        if not 1 <= S <= self.N:
            raise ValueError(f"n_steps={S} outside [1, {self.N}]")
        ts = ts[:S]
        k = (ts * (self.N - 1) / self.T).long()
        beta = self.discrete_betas[k]
        alpha = self.alphas[k]
        G = torch.sqrt(beta)
        _, std = self.marginal_prob(torch.zeros(S, 1, 1), ts)
        tab = torch.zeros(S, _lib.SCHED_COLS, dtype=torch.float32)
        tab[:, _lib.SCHED_LABEL] = ts * 999
        tab[:, _lib.SCHED_STD] = std
        tab[:, _lib.SCHED_ALPHA] = alpha
        tab[:, _lib.SCHED_SQRTA] = torch.sqrt(alpha)
        tab[:, _lib.SCHED_G] = G
        tab[:, _lib.SCHED_G2] = G ** 2
        tab[:, _lib.SCHED_T] = ts
        return np.ascontiguousarray(tab.numpy())


class VESDE:
    """Placeholder so that ``isinstance`` dispatch mirrors the reference; sampling with it is a
    "next" row and raises ``NotImplementedError`` in ``get_sampling_fn``."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        self.sigma_min, self.sigma_max, self.N = sigma_min, sigma_max, N

    @property
    def T(self):
        return 1
