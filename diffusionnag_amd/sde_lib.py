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

    def sde(self, x, t):
        """``sde_lib.py:168-177``."""
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        drift = -0.5 * beta_t[:, None, None] * x
        diffusion = torch.sqrt(beta_t)
        return drift, diffusion

    score_mode = _lib.SCORE_VP          # models/utils.py:87-104
    prior_scale = 1.0

    def time_cond(self, t):
        """Time conditioning of a continuously trained model, ``models/utils.py:93``."""
        return t * 999

    # ---------------------------------------------------------------------------------------
    def schedule_table(self, eps: float, n_steps: int | None = None, predictor: str = "reverse_diffusion") -> np.ndarray:
        """Per-step scalars of the PC loop as a host ``[n_steps][8]`` fp32 table (``include/dnag.h``).

        Row i holds, for ``t_i = linspace(T, eps, N)[i]`` (``sampling.py:437``):
        ``999 t`` (``models/utils.py:93``), the marginal std, ``alphas[k]`` and ``t``, with
        ``k = (t (N-1) / T).long()`` (``sde_lib.py:200``, ``sampling.py:238``); for the reverse-diffusion
        predictor ``sqrt(alphas[k])``, ``G = sqrt(discrete_betas[k])``, ``G*G`` (``sde_lib.py:122`` squares
        the rounded root); for Euler-Maruyama ``-0.5 beta(t)``, ``diffusion * sqrt(-dt)``,
        ``diffusion ** 2`` and ``dt = -1/N`` (``sampling.py:194-200``, ``sde_lib.py:95-103``, ``:168-177``).
        """
        ts, S = _time_grid(self, eps, n_steps)
        k = (ts * (self.N - 1) / self.T).long()
        alpha = self.alphas[k]
        _, std = self.marginal_prob(torch.zeros(S, 1, 1), ts)
        tab = torch.zeros(S, _lib.SCHED_COLS, dtype=torch.float32)
        tab[:, _lib.SCHED_LABEL] = self.time_cond(ts)
        tab[:, _lib.SCHED_STD] = std
        tab[:, _lib.SCHED_ALPHA] = alpha
        tab[:, _lib.SCHED_T] = ts
        if predictor == "euler_maruyama":
            _fill_euler_maruyama(self, tab, ts)
        else:
            G = torch.sqrt(self.discrete_betas[k])
            tab[:, _lib.SCHED_SQRTA] = torch.sqrt(alpha)
            tab[:, _lib.SCHED_G] = G
            tab[:, _lib.SCHED_G2] = G ** 2
        return np.ascontiguousarray(tab.numpy())


def _time_grid(sde, eps, n_steps):
    ts = torch.linspace(sde.T, eps, sde.N)                                          # sampling.py:437
    S = sde.N if n_steps is None else int(n_steps)
    if not 1 <= S <= sde.N:
        raise ValueError(f"n_steps={S} outside [1, {sde.N}]")
    return ts[:S], S


def _fill_euler_maruyama(sde, tab, ts):
    """Scalars of ``EulerMaruyamaPredictor.update_fn`` (``sampling.py:194-200``) on ``RSDE.sde``
    (``sde_lib.py:95-103``), computed with the reference's own torch expressions: the drift of both
    SDEs is ``c(t) x`` (``c = 0`` for the VESDE)."""
    dt = -1. / sde.N
    ones = torch.ones(ts.shape[0], 1, 1)
    drift, diffusion = sde.sde(ones, ts)
    tab[:, _lib.SCHED_SQRTA] = drift.reshape(-1)                # c(t) * 1
    tab[:, _lib.SCHED_G] = (diffusion[:, None, None] * np.sqrt(-dt)).reshape(-1)
    tab[:, _lib.SCHED_G2] = diffusion ** 2
    tab[:, _lib.SCHED_DT] = (ones * dt).reshape(-1)


class VESDE:
    """Variance-exploding SDE, reference ``sde_lib.py:253-300`` (same constructor and attributes)."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))   # :265
        self.N = N

    @property
    def T(self):
        return 1

    def sde(self, x, t):
        """``sde_lib.py:272-277``."""
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        drift = torch.zeros_like(x)
        diffusion = sigma * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))))
        return drift, diffusion

    def marginal_prob(self, x, t):
        """``sde_lib.py:279-282``."""
        std = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return x, std

    def prior_sampling(self, shape):
        """``sde_lib.py:284-285``."""
        return torch.randn(*shape) * self.sigma_max

    score_mode = _lib.SCORE_VE          # models/utils.py:106-117: the model output is the score

    @property
    def prior_scale(self):
        return float(self.sigma_max)

    def time_cond(self, t):
        """``models/utils.py:109``: the noise level itself conditions a continuously trained model."""
        return self.marginal_prob(torch.zeros(t.shape[0], 1, 1), t)[1]

    def schedule_table(self, eps: float, n_steps: int | None = None, predictor: str = "reverse_diffusion") -> np.ndarray:
        """As ``VPSDE.schedule_table``: ``sigma(t)`` as time conditioning, Langevin ``alpha = 1``
        (``sampling.py:241-242``); SMLD discretisation ``f = 0``, ``G = sqrt(sigma_k^2 - sigma_{k-1}^2)``
        (``sde_lib.py:292-300``) for the reverse-diffusion predictor."""
        ts, S = _time_grid(self, eps, n_steps)
        tab = torch.zeros(S, _lib.SCHED_COLS, dtype=torch.float32)
        tab[:, _lib.SCHED_LABEL] = self.time_cond(ts)
        tab[:, _lib.SCHED_STD] = 1.0
        tab[:, _lib.SCHED_ALPHA] = torch.ones_like(ts)
        tab[:, _lib.SCHED_T] = ts
        if predictor == "euler_maruyama":
            _fill_euler_maruyama(self, tab, ts)
        else:
            k = (ts * (self.N - 1) / self.T).long()
            sigma = self.discrete_sigmas[k]
            adjacent = torch.where(k == 0, torch.zeros_like(ts), self.discrete_sigmas[k - 1])
            G = torch.sqrt(sigma ** 2 - adjacent ** 2)
            tab[:, _lib.SCHED_SQRTA] = 1.0
            tab[:, _lib.SCHED_G] = G
            tab[:, _lib.SCHED_G2] = G ** 2
        return np.ascontiguousarray(tab.numpy())
