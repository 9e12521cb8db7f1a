"""Seeded random-init weights and inputs (no checkpoints or datasets ship with the reference).

Weights are drawn with NumPy's ``RandomState`` (bit-stable across NumPy/PyTorch
versions) from the same *distributions* the reference's constructors use
(``nn.Linear`` default Kaiming-uniform, LayerNorm (1, 0), Xavier-uniform for the
DiGCN weights, ``digcn_meta.py:35-37``), so fixtures only need to record a seed.

``preset='contractive'`` scales the score network's last layer so that the learned
field is small against the closed-form drift; free-running trajectories are then
well conditioned and end-to-end discretised architectures can be compared bit for
bit ("Hard parts").
"""

from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from .specs import (CateDims, PredictorCateDims, PredictorDims, Spec, cate_param_spec,
                    predictor_cate_param_spec, predictor_param_spec)


def _draw(rs: np.random.RandomState, shape, init) -> np.ndarray:
    kind = init[0]
    if kind in ("lin_w", "lin_b"):
        b = 1.0 / math.sqrt(init[1])
        return rs.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "ones":
        return np.ones(shape, np.float32)
    if kind == "zeros":
        return np.zeros(shape, np.float32)
    if kind == "xavier":
        a = math.sqrt(6.0 / (init[1] + init[2]))
        return rs.uniform(-a, a, size=shape).astype(np.float32)
    raise ValueError(kind)


def state_dict_from_spec(spec: Spec, seed: int) -> "OrderedDict[str, torch.Tensor]":
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, shape, init in spec:
        sd[name] = torch.from_numpy(_draw(rs, shape, init))
    return sd


def cate_state_dict(dims: CateDims = CateDims(), seed: int = 0, preset: str = "default",
                    perturb_norms: bool = False):
    """Random-init CATE weights.  ``perturb_norms`` makes LayerNorm affine non-trivial
    (a trained checkpoint would be) so that gamma/beta indexing bugs cannot hide."""
    sd = state_dict_from_spec(cate_param_spec(dims), seed)
    if perturb_norms:
        rs = np.random.RandomState(seed + 7919)
        for k in sd:
            if ".norm" in k or ".norms." in k:
                if k.endswith(".weight"):
                    sd[k] = torch.from_numpy((1.0 + 0.2 * rs.uniform(-1, 1, sd[k].shape)).astype(np.float32))
                else:
                    sd[k] = torch.from_numpy((0.1 * rs.uniform(-1, 1, sd[k].shape)).astype(np.float32))
    if preset == "contractive":
        sd["final.linears.2.weight"] = sd["final.linears.2.weight"] * 0.05
        sd["final.linears.2.bias"] = sd["final.linears.2.bias"] * 0.05
    elif preset != "default":
        raise ValueError(preset)
    return sd


def predictor_state_dict(dims: PredictorDims = PredictorDims(), seed: int = 1):
    return state_dict_from_spec(predictor_param_spec(dims), seed)


def predictor_cate_state_dict(dims: PredictorCateDims = PredictorCateDims(), seed: int = 5, perturb_norms: bool = True):
    """Random-init MetaPredictorCATE (OFA surrogate) weights; LayerNorm affine perturbed by default."""
    sd = state_dict_from_spec(predictor_cate_param_spec(dims), seed)
    if perturb_norms:
        rs = np.random.RandomState(seed + 7919)
        for k in sd:
            if ".norm" in k or ".norms." in k:
                if k.endswith(".weight"):
                    sd[k] = torch.from_numpy((1.0 + 0.2 * rs.uniform(-1, 1, sd[k].shape)).astype(np.float32))
                else:
                    sd[k] = torch.from_numpy((0.1 * rs.uniform(-1, 1, sd[k].shape)).astype(np.float32))
    return sd


def nb201_adjacency() -> torch.Tensor:
    """The fixed 8-node NAS-Bench-201 cell DAG (values of reference ``datasets_nas.py:183-192``)."""
    edges = [(0, 1), (0, 2), (0, 3), (1, 4), (1, 5), (2, 6), (3, 7), (4, 6), (5, 7), (6, 7)]
    a = torch.zeros(8, 8)
    for i, j in edges:
        a[i, j] = 1.0
    return a


def ofa_adjacency(n: int = 20) -> torch.Tensor:
    """OFA chain DAG: node i feeds node i+1 (reference ``MBV3/main_exp/diffusion/run_lib.py:147-153``)."""
    a = torch.zeros(n, n)
    for i in range(n - 1):
        a[i, i + 1] = 1.0
    return a


def transitive_closure(adj: torch.Tensor) -> torch.Tensor:
    """Floyd-Warshall reachability, the ``algo='floyed'`` mask of reference ``utils.py:123-139``."""
    r = adj.clone().numpy()
    n = r.shape[0]
    for k in range(n):
        for i in range(n):
            for j in range(n):
                if r[i, k] > 0 and r[k, j] > 0:
                    r[i, j] = 1
    return torch.from_numpy(r)


def synthetic_task(num_class: int = 10, num_sample: int = 20, seed: int = 2) -> torch.Tensor:
    """A stand-in dataset-encoding task ``[num_class*num_sample, 512]``."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((num_class * num_sample, 512)).astype(np.float32))


def noise_tape(n_steps: int, shape, seed: int) -> dict:
    """x0 and the per-step (corrector, predictor) noise in reference RNG order."""
    rs = np.random.RandomState(seed)
    x0 = rs.standard_normal(shape).astype(np.float32)
    eps = rs.standard_normal((n_steps,) + tuple(shape)).astype(np.float32)
    z = rs.standard_normal((n_steps,) + tuple(shape)).astype(np.float32)
    return {"x0": torch.from_numpy(x0), "corrector": torch.from_numpy(eps), "predictor": torch.from_numpy(z)}
