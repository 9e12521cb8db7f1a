"""CPU oracle for the DiffusionNAG reverse-diffusion sampling hot path.

TEST INFRASTRUCTURE ONLY.  This file is a functional, PyTorch-CPU (fp32) restatement
of the reference algorithm; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``diffusionnag_amd``) never does: it calls the HIP library through the C ABI and
fails loudly when that library is missing.

Pinning: the reference has no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, captured in this
container by ``tests/golden/make_golden.py`` (which imports
``/root/reference/NAS-Bench-201``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against them.

Every function cites the reference lines it restates (paths relative to
``/root/reference/NAS-Bench-201/``).  Weights are passed as plain ``state_dict``
mappings (reference parameter names), no ``nn.Module`` is involved.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# masks and scalers
# ----------------------------------------------------------------------------------------------

def nb201_adj() -> Tensor:
    """Fixed NAS-Bench-201 cell adjacency, datasets_nas.py:182-194."""
    rows = ["01110000", "00001100", "00000010", "00000001",
            "00000010", "00000001", "00000001", "00000000"]
    return torch.tensor([[float(c) for c in r] for r in rows])


def floyed(adj: Tensor) -> Tensor:
    """Reachability closure, utils.py:123-139 (the branch taken for ``algo='floyed'``)."""
    r = adj.clone()
    n = r.shape[0]
    for k in range(n):
        for i in range(n):
            for j in range(n):
                if r[i, k] > 0 and r[k, j] > 0:
                    r[i, j] = 1.0
    return r


def aug_mask(adj: Tensor, algo: str = "floyed") -> Tensor:
    """utils.py:142-156: only the literal string 'floyed' closes the mask; anything else
    (including the configs' 'floyd') returns the raw adjacency."""
    if adj.dim() == 3:
        adj = adj[0]
    return floyed(adj) if algo == "floyed" else adj.clone()


def inverse_scaler(x: Tensor) -> Tensor:
    """datasets_nas.py:21-28 with ``data.centered=True``."""
    return (x + 1.0) / 2.0


def quantize(x: Tensor) -> Tensor:
    """utils.py:244-262: threshold at 0.5 (returned as one tensor instead of a list of arrays)."""
    return torch.where(x >= 0.5, torch.ones_like(x), torch.zeros_like(x))


# ----------------------------------------------------------------------------------------------
# VPSDE (sde_lib.py:145-211)
# ----------------------------------------------------------------------------------------------

class VPSDE:
    """Scalar schedules of the variance-preserving SDE, sde_lib.py:146-162."""

    def __init__(self, beta_min: float = 0.1, beta_max: float = 20.0, N: int = 1000):
        self.beta_0, self.beta_1, self.N, self.T = beta_min, beta_max, N, 1
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)   # :158
        self.alphas = 1.0 - self.discrete_betas                               # :159

    def marginal_std(self, t: Tensor) -> Tensor:
        """sde_lib.py:179-188 (only the std is used while sampling)."""
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def timestep_index(self, t: Tensor) -> Tensor:
        """sde_lib.py:200 / sampling.py:238: truncating float -> int64."""
        return (t * (self.N - 1) / self.T).long()

    def discretize(self, x: Tensor, t: Tensor):
        """DDPM discretisation, sde_lib.py:198-211."""
        k = self.timestep_index(t)
        beta = self.discrete_betas[k]
        alpha = self.alphas[k]
        f = torch.sqrt(alpha)[:, None, None] * x - x
        G = torch.sqrt(beta)
        return f, G

    def sde(self, x: Tensor, t: Tensor):
        """sde_lib.py:168-177."""
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        return -0.5 * beta_t[:, None, None] * x, torch.sqrt(beta_t)

    def langevin_alpha(self, t: Tensor) -> Tensor:
        """sampling.py:237-240."""
        return self.alphas[self.timestep_index(t)]

    def time_cond(self, t: Tensor) -> Tensor:
        """models/utils.py:93 (continuous=True)."""
        return t * 999

    def prior_sampling(self, shape) -> Tensor:
        """sde_lib.py:190-191."""
        return torch.randn(*shape)


class VESDE:
    """Variance-exploding SDE, sde_lib.py:253-300."""

    def __init__(self, sigma_min: float = 0.01, sigma_max: float = 50.0, N: int = 1000):
        self.sigma_min, self.sigma_max, self.N, self.T = sigma_min, sigma_max, N, 1
        self.discrete_sigmas = torch.exp(torch.linspace(math.log(sigma_min), math.log(sigma_max), N))   # :265

    def marginal_std(self, t: Tensor) -> Tensor:
        """sde_lib.py:279-282."""
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def timestep_index(self, t: Tensor) -> Tensor:
        return (t * (self.N - 1) / self.T).long()

    def discretize(self, x: Tensor, t: Tensor):
        """SMLD discretisation, sde_lib.py:292-300."""
        k = self.timestep_index(t)
        sigma = self.discrete_sigmas[k]
        adjacent = torch.where(k == 0, torch.zeros_like(t), self.discrete_sigmas[k - 1])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent ** 2)

    def sde(self, x: Tensor, t: Tensor):
        """sde_lib.py:272-277."""
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        diffusion = sigma * torch.sqrt(torch.tensor(2 * (math.log(self.sigma_max) - math.log(self.sigma_min))))
        return torch.zeros_like(x), diffusion

    def langevin_alpha(self, t: Tensor) -> Tensor:
        """sampling.py:241-242."""
        return torch.ones_like(t)

    def time_cond(self, t: Tensor) -> Tensor:
        """models/utils.py:109 (continuous=True): the noise level."""
        return self.marginal_std(t)

    def prior_sampling(self, shape) -> Tensor:
        """sde_lib.py:284-285."""
        return torch.randn(*shape) * self.sigma_max


def timesteps(sde: VPSDE, eps: float) -> Tensor:
    """sampling.py:437."""
    return torch.linspace(sde.T, eps, sde.N)


# ----------------------------------------------------------------------------------------------
# CATE score network (models/cate.py:108-164, models/transformer.py)
# ----------------------------------------------------------------------------------------------

def timestep_embedding(labels: Tensor, dim: int, max_positions: int = 10000) -> Tensor:
    """cate.py:363-374 / digcn_meta.py:162-173."""
    half = dim // 2
    c = math.log(max_positions) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -c)
    arg = labels.float()[:, None] * freq[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def positional_table(max_node: int, d_model: int, stagewise: bool) -> Tensor:
    """cate.py:231-248 (Cell) and cate.py:211-228 (StageWise: table of max_node/5 rows tiled 5x)."""
    stages = 5 if stagewise else 1
    length = int(max_node / stages)
    enc = torch.zeros(length, d_model)
    pos = torch.arange(0, length).float().unsqueeze(1)
    two_i = torch.arange(0, d_model, step=2).float()
    enc[:, ::2] = torch.sin(pos / (10000 ** (two_i / d_model)))
    enc[:, 1::2] = torch.cos(pos / (10000 ** (two_i / d_model)))
    return torch.cat([enc] * stages, dim=0)


def _lin(sd: SD, prefix: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _ln(sd: SD, prefix: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def time_mlp(sd: SD, labels: Tensor, d_model: int) -> Tensor:
    """cate.py:155-157: timeEmb2(SiLU(timeEmb1(sincos(labels))))  (config nonlinearity 'swish')."""
    e = timestep_embedding(labels, d_model).to(sd["timeEmb1.weight"].dtype)     # fp32 in the reference; a no-op unless the
    return _lin(sd, "timeEmb2", F.silu(_lin(sd, "timeEmb1", e)))                # oracle is run in fp64 as a tie-breaker


def _drop(x: Tensor, dropout: Optional[dict], key: str) -> Tensor:
    """nn.Dropout in training mode with an injected keep mask: x * keep / (1 - p).  ``dropout`` is
    ``{'p': p, 'masks': {site: keep}}``; None (or p = 0) is the identity (eval mode)."""
    if dropout is None or dropout["p"] == 0.0:
        return x
    return x * dropout["masks"][key] / (1.0 - dropout["p"])


def dropout_sites(B: int, N: int, d: int = 64, n_head: int = 8, d_ff: int = 128, n_layers: int = 12) -> "OrderedDict":
    """Dropout sites of CATE in the order the reference's forward reaches them, with their shapes:
    cate.py:147 (dropout_op), then per layer transformer.py:19 (attention probabilities), :80
    (SelfAttentionBlock), :54 (FFN hidden), :108 (FeedForwardBlock)."""
    from collections import OrderedDict
    sites = OrderedDict(emb=(B, N, d))
    for l in range(n_layers):
        sites[f"attn{l}"] = (B, n_head, N, N)
        sites[f"sa{l}"] = (B, N, d)
        sites[f"ffh{l}"] = (B, N, d_ff)
        sites[f"ffo{l}"] = (B, N, d)
    return sites


def encoder_stack(sd: SD, prefix: str, emb: Tensor, mask: Tensor, n_layers: int, n_head: int,
                  record: Optional[dict] = None, dropout: Optional[dict] = None) -> List[Tensor]:
    """transformer.py:166-179 (Encoder), :127-136 (EncoderBlock), :69-80, :23-42, :12-20, :97-108."""
    B, N, d = emb.shape
    dk = d // n_head
    x = emb
    outs = []
    m4 = mask.unsqueeze(1)                                                      # :36
    for l in range(n_layers):
        p = f"{prefix}.layers.{l}"
        u = _ln(sd, p + ".self_attn.norm", x)                                   # :78
        q, k, v = [_lin(sd, p + f".self_attn.attn.linears.{j}", u).view(B, -1, n_head, dk).transpose(1, 2)
                   for j in range(3)]                                           # :39
        s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)                # :14
        s = s.masked_fill(m4 == 0, -1e9)                                        # :16
        a = F.softmax(s, dim=-1)                                                # :17
        if record is not None and l == 0:
            record["attn0"] = a
        a = _drop(a, dropout, f"attn{l}")                                       # :19
        o = torch.matmul(a, v).transpose(1, 2).contiguous().view(B, -1, d)      # :20, :41
        x = _drop(_lin(sd, p + ".self_attn.attn.linears.3", o), dropout, f"sa{l}") + x      # :42, :80
        u = _ln(sd, p + ".feed_forward.norm", x)                                # :106
        hid = _drop(F.relu(_lin(sd, p + ".feed_forward.feed_forward.w_1", u)), dropout, f"ffh{l}")   # :54
        x = _drop(_lin(sd, p + ".feed_forward.feed_forward.w_2", hid), dropout, f"ffo{l}") + x       # :54, :108
        outs.append(_ln(sd, f"{prefix}.norms.{l}", x))                          # :177
    return outs


def mlp_head(sd: SD, prefix: str, h: Tensor, n_lin: int) -> Tensor:
    """cate.py:55-73 with activate_func=F.elu."""
    for j in range(n_lin - 1):
        h = F.elu(_lin(sd, f"{prefix}.linears.{j}", h))
    return _lin(sd, f"{prefix}.linears.{n_lin - 1}", h)


def cate_forward(sd: SD, X: Tensor, time_cond: Tensor, maskX: Tensor, *, n_layers: int = 12,
                 n_head: int = 8, stagewise_pos: bool = False, record: Optional[dict] = None,
                 dropout: Optional[dict] = None) -> Tensor:
    """CATE.forward, cate.py:145-164.  ``dropout=None``: eval mode (dropouts are identity,
    models/utils.py:63-65); otherwise training mode with injected keep masks (``_drop``)."""
    d = sd["opEmb.fc.weight"].shape[0]
    emb_x = _drop(_lin(sd, "opEmb.fc", X) * math.sqrt(d), dropout, "emb")       # transformer.py:230, cate.py:147
    emb_x = emb_x + positional_table(X.shape[1], d, stagewise_pos)[: X.shape[1]]  # cate.py:149-151
    emb = emb_x + time_mlp(sd, time_cond, d).unsqueeze(1)                       # cate.py:155-159
    hs = encoder_stack(sd, "graph_encoder.encoder_f", emb, maskX, n_layers, n_head, record, dropout)
    h = torch.cat(hs, dim=-1)                                                   # cate.py:85
    if record is not None:
        record["concat"] = h
    return mlp_head(sd, "final", h, 3)                                          # cate.py:162


def score_fn(sd: SD, sde, x: Tensor, t: Tensor, maskX: Tensor, **kw) -> Tensor:
    """models/utils.py:88-104 (VPSDE, continuous=True): labels = 999 t, score = -model/std;
    models/utils.py:106-117 (VESDE): labels = sigma(t), the model output is the score."""
    if isinstance(sde, VESDE):
        return cate_forward(sd, x, sde.time_cond(t), maskX, **kw)
    out = cate_forward(sd, x, t * 999, maskX, **kw)
    std = sde.marginal_std(t)
    return -out / std[:, None, None]


# ----------------------------------------------------------------------------------------------
# Denoising score matching loss and its parameter gradients (losses.py:110-168, :204-235)
# ----------------------------------------------------------------------------------------------

def dsm_loss(sd: SD, sde, x: Tensor, t: Tensor, z: Tensor, maskX: Tensor, *, reduce_mean: bool = True,
             dropout: Optional[dict] = None, **kw) -> Tensor:
    """get_sde_loss_fn_nas.loss_fn (losses.py:127-166) with likelihood_weighting=False and the random
    draws ``t`` (:139) and ``z`` (:141) injected.  VPSDE: mean = exp(lmc) x, std (sde_lib.py:179-188);
    VESDE: mean = x, std = sigma(t) (:279-282)."""
    if isinstance(sde, VESDE):
        mean, std = x, sde.marginal_std(t)
        labels = std
    else:
        lmc = -0.25 * t ** 2 * (sde.beta_1 - sde.beta_0) - 0.5 * t * sde.beta_0
        mean, std = torch.exp(lmc[:, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))
        labels = t * 999
    perturbed = mean + std[:, None, None] * z                                   # :145
    out = cate_forward(sd, perturbed, labels, maskX, dropout=dropout, **kw)
    score = out if isinstance(sde, VESDE) else -out / std[:, None, None]        # models/utils.py:103 / :116
    losses = torch.square(score * std[:, None, None] + z)                       # :149
    losses = losses.reshape(losses.shape[0], -1)
    losses = torch.mean(losses, dim=-1) if reduce_mean else 0.5 * torch.sum(losses, dim=-1)   # :151-154
    return losses.mean()                                                        # :155


def dsm_loss_and_grads(sd: SD, sde, x: Tensor, t: Tensor, z: Tensor, maskX: Tensor, **kw):
    """step_fn's ``loss.backward()`` (losses.py:222-223): the loss and d loss / d parameter for every
    entry of the state_dict (insertion order)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        loss = dsm_loss(leaves, sde, x, t, z, maskX, **kw)
        grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), {k: g for k, g in zip(leaves.keys(), grads)}


# ----------------------------------------------------------------------------------------------
# MetaNeuralPredictor surrogate (models/digcn_meta.py) and its guidance gradient
# ----------------------------------------------------------------------------------------------

def normalize_adj(adj: Tensor) -> Tensor:
    """digcn_meta.py:12-16."""
    rowsum = adj.sum(2, keepdim=True).repeat(1, 1, adj.size(-1))
    return torch.div(adj, rowsum)


def _mab(sd: SD, p: str, Q: Tensor, K: Tensor, heads: int) -> Tensor:
    """set_encoder/setenc_modules.py:23-37 (ln=False)."""
    Q = _lin(sd, p + ".fc_q", Q)
    Kp, V = _lin(sd, p + ".fc_k", K), _lin(sd, p + ".fc_v", K)
    dim_v = Q.shape[-1]
    split = dim_v // heads
    Q_ = torch.cat(Q.split(split, 2), 0)
    K_ = torch.cat(Kp.split(split, 2), 0)
    V_ = torch.cat(V.split(split, 2), 0)
    A = torch.softmax(Q_.bmm(K_.transpose(1, 2)) / math.sqrt(dim_v), 2)
    O = torch.cat((Q_ + A.bmm(V_)).split(Q.size(0), 0), 2)
    return O + F.relu(_lin(sd, p + ".fc_o", O))


def _setpool(sd: SD, p: str, X: Tensor, heads: int) -> Tensor:
    """set_encoder/setenc_models.py:35-38 for mode='sabPF'."""
    h = _mab(sd, p + ".enc.0.mab", X, X, heads)
    h = _mab(sd, p + ".enc.1.mab", h, h, heads)
    S = sd[p + ".dec.0.S"].repeat(h.size(0), 1, 1)
    h = _mab(sd, p + ".dec.0.mab", S, h, heads)
    return _lin(sd, p + ".dec.1", h)


def set_encode(sd: SD, task: Tensor, num_sample: int = 20, heads: int = 4) -> Tensor:
    """digcn_meta.py:131-139 for ONE task ``[num_class*num_sample, 512]`` -> D_mu ``[nz]``.
    The reference repeats the task B times and recomputes identical rows."""
    cls_protos = _setpool(sd, "intra_setpool", task.view(-1, num_sample, 512), heads).squeeze(1)
    return _setpool(sd, "inter_setpool", cls_protos.unsqueeze(0), heads).reshape(-1)


def predictor_forward(sd: SD, X: Tensor, time_cond: Tensor, maskX: Tensor, D_mu: Tensor) -> Tensor:
    """MetaNeuralPredictor.forward with cached D_mu, digcn_meta.py:109-129, :141-159, :39-48."""
    B, N, _ = X.shape
    g = sd["fc1.weight"].shape[1]
    te = timestep_embedding(time_cond, g).to(sd["timeEmb1.weight"].dtype)     # fp32 in the reference; a no-op unless run in fp64
    emb_t = _lin(sd, "timeEmb2", F.silu(_lin(sd, "timeEmb1", te)))              # :116-118
    adj = normalize_adj(maskX + torch.eye(N))                                   # :120
    out = X
    n_gcn = len([k for k in sd if k.startswith("gcn.") and k.endswith("weight1")])
    for l in range(n_gcn):
        na = normalize_adj(adj)                                                 # :42
        o1 = F.relu(torch.matmul(na, torch.matmul(out, sd[f"gcn.{l}.weight1"])))  # :43
        ina = normalize_adj(adj.transpose(1, 2))                                # :44
        o2 = F.relu(torch.matmul(ina, torch.matmul(out, sd[f"gcn.{l}.weight2"])))  # :45
        out = (o1 + o2) / 2                                                     # :46
    numv = torch.tensor([N] * B)
    out = torch.div(out.sum(1), numv.unsqueeze(-1).expand(B, g))                # :19-22
    out = out + emb_t                                                           # :125
    a_mu = F.linear(out, sd["fc1.weight"])                                      # :126
    vec = torch.cat([D_mu.reshape(1, -1).expand(B, -1), a_mu], dim=1)           # :142-147
    return _lin(sd, "pred_fc.2", torch.tanh(_lin(sd, "pred_fc.0", vec)))        # :148


def classifier_grad(sd: SD, x: Tensor, t: Tensor, maskX: Tensor, D_mu: Tensor, labels: str = "max", sde=None) -> Tensor:
    """models/utils.py:130-145 (regress=True): d/dx of +-sum_b f(x_b); logit_fn labels = 999 t (:159),
    or sigma(t) for a VESDE (:170)."""
    with torch.enable_grad():
        x_in = x.detach().clone().requires_grad_(True)
        logit = predictor_forward(sd, x_in, t * 999 if sde is None else sde.time_cond(t), maskX, D_mu)
        prob = logit.sum() if labels == "max" else -logit.sum()
        return torch.autograd.grad(prob, x_in)[0]


# ----------------------------------------------------------------------------------------------
# MetaPredictorCATE, the OFA surrogate (models/cate.py:251-360) and its guidance gradient
# ----------------------------------------------------------------------------------------------

def predictor_cate_forward(sd: SD, X: Tensor, time_cond: Tensor, maskX: Tensor, D_mu: Tensor, *,
                           n_layers: int = 2, n_head: int = 2) -> Tensor:
    """MetaPredictorCATE.forward with cached D_mu: arch_encode (cate.py:311-327: embedding WITHOUT
    positional encoding, encoder, `final` MLP, flatten, `regeress` MLP) then predict (:341-348)."""
    d = sd["opEmb.fc.weight"].shape[0]
    B = X.shape[0]
    emb = _lin(sd, "opEmb.fc", X) * math.sqrt(d) + time_mlp(sd, time_cond, d).unsqueeze(1)   # :312-320
    hs = encoder_stack(sd, "graph_encoder.encoder_f", emb, maskX, n_layers, n_head)
    h = mlp_head(sd, "final", torch.cat(hs, dim=-1), 3)                          # :323
    h = mlp_head(sd, "regeress", h.reshape(B, -1), 2)                            # :325-326
    vec = torch.cat([D_mu.reshape(1, -1).expand(B, -1), h], dim=1)               # :342-347
    return _lin(sd, "pred_fc.2", torch.tanh(_lin(sd, "pred_fc.0", vec)))         # :348


def classifier_grad_cate(sd: SD, x: Tensor, t: Tensor, maskX: Tensor, D_mu: Tensor, labels: str = "max", sde=None) -> Tensor:
    """models/utils.py:130-145 (regress=True) for the OFA surrogate."""
    with torch.enable_grad():
        x_in = x.detach().clone().requires_grad_(True)
        logit = predictor_cate_forward(sd, x_in, t * 999 if sde is None else sde.time_cond(t), maskX, D_mu)
        prob = logit.sum() if labels == "max" else -logit.sum()
        return torch.autograd.grad(prob, x_in)[0]


# ----------------------------------------------------------------------------------------------
# predictor-corrector loop (sampling.py:203-256, :424-445, :496-577)
# ----------------------------------------------------------------------------------------------

def pc_sample(score_sd: SD, sde, mask: Tensor, tape: dict, *, eps: float = 1e-3, snr: float = 0.16,
              n_steps: Optional[int] = None, pred_sd: Optional[SD] = None, D_mu: Optional[Tensor] = None,
              classifier_scale: float = 0.0, labels: str = "max", record: Optional[list] = None,
              model_kw: Optional[dict] = None, predictor: str = "reverse_diffusion", corrector: str = "langevin",
              weighting: Optional[dict] = None, chain: Optional[dict] = None, n_steps_each: int = 1) -> Tensor:
    """Predictor-corrector loop with injected noise (sampling.py:424-445 / :534-577).

    ``n_steps_each`` = c: corrector updates per PC step (sampling.py:244); the corrector tape then holds ``S * c``
    entries (entry ``i * c + k``) and a record gains ``corr`` = the list of the c updates' dicts (the flat keys
    ``score_c`` ... ``x_corr`` describe the last one).

    ``chain`` = {'number_chain_steps': n, 'keep_chain': k, 'denoise': bool} restates the MobileNetV3 sampler's chain
    recording (MobileNetV3/sampling.py:980-988, :1030-1034): after PC step i the first k graphs of
    ``inverse_scaler(x_mean if denoise else x)`` go to slot ``n - 1 - (i * n) // sde.N`` of a zero tensor [n, k, N, V],
    which is stored in ``chain['chain_x']`` (PARITY UNPINNED like the weighting options: same file).

    ``weighting`` (None: the NAS-Bench-201 tree's ``score + classifier_scale * grad``) restates the guidance weighting
    options that only the MobileNetV3 tree's sampler has (MobileNetV3/sampling.py:858-910, :924-957): keys
    ``weight_scheduling`` (w = scale * 0.1 ** t), ``weight_ratio`` (w *= ||score_b||_p / ||grad_b||_p per sample;
    ``prod_w``: / ||w grad_b||_p), ``weight_ratio_abs`` (w *= |score| / |grad| per element), ``p``, ``t_spot`` /
    ``t_spot_end`` (guidance only while t_spot_end <= t <= t_spot).  PARITY UNPINNED for these options: that file
    imports torchdiffeq, which is not installed here, so the reference cannot be run for them; the restatement
    follows its text and is checked against the kernels only.

    ``predictor``: 'reverse_diffusion' (sampling.py:203-213), 'euler_maruyama' (:189-200) or 'none'
    (:216-224); ``corrector``: 'langevin' (:227-256) or 'none' (:259-267); ``sde``: VPSDE or VESDE.
    ``tape`` = {'x0': [B,N,V], 'corrector': [S,B,N,V], 'predictor': [S,B,N,V]} replaces
    ``prior_sampling`` (sampling.py:436), ``randn_like`` at :247 and at :210 / :196, in that order.
    ``n_steps`` truncates the loop (the time grid is always the full ``linspace(T, eps, sde.N)``).
    Returns ``inverse_scaler(x_mean_last)`` (sampling.py:445 / :577, noise_removal=True).
    """
    kw = model_kw or {}
    x = tape["x0"].clone()
    B = x.shape[0]
    ts = timesteps(sde, eps)
    maskX = mask.unsqueeze(0).repeat(B, 1, 1)                                   # :438
    S = sde.N if n_steps is None else n_steps

    def total_grad(x, vec_t):                                                   # :502-505
        s = score_fn(score_sd, sde, x, vec_t, maskX, **kw)
        if pred_sd is None:
            return s, s, None
        grad_fn = classifier_grad_cate if "regeress.linears.0.weight" in pred_sd else classifier_grad
        cg = grad_fn(pred_sd, x, vec_t, maskX, D_mu, labels, sde)
        if weighting is None:
            return s + classifier_scale * cg, s, cg
        wo = weighting
        p = wo.get("p", 1)
        t0 = vec_t[0].item()
        w = classifier_scale * 0.1 ** t0 if wo.get("weight_scheduling") else classifier_scale      # MBV3 :858-859, :875-878
        if wo.get("weight_ratio"):                                                                   # :880-886
            den = (w * cg) if wo.get("prod_w") else cg
            ratio = s.view(B, -1).norm(p=p, dim=-1) / den.view(B, -1).norm(p=p, dim=-1)
            w = w * ratio[:, None, None]
        if wo.get("weight_ratio_abs"):                                                               # :888-891
            assert not wo.get("weight_ratio")
            w = w * torch.div(torch.abs(s), torch.abs(cg))
        t_spot = wo.get("t_spot", 1.)
        t_end = wo.get("t_spot_end") or eps                                                          # :855-856
        if t_spot < 1. and not (t0 <= t_spot and vec_t[0] >= t_end):                                 # :902-907
            return s, s, cg
        return s + w * cg, s, cg

    x_mean = x
    if chain is not None:
        ncs = chain.get('number_chain_steps') or sde.N                          # MBV3 :982-983
        keep = chain.get('keep_chain') or B                                     # :984-985
        assert ncs <= sde.N                                                     # :986
        chain['chain_x'] = torch.zeros(ncs, keep, *x.shape[1:])                 # :987-988
    with torch.no_grad():
        for i in range(S):
            vec_t = torch.ones(B) * ts[i]                                       # :442
            rec = {"x_in": x.clone()} if record is not None else None
            if corrector == "langevin":
                # -- LangevinCorrector.update_fn, sampling.py:232-256
                alpha = sde.langevin_alpha(vec_t)
                for k in range(n_steps_each):                                   # :244
                    grad, s1, cg1 = total_grad(x, vec_t)
                    noise = tape["corrector"][i * n_steps_each + k]
                    grad_norm = torch.norm(grad.reshape(B, -1), dim=-1).mean()
                    noise_norm = torch.norm(noise.reshape(B, -1), dim=-1).mean()
                    step_size = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
                    x_mean = x + step_size[:, None, None] * grad
                    x = x_mean + torch.sqrt(step_size * 2)[:, None, None] * noise
                    if rec is not None:
                        one = dict(score_c=s1, cgrad_c=cg1, total_c=grad, grad_norm=grad_norm, noise_norm=noise_norm,
                                   step_size=step_size[0].clone(), x_corr=x.clone())
                        rec.update(one)
                        rec.setdefault("corr", []).append(one)
            else:
                assert corrector == "none"
                x_mean = x
            if predictor == "reverse_diffusion":
                # -- ReverseDiffusionPredictor.update_fn, sampling.py:208-213 + sde_lib.py:119-125
                f, G = sde.discretize(x, vec_t)
                grad2, s2, cg2 = total_grad(x, vec_t)
                rev_f = f - G[:, None, None] ** 2 * grad2
                z = tape["predictor"][i]
                x_mean = x - rev_f
                x = x_mean + G[:, None, None] * z
            elif predictor == "euler_maruyama":
                # -- EulerMaruyamaPredictor.update_fn, sampling.py:194-200 + sde_lib.py:95-103
                dt = -1. / sde.N
                z = tape["predictor"][i]
                drift, diffusion = sde.sde(x, vec_t)
                grad2, s2, cg2 = total_grad(x, vec_t)
                drift = drift - diffusion[:, None, None] ** 2 * grad2 * 1.
                x_mean = x + drift * dt
                x = x_mean + diffusion[:, None, None] * math.sqrt(-dt) * z
            else:
                assert predictor == "none"
                grad2 = s2 = cg2 = None
                x_mean = x
            if rec is not None:
                rec.update(score_p=s2, cgrad_p=cg2, total_p=grad2, x_mean=x_mean.clone(), x_pred=x.clone())
                record.append(rec)
            if chain is not None:                                               # MBV3 :1030-1034
                write_index = ncs - 1 - int((i * ncs) // sde.N)
                chain['chain_x'][write_index] = inverse_scaler(x_mean if chain.get('denoise', True) else x)[:keep]
    return inverse_scaler(x_mean)


# ----------------------------------------------------------------------------------------------
# Exit of a sampling round: validity rules and architecture strings of NAS-Bench-201
# (analysis/arch_functions.py:209-256, :284-329; BasicArchMetricsMeta.compute_validity :95-118)
# ----------------------------------------------------------------------------------------------
NB201_OPS = ['input', 'output', 'none', 'skip_connect', 'nor_conv_1x1', 'nor_conv_3x3', 'avg_pool_3x3']   # :88


def is_valid_nas201_x(x, START_TYPE: int = 0, END_TYPE: int = 1):
    """arch_functions.py:316-329 on a quantised [8, 7] array: (is_valid, error code) with the codes of
    :308-313 (1 MULTIPLE_NODE_TYPES, 2 No_START_END, 3 INTERM_START_END, -1 NO_ERROR)."""
    x = np.asarray(x)
    for row in x:                                                               # :284-288
        if int(np.sum(row)) != 1:
            return False, 1
    if x[0][START_TYPE] != 1 or x[-1][END_TYPE] != 1:                           # :291-296
        return False, 2
    for row in x[1:-1]:                                                         # :299-305
        if row[START_TYPE] == 1 or row[END_TYPE] == 1:
            return False, 3
    return True, -1


def nb201_arch_code(x) -> int:
    """The six edge operations of a VALID cell as one integer: sum_i (argmax(x[i+1]) - 2) 5^i; the argmax
    decode is arch_functions.py:209-216 (edge order (1,0), (2,0), (2,1), (3,0), (3,1), (3,2))."""
    x = np.asarray(x)
    return int(sum((int(np.argmax(x[i + 1])) - 2) * 5 ** i for i in range(6)))


def nb201_code_to_string(code: int, ops=NB201_OPS) -> str:
    """arch_functions.py:252-255 from the integer code."""
    t = [ops[(code // 5 ** i) % 5 + 2] for i in range(6)]
    return '|{}~0|+|{}~0|{}~1|+|{}~0|{}~1|{}~2|'.format(*t)


def compute_validity(generated, ops=NB201_OPS):
    """BasicArchMetricsMeta.compute_validity (:95-118) on a list of quantised [8, 7] arrays:
    (error codes, strings or None per sample)."""
    errs, strs = [], []
    for x in generated:
        ok, err = is_valid_nas201_x(x, ops.index('input'), ops.index('output'))
        errs.append(err)
        strs.append(nb201_code_to_string(nb201_arch_code(x), ops) if ok else None)
    return errs, strs


# ----------------------------------------------------------------------------------------------
# Exit of a MobileNetV3 / OFA sampling round (MobileNetV3/analysis/arch_functions.py:8-24, :135-163, :374-429).
# PARITY UNPINNED: that module imports wandb and igraph, which are not installed here, so it cannot be imported to
# capture goldens; this restatement follows its text and is checked on hand-derived known answers
# (tests/test_oracle_golden.py) and against the kernel.
# ----------------------------------------------------------------------------------------------
OFA_NUM_STAGE, OFA_MAX_LAYER_PER_STAGE = 5, 4                                      # :11-12
OFA_OPS2STR = {0: '3-3', 1: '3-4', 2: '3-6', 3: '5-3', 4: '5-4', 5: '5-6', 6: '7-3', 7: '7-4', 8: '7-6'}   # :20-24


def is_valid_ofa_x(x):
    """is_valid_OFA_x (:413-429): (ok, error code) for a quantised [20, 9] array."""
    x = np.asarray(x)
    assert x.ndim == 2
    for row in x:                                                                  # check_multiple_nodes :374-380
        if int(np.sum(row)) > 1:
            return False, 1
    for layer in [0, 1, 4, 5, 8, 9, 12, 13, 16, 17]:                               # check_none_in_1_and_2_layers :386-392
        if int(np.sum(x[layer])) == 0:
            return False, 2
    for layer in [2, 6, 10, 14, 18]:                                               # check_none_in_3_and_4_layers :394-402
        if int(np.sum(x[layer])) == 0 and int(np.sum(x[layer + 1])) != 0:
            return False, 3
    return True, -1


def ofa_string_from_onehot_x(x):
    """get_string_from_onehot_x (:135-144): '<stage depth>-<kernel>-<expand>' per layer slot, '0-0-0' for a skipped one."""
    x = np.asarray(x)
    ds = x.reshape(OFA_NUM_STAGE, -1).sum(axis=1)
    parts = []
    for i, row in enumerate(x):
        if row.sum() == 0:
            parts.append('0-0-0')
        else:
            parts.append(f'{int(ds[i // OFA_MAX_LAYER_PER_STAGE])}-' + OFA_OPS2STR[int(np.nonzero(row)[0][0])])
    return '_'.join(parts)


def compute_validity_ofa(generated):
    """BasicArchMetricsMetaOFA.compute_validity (:146-163): (error code per sample, string or None per sample)."""
    errs, strs = [], []
    for x in generated:
        ok, err = is_valid_ofa_x(x)
        errs.append(err)
        strs.append(ofa_string_from_onehot_x(x) if ok else None)
    return errs, strs
