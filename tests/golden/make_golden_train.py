"""Capture golden vectors of the DSM TRAINING STEP from the REFERENCE implementation (build container
only):  loss and parameter gradients of ``losses.get_step_fn(...).step_fn`` up to ``loss.backward()``
(reference ``losses.py:110-168``, ``:204-223``) for a small NAS-Bench-201 batch.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py

The reference's own ``get_sde_loss_fn_nas`` closure is driven with ``model.train()`` semantics
(``train=True``); the random draws are injected by wrapping -- in this process only -- ``torch.rand``
(t), ``torch.randn_like`` (z) and ``torch.nn.functional.dropout`` (keep masks, consumed in the order
the reference's forward reaches its dropout sites).  Only inputs, seeds and outputs are stored.
"""

from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/NAS-Bench-201"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from diffusionnag_amd import synthetic  # noqa: E402
from diffusionnag_amd.specs import CateDims  # noqa: E402
from make_golden import score_config  # noqa: E402
from oracle import dnag_oracle as O  # noqa: E402  (only for the list of dropout sites and shapes)


def train_inputs(B, seed):
    """Seeded inputs of one training step: one-hot architectures scaled to +-1, raw-adjacency masks,
    t ~ U(eps, 1) (as a uniform draw u; t = u (T - eps) + eps), z ~ N(0, 1)."""
    rs = np.random.RandomState(seed)
    ops = rs.randint(0, 7, size=(B, 8))
    x = np.zeros((B, 8, 7), np.float32)
    x[np.arange(B)[:, None], np.arange(8)[None, :], ops] = 1.0
    x = 2.0 * x - 1.0                                          # data.centered (datasets_nas.py:11-18)
    u = rs.uniform(0, 1, size=B).astype(np.float32)
    z = rs.standard_normal((B, 8, 7)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(u), torch.from_numpy(z)


def keep_masks(B, p, seed):
    rs = np.random.RandomState(seed)
    return {k: torch.from_numpy((rs.uniform(0, 1, size=shape) >= p).astype(np.float32))
            for k, shape in O.dropout_sites(B, 8).items()}


def main():
    scratch = tempfile.mkdtemp(prefix="dnag_golden_")
    os.chdir(scratch)
    sys.path.insert(0, REF)
    import sde_lib                              # reference
    import losses                               # reference
    import models.utils as mutils               # reference
    from models import cate                     # noqa: F401

    nb = CateDims()
    adj = synthetic.nb201_adjacency()

    def run(tag, *, B, p, sde_kind, seed, mask_seed, eps=1e-5):
        cfg = score_config(nb)
        cfg.model["dropout"] = 0.1
        model = mutils.create_model(cfg)
        model.load_state_dict(synthetic.cate_state_dict(nb, seed=0, perturb_norms=True), strict=True)
        sde = sde_lib.VPSDE(0.1, 20.0, 1000) if sde_kind == "vp" else sde_lib.VESDE(0.1, 5.0, 1000)
        x, u, z = train_inputs(B, seed)
        masks = keep_masks(B, p, mask_seed) if p > 0 else {}
        queue = list(masks.values())
        orig = (torch.rand, torch.randn_like, torch.nn.functional.dropout)

        def fake_dropout(inp, p_=0.5, training=True, inplace=False):
            assert training and abs(p_ - 0.1) < 1e-12
            if p == 0:
                return inp
            keep = queue.pop(0)
            assert keep.shape == inp.shape, (keep.shape, inp.shape)
            return inp * keep / (1.0 - p)

        torch.rand = lambda *a, **k: u.clone()
        torch.randn_like = lambda t_: z.clone()
        torch.nn.functional.dropout = fake_dropout
        try:
            loss_fn = losses.get_sde_loss_fn_nas(sde, train=True, reduce_mean=True, continuous=True,
                                                 likelihood_weighting=False, eps=eps)
            model.zero_grad()
            maskB = adj.unsqueeze(0).repeat(B, 1, 1)
            loss = loss_fn(model, (x, maskB, maskB))            # batch = (x, adj, mask), losses.py:137
            loss.backward()
        finally:
            torch.rand, torch.randn_like, torch.nn.functional.dropout = orig
        assert not queue, f"{len(queue)} dropout masks were not consumed"
        arrs = dict(seed=0, input_seed=seed, mask_seed=mask_seed, B=B, p=p, sde_kind=sde_kind, eps=eps, mask=adj,
                    loss=loss.detach())
        names = [n for n, _ in model.named_parameters()]
        assert names == list(model.state_dict().keys())
        # gradients: a few whole tensors and the norm of every tensor (keeps the fixture small)
        keep_full = ["opEmb.fc.weight", "opEmb.fc.bias", "timeEmb1.bias", "timeEmb2.weight",
                     "graph_encoder.encoder_f.layers.0.self_attn.norm.weight", "graph_encoder.encoder_f.layers.0.self_attn.norm.bias",
                     "graph_encoder.encoder_f.layers.0.self_attn.attn.linears.0.weight",
                     "graph_encoder.encoder_f.layers.5.self_attn.attn.linears.1.bias",
                     "graph_encoder.encoder_f.layers.5.self_attn.attn.linears.2.weight",
                     "graph_encoder.encoder_f.layers.11.self_attn.attn.linears.3.weight",
                     "graph_encoder.encoder_f.layers.3.feed_forward.norm.weight", "graph_encoder.encoder_f.layers.3.feed_forward.norm.bias",
                     "graph_encoder.encoder_f.layers.3.feed_forward.feed_forward.w_1.weight",
                     "graph_encoder.encoder_f.layers.11.feed_forward.feed_forward.w_2.weight",
                     "graph_encoder.encoder_f.layers.7.feed_forward.feed_forward.w_2.bias",
                     "graph_encoder.encoder_f.norms.0.weight", "graph_encoder.encoder_f.norms.11.bias",
                     "final.linears.0.bias", "final.linears.1.bias", "final.linears.2.weight", "final.linears.2.bias"]
        params = dict(model.named_parameters())
        for k in keep_full:
            arrs["grad:" + k] = params[k].grad.detach()
        arrs["grad_norms"] = torch.stack([params[k].grad.detach().double().norm() for k in names]).float()
        arrs["grad_sums"] = torch.stack([params[k].grad.detach().double().sum() for k in names]).float()
        # slices of the two big head matrices
        arrs["grad:final.linears.0.weight[:8]"] = params["final.linears.0.weight"].grad[:8].detach()
        arrs["grad:final.linears.1.weight[:8]"] = params["final.linears.1.weight"].grad[:8].detach()
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
        print(f"{tag}.npz  {os.path.getsize(path)/1024:.1f} KiB  loss={float(loss):.6f}")

    run("train_nb201_vp_nodrop", B=6, p=0.0, sde_kind="vp", seed=41, mask_seed=0)
    run("train_nb201_vp_drop", B=6, p=0.1, sde_kind="vp", seed=42, mask_seed=43)
    run("train_nb201_ve_drop", B=5, p=0.1, sde_kind="ve", seed=44, mask_seed=45)


if __name__ == "__main__":
    main()
