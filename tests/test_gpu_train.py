"""Parity of the HIP DSM training step (dnag_cate_dsm_step through the C ABI, driven by the host mirror of
the reference's losses.py) with the golden vectors captured from the reference's own loss_fn + backward
(tests/golden/train_*.npz) and with the CPU oracle on seeded inputs.  Needs an MI355X.

Tolerances: fp32 with a different summation order than autograd on the CPU.  Loss 1e-5 relative; a gradient
tensor to 5e-5 of its own max (measured ~1e-6 .. 1e-5; the encoder weight gradients sum ~10^2..10^4 rows)."""

from unittest import mock

import numpy as np
import pytest
import torch

from diffusionnag_amd import cate as _cate  # noqa: F401  (registers the model)
from diffusionnag_amd import losses, mutils, sde_lib, synthetic
from diffusionnag_amd.specs import CateDims
from oracle import dnag_oracle as O
from tests.helpers import golden, score_config
from tests.test_oracle_golden import train_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_model(seed=0):
    m = mutils.create_model(score_config(CateDims(), DEV))
    m.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=seed, perturb_norms=True))
    return m


def rel_err(a, b, floor=0.0):
    """max|a - b| relative to max|b|.  `floor` = 2e-6 of the largest gradient entry of the whole network is
    forgiven first: a gradient that is zero analytically (the key bias: softmax is shift invariant) is fp32
    rounding noise of that size in both implementations."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert np.isfinite(a).all()
    return max(np.abs(a - b).max() - floor, 0.0) / (np.abs(b).max() + 1e-30)


def run_step(model, sde, x, t, z, mask, dropout, train=True, reduce_mean=True):
    """loss_fn + loss.backward() exactly as the reference's step_fn does (losses.py:221-223)."""
    fn = losses.get_sde_loss_fn_nas(sde, train=train, reduce_mean=reduce_mean, continuous=True, likelihood_weighting=False)
    fn.draws = (t.to(DEV), z.to(DEV))
    if dropout is not None:
        fn.dropout_keep = {k: v.to(DEV) for k, v in dropout["masks"].items()}
        model.dropout_p = dropout["p"]
    else:
        model.dropout_p = 0.0
    model.train(train)
    model.zero_grad()
    loss = fn(model, (x.to(DEV), mask.to(DEV), mask.to(DEV)))
    if train:
        loss.backward()
    torch.cuda.synchronize()
    return loss


@pytest.mark.parametrize("name", ["train_nb201_vp_nodrop", "train_nb201_vp_drop", "train_nb201_ve_drop"])
def test_dsm_step_against_reference_golden(name):
    g = golden(name)
    x, t, z, dropout = train_case(g)
    model = make_model(0)
    sde = sde_lib.VPSDE(0.1, 20.0, 1000) if str(g["sde_kind"]) == "vp" else sde_lib.VESDE(0.1, 5.0, 1000)
    maskB = torch.from_numpy(g["mask"]).unsqueeze(0).repeat(x.shape[0], 1, 1)
    loss = run_step(model, sde, x, t, z, maskB, dropout)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    params = dict(model.named_parameters())
    names = list(model.state_dict().keys())
    norms = np.array([float(params[k].grad.double().norm()) for k in names])
    assert np.abs(norms - g["grad_norms"]).max() <= 1e-4 * np.abs(g["grad_norms"]).max()
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-7)
    floor = 2e-6 * max(np.abs(g[key]).max() for key in g if key.startswith("grad:"))
    for key in g:
        if key.startswith("grad:"):
            k = key[5:]
            got = params[k[:-4]].grad[:8] if k.endswith("[:8]") else params[k].grad
            # only the key bias is zero analytically (softmax is shift invariant): its gradient is rounding noise
            e = rel_err(got.cpu().numpy(), g[key], floor if ".attn.linears.1.bias" in k else 0.0)
            assert e <= 5e-5, f"{key}: {e:.3e}"


@pytest.mark.parametrize("sde_kind,B,reduce_mean", [("vp", 19, True), ("ve", 9, False)])
def test_dsm_step_against_oracle_ragged_batch(sde_kind, B, reduce_mean):
    """A batch that does not fill the last workgroup, per-graph masks (raw adjacency and its closure mixed),
    dropout active: every gradient tensor against the oracle's autograd."""
    rs = np.random.RandomState(7)
    ops = rs.randint(0, 7, size=(B, 8))
    x = np.full((B, 8, 7), -1.0, np.float32)
    x[np.arange(B)[:, None], np.arange(8)[None, :], ops] = 1.0
    x = torch.from_numpy(x)
    t = torch.from_numpy(rs.uniform(1e-5, 1, size=B).astype(np.float32))
    z = torch.from_numpy(rs.standard_normal((B, 8, 7)).astype(np.float32))
    adj = O.nb201_adj()
    clo = torch.as_tensor(O.floyed(adj.clone()), dtype=torch.float32)
    mask = torch.stack([adj if i % 2 == 0 else clo for i in range(B)])
    p = 0.1
    dropout = {"p": p, "masks": {k: torch.from_numpy((rs.uniform(0, 1, size=shape) >= p).astype(np.float32))
                                 for k, shape in O.dropout_sites(B, 8).items()}}
    sd = synthetic.cate_state_dict(CateDims(), seed=3, perturb_norms=True)
    osde = O.VPSDE(0.1, 20.0, 1000) if sde_kind == "vp" else O.VESDE(0.1, 5.0, 1000)
    want_loss, want = O.dsm_loss_and_grads(sd, osde, x, t, z, mask, dropout=dropout, reduce_mean=reduce_mean)
    model = make_model(3)
    sde = sde_lib.VPSDE(0.1, 20.0, 1000) if sde_kind == "vp" else sde_lib.VESDE(0.1, 5.0, 1000)
    loss = run_step(model, sde, x, t, z, mask, dropout, reduce_mean=reduce_mean)
    assert abs(float(loss.detach()) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    worst = ("", 0.0)
    floor = 2e-6 * max(float(v.abs().max()) for v in want.values())
    for k, prm in model.named_parameters():
        e = rel_err(prm.grad.cpu().numpy(), want[k].numpy(), floor if ".attn.linears.1.bias" in k else 0.0)
        if e > worst[1]:
            worst = (k, e)
        assert e <= 5e-5, f"{k}: {e:.3e}"
    print("worst gradient tensor:", worst)


def test_full_size_batch_against_oracle():
    """BASELINE config 3 size (B = 1024, dropout active, masks injected): the loss and EVERY gradient tensor against
    the oracle's autograd (the reference's loss_fn + backward restated, losses.py:127-166).  At this size the encoder
    backward runs 4 graphs per workgroup (256 workgroups) and the weight gradients reduce 8192 rows."""
    B = 1024
    rs = np.random.RandomState(23)
    ops = rs.randint(0, 7, size=(B, 8))
    x = np.full((B, 8, 7), -1.0, np.float32)
    x[np.arange(B)[:, None], np.arange(8)[None, :], ops] = 1.0
    x = torch.from_numpy(x)
    t = torch.from_numpy(rs.uniform(1e-5, 1, size=B).astype(np.float32))
    z = torch.from_numpy(rs.standard_normal((B, 8, 7)).astype(np.float32))
    mask = O.nb201_adj().unsqueeze(0).repeat(B, 1, 1)
    p = 0.1
    dropout = {"p": p, "masks": {k: torch.from_numpy((rs.uniform(0, 1, size=shape) >= p).astype(np.float32))
                                 for k, shape in O.dropout_sites(B, 8).items()}}
    sd = synthetic.cate_state_dict(CateDims(), seed=3, perturb_norms=True)
    osde = O.VPSDE(0.1, 20.0, 1000)
    model = make_model(3)
    loss = run_step(model, sde_lib.VPSDE(0.1, 20.0, 1000), x, t, z, mask, dropout, reduce_mean=True)
    got = {k: prm.grad.cpu().numpy() for k, prm in model.named_parameters()}

    # The FFN's ReLU (transformer.py:106) makes the loss piecewise smooth in the weights: among the 12.6 M hidden
    # pre-activations of this batch about twenty are zero to fp32 rounding (|pre| < 5e-7 of the layer's largest; e.g.
    # layer 6, unit 35, row 7883: 1.7e-8), and on which side such a unit falls is not determined in fp32 -- its whole
    # contribution to the gradient (1e-4 of the max of that layer's w_1 gradient, tests/micro/train_err_b1024.py) comes
    # or goes with it.  Graphs are independent terms of the loss, so switching one such unit changes the gradient by a
    # fixed vector delta_c (computed by the oracle on that unit's graph alone); the HIP gradients must equal the
    # oracle's plus the deltas of SOME subset of the undecidable units, chosen by projection.
    pre = []
    relu0 = O.F.relu

    def rec_relu(v):
        pre.append(v.detach())
        return relu0(v)
    with mock.patch.object(O.F, "relu", rec_relu):
        want_loss, want = O.dsm_loss_and_grads(sd, osde, x, t, z, mask, dropout=dropout, reduce_mean=True)
    assert len(pre) == 12
    assert abs(float(loss.detach()) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    cands = []
    for l, v in enumerate(pre):
        near = (v.abs() < 5e-7 * v.abs().max()) & (dropout["masks"][f"ffh{l}"].reshape(v.shape) > 0)
        cands += [(l, tuple(int(i) for i in idx)) for idx in torch.nonzero(near)]
    assert len(cands) < 100 and len({c[1][0] for c in cands}) == len(cands), "undecidable units must sit in different graphs"
    names = list(got.keys())
    want = {k: want[k].numpy().astype(np.float64) for k in names}
    switched = []
    for l, (g_, n_, j_) in cands:
        sub = {"p": p, "masks": {k: v[g_:g_ + 1] for k, v in dropout["masks"].items()}}
        one = (x[g_:g_ + 1], t[g_:g_ + 1], z[g_:g_ + 1], mask[g_:g_ + 1])
        # the unit is forced to either side explicitly: on its own the graph runs through other BLAS kernels than inside
        # the batch and may round this pre-activation to the other sign
        side = {}
        for state in (False, True):
            calls = [0]

            def relu(v, calls=calls, l=l, n_=n_, j_=j_, state=state):
                on = v > 0
                if calls[0] == l:
                    on[0, n_, j_] = state
                calls[0] += 1
                return v * on.to(v.dtype)
            with mock.patch.object(O.F, "relu", relu):
                side[state] = O.dsm_loss_and_grads(sd, osde, *one, dropout=sub, reduce_mean=True)[1]
        was_on = bool(pre[l][g_, n_, j_] > 0)
        g0, g1 = side[was_on], side[not was_on]
        delta = {k: (g1[k].numpy().astype(np.float64) - g0[k].numpy()) / B for k in names}
        num = sum(float(((got[k] - want[k]) * delta[k]).sum()) for k in names)
        den = sum(float((delta[k] ** 2).sum()) for k in names)
        if den > 0 and num / den > 0.5:
            switched.append((l, g_, n_, j_))
            for k in names:
                want[k] = want[k] + delta[k]
    floor = 2e-6 * max(float(np.abs(v).max()) for v in want.values())
    errs = {k: rel_err(got[k], want[k], floor if ".attn.linears.1.bias" in k else 0.0) for k in names}
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(f"B=1024: {len(cands)} undecidable relu units, {len(switched)} on the other side here {switched}; worst gradient tensor {worst}")
    assert len(switched) <= len(cands) // 2 + 2
    assert worst[1] <= 5e-5, f"{worst[0]}: {worst[1]:.3e}"


def test_eval_loss_and_no_grad():
    """train=False (losses.py:228-233): dropout off, no gradients; equals the oracle's eval-mode loss."""
    g = golden("train_nb201_vp_nodrop")
    x, t, z, _ = train_case(g)
    model = make_model(0)
    maskB = torch.from_numpy(g["mask"]).unsqueeze(0).repeat(x.shape[0], 1, 1)
    with torch.no_grad():
        loss = run_step(model, sde_lib.VPSDE(0.1, 20.0, 1000), x, t, z, maskB, None, train=False)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert all(p.grad is None for p in model.parameters())


def test_full_size_batch_properties():
    """BASELINE config 3 size (B = 1024).  Size-independent properties: (i) a batch made of two copies of a
    half batch has the half batch's loss and gradients (both are batch means); (ii) the on-device Philox dropout
    masks keep ~ 1 - p of the sites and give finite gradients."""
    B = 1024
    rs = np.random.RandomState(11)
    ops = rs.randint(0, 7, size=(B // 2, 8))
    x = np.full((B // 2, 8, 7), -1.0, np.float32)
    x[np.arange(B // 2)[:, None], np.arange(8)[None, :], ops] = 1.0
    x = torch.from_numpy(x)
    t = torch.from_numpy(rs.uniform(1e-5, 1, size=B // 2).astype(np.float32))
    z = torch.from_numpy(rs.standard_normal((B // 2, 8, 7)).astype(np.float32))
    mask = O.nb201_adj().unsqueeze(0).repeat(B // 2, 1, 1)
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    model = make_model(0)
    l1 = run_step(model, sde, x, t, z, mask, None)
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    gmax = max(float(v.abs().max()) for v in g1.values())
    l2 = run_step(model, sde, torch.cat([x, x]), torch.cat([t, t]), torch.cat([z, z]), torch.cat([mask, mask]), None)
    assert abs(float(l1.detach()) - float(l2.detach())) <= 2e-6 * abs(float(l1.detach()))
    for k, p in model.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), g1[k].cpu().numpy(), 2e-6 * gmax)
        assert e <= 2e-5, f"{k}: {e:.3e}"
    # Philox dropout
    import ctypes as C
    from diffusionnag_amd import _lib
    lib = _lib.load()
    dc = model.dims_c()
    n = lib.dnag_dropout_mask_bytes(C.byref(dc), B)
    keep = torch.empty(n, dtype=torch.uint8, device=DEV)
    _lib.check(lib.dnag_dropout_fill(C.byref(dc), B, 0.1, 5, keep.data_ptr(), _lib.current_stream(torch.device(DEV))), "fill")
    rate = float(keep.float().mean())
    assert abs(rate - 0.9) < 2e-3, rate
    fn = losses.get_sde_loss_fn_nas(sde, train=True, reduce_mean=True, continuous=True, likelihood_weighting=False)
    model.dropout_p = 0.1
    model.train()
    model.zero_grad()
    xx, mm = torch.cat([x, x]).to(DEV), torch.cat([mask, mask]).to(DEV)
    loss = fn(model, (xx, mm, mm))
    loss.backward()
    assert np.isfinite(float(loss.detach()))
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_second_stream_changes_no_gradient_bit(monkeypatch):
    """The head's weight gradients run on the library's second stream (dnag_lib.hip SideStream) beside the chain that
    leads to the encoder backward; DNAG_OVERLAP=0 keeps the step on one stream.  Same kernels, same operands: every
    gradient and the loss equal, over three steps in a row on the same buffers (an ordering slip would show as a race)."""
    g = golden("train_nb201_vp_drop")
    x, t, z, dropout = train_case(g)
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    maskB = torch.from_numpy(g["mask"]).unsqueeze(0).repeat(x.shape[0], 1, 1)

    def grads():
        model = make_model(0)
        out = []
        for _ in range(3):
            loss = run_step(model, sde, x, t, z, maskB, dropout)
            out.append([float(loss.detach())] + [p.grad.detach().clone() for _, p in model.named_parameters()])
        return out

    two = grads()
    monkeypatch.setenv("DNAG_OVERLAP", "0")
    one = grads()
    for a, b in zip(one, two):
        assert a[0] == b[0]
        for ga, gb in zip(a[1:], b[1:]):
            assert torch.equal(ga, gb)
    for a in two[1:]:
        for ga, gb in zip(a[1:], two[0][1:]):
            assert torch.equal(ga, gb)


def test_sampler_and_training_step_share_the_second_stream():
    """dnag_pc_sample and dnag_cate_dsm_step use the same library-owned stream per device: a trajectory, a training
    step and the trajectory again, back to back on one caller stream -- the two trajectories are equal to the bit and
    the step's gradients equal those of a step taken alone."""
    from tests.test_gpu_parity import run_sampler
    g = golden("train_nb201_vp_drop")
    x, t, z, dropout = train_case(g)
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    maskB = torch.from_numpy(g["mask"]).unsqueeze(0).repeat(x.shape[0], 1, 1)
    gs = golden("pc_guided_nb201_n1000_first8")
    model = make_model(0)
    run_step(model, sde, x, t, z, maskB, dropout)
    alone = [p.grad.detach().clone() for p in model.parameters()]
    f1, _ = run_sampler(gs, True, record=False)
    model2 = make_model(0)
    run_step(model2, sde, x, t, z, maskB, dropout)
    f2, _ = run_sampler(gs, True, record=False)
    host = lambda v: v.detach().cpu() if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
    assert torch.equal(host(f1), host(f2))
    for a, p in zip(alone, model2.parameters()):
        assert torch.equal(a, p.grad)
