"""Pin the CPU oracle (oracle/dnag_oracle.py) against vectors captured from the reference
(tests/golden/*.npz, produced by tests/golden/make_golden.py).  Same aten CPU kernels on both
sides, so the tolerances are tight; they are not zero because the oracle batches/fuses a few
ops differently (e.g. D_mu computed once instead of B times)."""

import numpy as np
import pytest
import torch

from diffusionnag_amd import synthetic
from diffusionnag_amd.specs import CateDims, PredictorDims
from oracle import dnag_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    ok = err <= atol + rtol * np.abs(b)
    assert ok.all(), f"max abs err {err.max():.3e} (ref scale {np.abs(b).max():.3e}), {(~ok).sum()} bad"


@pytest.mark.parametrize("mname", ["raw", "closure"])
def test_cate_forward_nb201(golden, mname):
    g = golden(f"cate_fwd_nb201_{mname}")
    sd = synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), perturb_norms=True)
    rec = {}
    X, mask = T(g["X"]), T(g["mask"])
    out = O.cate_forward(sd, X, T(g["time_cond"]), mask.unsqueeze(0).repeat(X.shape[0], 1, 1), record=rec)
    close(out, g["out"], 1e-5, 1e-6)
    close(rec["attn0"], g["attn0"], 1e-5, 1e-7)
    close(rec["concat"], g["concat"], 1e-5, 1e-5)
    # a fully masked query row (the sink node, row 7 of the raw adjacency) attends uniformly
    if mname == "raw":
        assert np.allclose(g["attn0"][:, :, 7, :], 1.0 / 8.0)


def test_cate_forward_ofa(golden):
    g = golden("cate_fwd_ofa_chain")
    dims = CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
    sd = synthetic.cate_state_dict(dims, seed=int(g["seed"]), perturb_norms=True)
    X, mask = T(g["X"]), T(g["mask"])
    out = O.cate_forward(sd, X, T(g["time_cond"]), mask.unsqueeze(0).repeat(X.shape[0], 1, 1), stagewise_pos=True)
    close(out, g["out"], 1e-5, 1e-6)


@pytest.mark.parametrize("mname", ["raw", "closure"])
def test_predictor_and_guidance_gradient(golden, mname):
    g = golden(f"predictor_nb201_{mname}")
    sdp = synthetic.predictor_state_dict(PredictorDims(), seed=int(g["seed"]))
    task = synthetic.synthetic_task(10, 20, seed=int(g["task_seed"]))
    D_mu = O.set_encode(sdp, task)
    close(D_mu, g["D_mu"], 1e-5, 1e-6)
    X, mask = T(g["X"]), T(g["mask"])
    mB = mask.unsqueeze(0).repeat(X.shape[0], 1, 1)
    for i, tval in enumerate(g["t"]):
        t = torch.ones(X.shape[0]) * float(tval)
        y = O.predictor_forward(sdp, X, t * 999, mB, D_mu)
        close(y, g["y"][i], 1e-5, 1e-6)
        gr = O.classifier_grad(sdp, X, t, mB, D_mu, "max")
        close(gr, g["grad"][i], 1e-4, 1e-8)
        close(O.classifier_grad(sdp, X, t, mB, D_mu, "min"), -g["grad"][i], 1e-4, 1e-8)


def _run(g, guided, record=None, n_steps=None):
    sd = synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), preset=str(g["preset"]))
    sde = O.VPSDE(0.1, float(g["beta_max"]), int(g["sde_N"]))
    B = int(g["B"])
    tape = synthetic.noise_tape(int(g["n_run"]), (B, 8, 7), int(g["tape_seed"]))
    kw = {}
    if guided:
        sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
        kw = dict(pred_sd=sdp, D_mu=O.set_encode(sdp, synthetic.synthetic_task(10, 20, seed=2)),
                  classifier_scale=float(g["classifier_scale"]))
    return O.pc_sample(sd, sde, T(g["mask"]), tape, eps=float(g["eps"]), snr=float(g["snr"]),
                       n_steps=int(g["n_run"]) if n_steps is None else n_steps, record=record, **kw)


@pytest.mark.parametrize("name,guided", [("pc_uncond_nb201_n1000_first8", False),
                                         ("pc_guided_nb201_n1000_first8", True)])
def test_pc_first_steps_random_init(golden, name, guided):
    g = golden(name)
    rec = []
    _run(g, guided, rec)
    for i, r in enumerate(rec):
        sc = np.abs(g["x_pred"][i]).max()
        close(r["score_c"], g["score_c"][i], 2e-4, 2e-4 * np.abs(g["score_c"][i]).max())
        close(r["x_corr"], g["x_corr"][i], 0, 2e-4 * sc)
        close(r["x_pred"], g["x_pred"][i], 0, 2e-4 * sc)
        if guided:
            close(r["cgrad_c"], g["cgrad_c"][i], 1e-3, 1e-3 * np.abs(g["cgrad_c"][i]).max())


@pytest.mark.parametrize("name,guided", [("pc_uncond_nb201_contractive_n100", False),
                                         ("pc_guided_nb201_contractive_n100", True)])
def test_pc_full_trajectory_contractive(golden, name, guided):
    g = golden(name)
    rec = []
    final = _run(g, guided, rec)
    for i in (0, 1, 10, 50, 99):
        close(rec[i]["x_pred"], g["x_pred"][i], 0, 1e-4 * np.abs(g["x_pred"][i]).max())
        close(rec[i]["score_p"], g["score_p"][i], 0, 1e-4 * np.abs(g["score_p"][i]).max())
    close(final, g["final"], 0, 1e-4 * np.abs(g["final"]).max())
    assert torch.equal(O.quantize(final.clone()), O.quantize(T(g["final"]).clone()))


@pytest.mark.parametrize("name", ["pc_guided_nb201_nsteps2_n1000_first6", "pc_guided_nb201_nsteps2_contractive_n40"])
def test_pc_two_corrector_updates_per_step(golden, name):
    """n_steps_each = 2 (sampling.py:244): goldens from the reference's own loop (tests/golden/make_golden_nsteps.py)."""
    g = golden(name)
    CS, S, B = int(g["n_steps_each"]), int(g["n_run"]), int(g["B"])
    sd = synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), preset=str(g["preset"]))
    sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
    tape = synthetic.noise_tape(S * CS, (B, 8, 7), int(g["tape_seed"]))
    rec = []
    final = O.pc_sample(sd, O.VPSDE(0.1, 20.0, int(g["sde_N"])), T(g["mask"]), tape, eps=float(g["eps"]), snr=float(g["snr"]),
                        n_steps=S, record=rec, pred_sd=sdp, D_mu=O.set_encode(sdp, synthetic.synthetic_task(10, 20, seed=2)),
                        classifier_scale=float(g["classifier_scale"]), n_steps_each=CS)
    first = "first" in name
    tol = 2e-4
    for i in (range(S) if first else (0, 1, S // 2, S - 1)):
        assert len(rec[i]["corr"]) == CS
        for k in range(CS):
            close(rec[i]["corr"][k]["score_c"], g["score_c"][i * CS + k], 0, tol * np.abs(g["score_c"][i * CS + k]).max())
            close(rec[i]["corr"][k]["cgrad_c"], g["cgrad_c"][i * CS + k], 0, 1e-3 * np.abs(g["cgrad_c"][i * CS + k]).max())
        close(rec[i]["x_corr"], g["x_corr_last"][i], 0, tol * np.abs(g["x_corr_last"][i]).max())
        close(rec[i]["score_p"], g["score_p"][i], 0, tol * np.abs(g["score_p"][i]).max())
        close(rec[i]["x_pred"], g["x_pred"][i], 0, tol * np.abs(g["x_pred"][i]).max())
    if "final" in g:
        close(final, g["final"], 0, 1e-4 * np.abs(g["final"]).max())
        assert torch.equal(O.quantize(final.clone()), O.quantize(T(g["final"]).clone()))


def test_linspace_and_index_edges():
    sde = O.VPSDE(0.1, 20.0, 1000)
    ts = O.timesteps(sde, 1e-3)
    k = sde.timestep_index(ts)
    assert int(k[0]) == 999 and int(k[-1]) == 0
    assert (k[:-1] >= k[1:]).all()


def test_mask_helpers():
    adj = O.nb201_adj()
    assert torch.equal(adj, synthetic.nb201_adjacency())
    assert torch.equal(O.aug_mask(adj, "floyd"), adj)            # the reference's typo path: raw adjacency
    cl = O.aug_mask(adj, "floyed")
    assert torch.equal(cl, synthetic.transitive_closure(adj))
    assert cl[0, 7] == 1 and cl[0, 0] == 0 and cl.sum() > adj.sum()


def test_pc_full_n1000_guided_contractive(golden):
    """The real schedule length (N=1000, 2000 guided evaluations) end to end at B=4."""
    g = golden("pc_guided_nb201_contractive_n1000")
    final = _run(g, True)
    close(final, g["final"], 0, 1e-4 * np.abs(g["final"]).max())
    q, qr = O.quantize(final.clone()), O.quantize(T(g["final"]).clone())
    assert torch.equal(q, qr)


# ----------------------------------------------------------------------------------------------------
# OFA (20 x 9) surrogate MetaPredictorCATE and the guided OFA sampler (tests/golden/make_golden_ofa.py)
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mname", ["chain", "random"])
def test_predictor_cate_and_guidance_gradient(golden, mname):
    from diffusionnag_amd.specs import PredictorCateDims
    g = golden(f"predictor_ofa_{mname}")
    sdp = synthetic.predictor_cate_state_dict(PredictorCateDims(), seed=int(g["seed"]))
    task = synthetic.synthetic_task(10, 20, seed=int(g["task_seed"]))
    D_mu = O.set_encode(sdp, task)
    close(D_mu, g["D_mu"], 1e-5, 1e-6)
    X, mask = T(g["X"]), T(g["mask"])
    mB = mask.unsqueeze(0).repeat(X.shape[0], 1, 1)
    for i, tval in enumerate(g["t"]):
        t = torch.ones(X.shape[0]) * float(tval)
        close(O.predictor_cate_forward(sdp, X, t * 999, mB, D_mu), g["y"][i], 1e-5, 1e-6)
        close(O.classifier_grad_cate(sdp, X, t, mB, D_mu, "max"), g["grad"][i], 1e-4, 2e-6 * np.abs(g["grad"][i]).max())


@pytest.mark.parametrize("name", ["pc_guided_ofa_n1000_first6", "pc_guided_ofa_contractive_n50"])
def test_pc_guided_ofa(golden, name):
    from diffusionnag_amd.specs import PredictorCateDims
    g = golden(name)
    dims = CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
    sd = synthetic.cate_state_dict(dims, seed=int(g["seed"]), preset=str(g["preset"]))
    sdp = synthetic.predictor_cate_state_dict(PredictorCateDims(), seed=int(g["pred_seed"]))
    sde = O.VPSDE(0.1, float(g["beta_max"]), int(g["sde_N"]))
    B, S = int(g["B"]), int(g["n_run"])
    tape = synthetic.noise_tape(S, (B, 20, 9), int(g["tape_seed"]))
    rec = []
    final = O.pc_sample(sd, sde, T(g["mask"]), tape, eps=float(g["eps"]), snr=float(g["snr"]), n_steps=S, pred_sd=sdp,
                        D_mu=O.set_encode(sdp, synthetic.synthetic_task(10, 20, seed=2)),
                        classifier_scale=float(g["classifier_scale"]), record=rec, model_kw=dict(stagewise_pos=True))
    for i in sorted({0, 1, S // 2, S - 1}):
        sc = np.abs(g["x_pred"][i]).max()
        close(rec[i]["x_pred"], g["x_pred"][i], 0, 2e-4 * sc)
        close(rec[i]["score_p"], g["score_p"][i], 0, 2e-4 * np.abs(g["score_p"][i]).max())
        close(rec[i]["cgrad_c"], g["cgrad_c"][i], 0, 1e-3 * np.abs(g["cgrad_c"][i]).max())
    if "final" in g:
        close(final, g["final"], 0, 2e-4 * np.abs(g["final"]).max())
        assert torch.equal(O.quantize(final.clone()), O.quantize(T(g["final"]).clone()))


# ----------------------------------------------------------------------------------------------------
# Sampler variants: VESDE, Euler-Maruyama / None predictors, None corrector (tests/golden/make_golden_ve.py)
# ----------------------------------------------------------------------------------------------------
VARIANTS = ["pc_guided_nb201_ve_em_n1000_first8", "pc_uncond_nb201_ve_rd_contractive_n60",
            "pc_guided_nb201_ve_em_contractive_n60", "pc_uncond_nb201_vp_em_nonecorr_contractive_n60",
            "pc_uncond_nb201_vp_nonepred_n1000_first6"]


def variant_sde(g, mod):
    N = int(g["sde_N"])
    return mod.VESDE(0.1, 5.0, N) if str(g["sde_kind"]) == "ve" else mod.VPSDE(0.1, 20.0, N)


def variant_tape(g):
    B, S = int(g["B"]), int(g["n_run"])
    tape = synthetic.noise_tape(S, (B, 8, 7), int(g["tape_seed"]))
    tape["x0"] = T(g["x0"])                       # prior_sampling output (randn * sigma_max for the VESDE)
    return tape


@pytest.mark.parametrize("name", VARIANTS)
def test_pc_sampler_variants(golden, name):
    g = golden(name)
    guided = bool(int(g["guided"]))
    sd = synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), preset=str(g["preset"]))
    kw = {}
    if guided:
        sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
        kw = dict(pred_sd=sdp, D_mu=O.set_encode(sdp, synthetic.synthetic_task(10, 20, seed=2)),
                  classifier_scale=float(g["classifier_scale"]))
    rec = []
    S = int(g["n_run"])
    final = O.pc_sample(sd, variant_sde(g, O), T(g["mask"]), variant_tape(g), eps=float(g["eps"]), snr=float(g["snr"]),
                        n_steps=S, record=rec, predictor=str(g["predictor"]), corrector=str(g["corrector"]), **kw)
    for i in sorted({0, 1, S // 2, S - 1}):
        sc = np.abs(g["x_pred"][i]).max()
        close(rec[i]["x_pred"], g["x_pred"][i], 0, 2e-4 * sc)
        close(rec[i]["x_mean"], g["x_mean"][i], 0, 2e-4 * sc)
        if "score_p" in g:
            close(rec[i]["score_p"], g["score_p"][i], 0, 2e-4 * np.abs(g["score_p"][i]).max())
        if "x_corr" in g:
            close(rec[i]["x_corr"], g["x_corr"][i], 0, 2e-4 * sc)
        if "cgrad_c" in g:
            close(rec[i]["cgrad_c"], g["cgrad_c"][i], 0, 1e-3 * np.abs(g["cgrad_c"][i]).max())
    if "final" in g:
        close(final, g["final"], 0, 2e-4 * np.abs(g["final"]).max())
        assert torch.equal(O.quantize(final.clone()), O.quantize(T(g["final"]).clone()))


# ----------------------------------------------------------------------------------------------------
# DSM training step: loss and parameter gradients (tests/golden/make_golden_train.py)
# ----------------------------------------------------------------------------------------------------
def train_case(g):
    """Inputs of a training golden, regenerated from its seeds exactly as make_golden_train.py drew them."""
    B, p = int(g["B"]), float(g["p"])
    rs = np.random.RandomState(int(g["input_seed"]))
    ops = rs.randint(0, 7, size=(B, 8))
    x = np.zeros((B, 8, 7), np.float32)
    x[np.arange(B)[:, None], np.arange(8)[None, :], ops] = 1.0
    x = 2.0 * x - 1.0
    u = rs.uniform(0, 1, size=B).astype(np.float32)
    z = rs.standard_normal((B, 8, 7)).astype(np.float32)
    eps = float(g["eps"])
    t = T(u) * (1 - eps) + eps                                      # losses.py:139
    dropout = None
    if p > 0:
        rs = np.random.RandomState(int(g["mask_seed"]))
        dropout = {"p": p, "masks": {k: T((rs.uniform(0, 1, size=shape) >= p).astype(np.float32))
                                     for k, shape in O.dropout_sites(B, 8).items()}}
    return T(x), t, T(z), dropout


@pytest.mark.parametrize("name", ["train_nb201_vp_nodrop", "train_nb201_vp_drop", "train_nb201_ve_drop"])
def test_dsm_loss_and_parameter_gradients(golden, name):
    g = golden(name)
    x, t, z, dropout = train_case(g)
    sd = synthetic.cate_state_dict(CateDims(), seed=0, perturb_norms=True)
    sde = O.VPSDE(0.1, 20.0, 1000) if str(g["sde_kind"]) == "vp" else O.VESDE(0.1, 5.0, 1000)
    maskB = T(g["mask"]).unsqueeze(0).repeat(x.shape[0], 1, 1)
    loss, grads = O.dsm_loss_and_grads(sd, sde, x, t, z, maskB, dropout=dropout)
    close(loss, g["loss"], 1e-5, 0)
    names = list(sd.keys())
    norms = np.array([float(grads[k].double().norm()) for k in names])
    close(norms, g["grad_norms"], 1e-4, 1e-7)
    for key in g:
        if key.startswith("grad:"):
            k = key[5:]
            got = grads[k[:-4]][:8] if k.endswith("[:8]") else grads[k]
            close(got, g[key], 0, 2e-5 * np.abs(g[key]).max() + 1e-9)


# ----------------------------------------------------------------------------------------------------
# Exit of a sampling round: validity rules and strings (tests/golden/make_golden_validity.py)
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["validity_nb201_a", "validity_nb201_b"])
def test_validity_rules_and_strings(golden, name):
    g = golden(name)
    q = [(xi >= 0.5).astype(np.float32) for xi in g["x"]]             # utils.py:254-255
    errs, strs = O.compute_validity(q)
    assert errs == list(g["error_types"])
    assert [s if s is not None else "" for s in strs] == list(g["all_arch_str"])
    valid = [s for s in strs if s is not None]
    assert len(valid) / len(strs) == float(g["validity"])
    assert len(set(valid)) == int(g["n_unique"])


# ----------------------------------------------------------------------------------------------
# OFA validity rules and strings: PARITY UNPINNED (the reference module imports wandb and igraph, not installed):
# hand-derived known answers from the text of MobileNetV3/analysis/arch_functions.py:135-144, :374-429
# ----------------------------------------------------------------------------------------------
def _ofa_x(depths, types=None):
    x = np.zeros((20, 9), np.float32)
    for s, d in enumerate(depths):
        for j in range(d):
            x[4 * s + j, (types or {}).get(4 * s + j, 0)] = 1.0
    return x


def test_ofa_validity_known_answers():
    ok = _ofa_x([2, 3, 4, 2, 3], {0: 8, 5: 4, 10: 2})
    assert O.is_valid_ofa_x(ok) == (True, -1)
    s = O.ofa_string_from_onehot_x(ok).split('_')
    assert len(s) == 20
    assert s[0] == '2-7-6' and s[1] == '2-3-3' and s[2] == '0-0-0' and s[3] == '0-0-0'      # stage 0: depth 2, type 8 = '7-6'
    assert s[4] == '3-3-3' and s[5] == '3-5-4' and s[6] == '3-3-3' and s[7] == '0-0-0'
    assert s[8] == '4-3-3' and s[10] == '4-3-6' and s[11] == '4-3-3'
    multi = ok.copy(); multi[9, 3] = 1.0                                                     # slot 9 already has type 0
    assert O.is_valid_ofa_x(multi) == (False, 1)
    no2 = ok.copy(); no2[13] = 0.0                                                           # second layer of stage 3
    assert O.is_valid_ofa_x(no2) == (False, 2)
    gap = ok.copy(); gap[2, 1] = 0.0; gap[3, 1] = 1.0                                        # stage 0: layer 4 without layer 3
    assert O.is_valid_ofa_x(gap) == (False, 3)
    both = multi.copy(); both[13] = 0.0                                                      # the first failing rule wins
    assert O.is_valid_ofa_x(both) == (False, 1)
    errs, strs = O.compute_validity_ofa([ok, multi, no2, gap])
    assert errs == [-1, 1, 2, 3] and strs[1:] == [None, None, None] and strs[0] == '_'.join(s)
