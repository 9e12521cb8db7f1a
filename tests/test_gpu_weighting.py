"""Guidance weighting options of the MobileNetV3 tree's sampler (MobileNetV3/sampling.py:858-910) on the HIP path.

PARITY UNPINNED: that file imports torchdiffeq, which is not installed, so the reference cannot be run for these
options.  The checks are (i) against the oracle's restatement of that code (oracle/dnag_oracle.py: pc_sample(weighting=)),
and (ii) internal consistency with quantities that ARE pinned by the reference's goldens: the recorded score and
guidance gradient of every step must recombine into the recorded total gradient by the option's formula.
Needs an MI355X."""

import numpy as np
import pytest
import torch

from diffusionnag_amd import cate as _cate, digcn_meta as _dm  # noqa: F401
from diffusionnag_amd import mutils, sampling, sde_lib, synthetic, utils
from diffusionnag_amd.specs import CateDims, PredictorDims
from oracle import dnag_oracle as O
from tests.helpers import predictor_config, score_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, S = 5, 6


def run(weighting_cfg, extra, scale=50.0, eps=1e-3):
    cfg = score_config(CateDims(), DEV)
    for k, v in weighting_cfg.items():
        cfg.sampling[k] = v
    sd = synthetic.cate_state_dict(CateDims(), seed=0, preset="contractive")
    sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
    model = mutils.create_model(cfg)
    model.load_state_dict(sd)
    clf = mutils.create_model(predictor_config(PredictorDims(), DEV))
    clf.load_state_dict(sdp)
    task = synthetic.synthetic_task(10, 20, seed=2)
    mask = utils.aug_mask(utils.nb201_adj())[0]
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), eps, conditional=True, **extra)
    tape = synthetic.noise_tape(S, (B, 8, 7), 17)
    fn.noise_tape = {k: v.to(DEV) for k, v in tape.items()}
    fn.max_steps, fn.record = S, True
    out = fn(model.eval(), mask, clf.eval(), classifier_scale=scale, task=task)
    rec = []
    wo = dict(weighting_cfg, **extra)
    ref = O.pc_sample(sd, O.VPSDE(0.1, 20.0, 1000), mask, tape, eps=eps, n_steps=S, pred_sd=sdp, D_mu=O.set_encode(sdp, task),
                      classifier_scale=scale, record=rec, weighting=wo)
    return out.cpu(), fn.last_record, ref, rec, fn


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("wcfg,extra", [
    (dict(weight_scheduling=True, weight_ratio=False, t_spot=1.0, t_spot_end=None), {}),
    (dict(weight_scheduling=True, weight_ratio=True, t_spot=1.0, t_spot_end=None), {}),
    (dict(weight_scheduling=False, weight_ratio=True, t_spot=1.0, t_spot_end=None), dict(p=2, prod_w=True)),
    (dict(weight_scheduling=False, weight_ratio=False, t_spot=1.0, t_spot_end=None), dict(weight_ratio_abs=True)),
])
def test_weighting_options_match_oracle_restatement(wcfg, extra):
    out, r, ref, rec, fn = run(wcfg, extra)
    for i in range(S):
        assert rel(r["rec_total_c"][i].cpu(), rec[i]["total_c"]) <= 2e-5, i
        assert rel(r["rec_total_p"][i].cpu(), rec[i]["total_p"]) <= 2e-5, i
        assert rel(r["rec_x_pred"][i].cpu(), rec[i]["x_pred"]) <= 1e-5, i
    assert rel(out, ref) <= 1e-5
    # recombination from the kernel's own recorded score and guidance gradient (pinned quantities)
    ts = torch.linspace(1, 1e-3, 1000)
    for i in range(S):
        s, cg = r["rec_score_p"][i].cpu().double(), r["rec_cgrad_p"][i].cpu().double()
        w = 50.0 * 0.1 ** ts[i].item() if wcfg["weight_scheduling"] else 50.0
        if wcfg["weight_ratio"]:
            p = extra.get("p", 1)
            den = (w * cg) if extra.get("prod_w") else cg
            w = w * (s.reshape(B, -1).norm(p=p, dim=-1) / den.reshape(B, -1).norm(p=p, dim=-1))[:, None, None]
        if extra.get("weight_ratio_abs"):
            w = w * s.abs() / cg.abs()
        assert rel(r["rec_total_p"][i].cpu(), s + w * cg) <= 1e-5, i


def test_t_spot_window_switches_guidance_off_outside():
    """t_spot = 0.9985 with the grid 1, 0.999, 0.998, ...: steps 0 and 1 are outside the window (plain score, and the
    surrogate is not even evaluated), later steps are guided."""
    wcfg = dict(weight_scheduling=False, weight_ratio=False, t_spot=0.9985, t_spot_end=None)
    out, r, ref, rec, fn = run(wcfg, {})
    for i in range(S):
        guided = i >= 2
        tot, sc = r["rec_total_p"][i].cpu(), r["rec_score_p"][i].cpu()
        assert torch.equal(tot, sc) != guided
        assert rel(r["rec_x_pred"][i].cpu(), rec[i]["x_pred"]) <= 1e-5, i
        if not guided:
            assert float(r["rec_cgrad_p"][i].abs().max()) == 0.0          # never written: the surrogate did not run
    assert rel(out, ref) <= 1e-5


def test_options_off_is_the_nb201_sampler():
    out0, r0, _, _, _ = run({}, {})
    out1, r1, _, _, _ = run(dict(weight_scheduling=False, weight_ratio=False, t_spot=1.0, t_spot_end=None), {})
    assert torch.equal(out0, out1)
