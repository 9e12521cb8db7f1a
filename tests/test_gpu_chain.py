"""Chain recording of the MobileNetV3 tree's sampler (MobileNetV3/sampling.py:980-988, :1030-1034) on the HIP path.

PARITY UNPINNED against the reference itself: that file imports torchdiffeq, which is not installed, so it cannot be
run here.  The checks are (i) bit-exact consistency with the per-step records of the same run, which ARE pinned by the
reference's trajectory goldens (tests/test_gpu_parity.py): slot w must hold inverse_scaler(x_mean) of the LAST step that
maps to it, (ii) the oracle's restatement of that code (oracle/dnag_oracle.py: pc_sample(chain=)).  Needs an MI355X."""

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
B, NSDE = 5, 48


def run(eval_chain, max_steps=None, **chain_kw):
    cfg = score_config(CateDims(), DEV)
    sd = synthetic.cate_state_dict(CateDims(), seed=0, preset="contractive")
    sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
    model = mutils.create_model(cfg)
    model.load_state_dict(sd)
    clf = mutils.create_model(predictor_config(PredictorDims(), DEV))
    clf.load_state_dict(sdp)
    task = synthetic.synthetic_task(10, 20, seed=2)
    mask = utils.aug_mask(utils.nb201_adj())[0]
    sde = sde_lib.VPSDE(0.1, 20.0, NSDE)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), 1e-3, conditional=True)
    tape = synthetic.noise_tape(NSDE, (B, 8, 7), 17)
    fn.noise_tape = {k: v.to(DEV) for k, v in tape.items()}
    fn.record = True
    fn.max_steps = max_steps
    ret = fn(model.eval(), mask, clf.eval(), classifier_scale=50.0, task=task, eval_chain=eval_chain, **chain_kw)
    S = NSDE if max_steps is None else max_steps
    ch = dict(number_chain_steps=chain_kw.get("number_chain_steps"), keep_chain=chain_kw.get("keep_chain"), denoise=True) if eval_chain else None
    ref = O.pc_sample(sd, O.VPSDE(0.1, 20.0, NSDE), mask, tape, eps=1e-3, n_steps=S, pred_sd=sdp, D_mu=O.set_encode(sdp, task),
                      classifier_scale=50.0, chain=ch)
    return ret, fn, ref, ch


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("ncs,keep", [(4, 3), (None, None), (48, 1), (7, 5)])
def test_chain_slots_hold_the_last_step_that_maps_to_them(ncs, keep):
    (x, n_evals, chain, norms), fn, ref, ch = run(True, number_chain_steps=ncs, keep_chain=keep)
    n, k = ncs or NSDE, keep or B
    assert n_evals == NSDE * 2 and chain.device.type == "cpu" and tuple(chain.shape) == (n, k, 8, 7)
    inv = fn.inverse_scaler
    xm = fn.last_record["rec_x_mean"]
    last = {}
    for i in range(NSDE):
        last[n - 1 - (i * n) // NSDE] = i                                   # later steps overwrite (sampling.py:1032-1034)
    assert sorted(last) == list(range(n))
    for w, i in last.items():
        assert torch.equal(chain[w], inv(xm[i])[:k].cpu()), (w, i)         # bit-exact: a device copy of the recorded state
    assert torch.equal(chain[0], x[:k].cpu())                               # slot 0 = the returned samples
    # oracle restatement (5e-6, the single-evaluation tolerance: measured 4e-7 .. 5e-7, the contractive preset does not amplify)
    assert tuple(ch["chain_x"].shape) == tuple(chain.shape)
    print(f"chain vs oracle {rel(chain, ch['chain_x']):.1e}, samples {rel(x.cpu(), ref):.1e}")
    assert rel(chain, ch["chain_x"]) <= 5e-6
    assert rel(x.cpu(), ref) <= 5e-6


def test_truncated_run_leaves_unreached_slots_zero():
    (x, _, chain, _), fn, ref, ch = run(True, max_steps=20, number_chain_steps=6, keep_chain=2)
    reached = {6 - 1 - (i * 6) // NSDE for i in range(20)}
    for w in range(6):
        if w in reached:
            assert float(chain[w].abs().max()) > 0
        else:
            assert float(chain[w].abs().max()) == 0.0
    assert rel(chain, ch["chain_x"]) <= 5e-6


def test_eval_chain_false_is_the_mobilenet_tuple_and_none_is_the_nb201_return():
    (x, n_evals, chain, norms), fn, ref, _ = run(False)
    assert chain is None and n_evals == NSDE * 2 and len(norms) == 4
    y, fn2, _, _ = run(None)
    assert torch.is_tensor(y) and torch.equal(x, y)
