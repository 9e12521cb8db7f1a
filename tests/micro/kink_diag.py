"""Which relu units explain a guidance-gradient mismatch of one graph?  Runs the B=8192 rank-0 case of
tests/test_gpu_parity.py::test_full_batch_guided_steps_match_oracle, takes graph G of PC step 1 and tries every
single / pair flip of the near-zero pre-activations of the oracle (fp32 and fp64)."""
import itertools, sys
from unittest import mock
import numpy as np, torch
sys.path.insert(0, ".")
from diffusionnag_amd import sampling, sde_lib, synthetic, utils
from diffusionnag_amd.specs import CateDims, PredictorDims
from tests.helpers import score_config
from tests.test_gpu_parity import make_score, make_pred, DEV
from oracle import dnag_oracle as O

B, S, G = 8192, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1275
cfg = score_config(CateDims(), DEV)
sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
model, clf = make_score(CateDims(), 0), make_pred(1)
sde = sde_lib.VPSDE(0.1, 20.0, 1000)
fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), 1e-3, conditional=True)
tape = synthetic.noise_tape(S, (B, 8, 7), 21)
fn.noise_tape = {k: v.to(DEV) for k, v in tape.items()}
fn.max_steps, fn.record = S, True
mask = utils.aug_mask(utils.nb201_adj())[0]
task = synthetic.synthetic_task(10, 20, seed=2)
fn(model, mask, clf, classifier_scale=10000.0, task=task)
r = {k: v.cpu() for k, v in fn.last_record.items()}
osde = O.VPSDE(0.1, 20.0, 1000)
ts = O.timesteps(osde, 1e-3)
D_mu = O.set_encode(sdp, task)
x = r["rec_x_pred"][0][G:G + 1]
ours = r["rec_cgrad_c"][1][G]
vt = torch.ones(1) * ts[1]
m1 = mask.unsqueeze(0)
pre = []
relu0 = O.F.relu
def rec(v):
    pre.append(v.detach().clone()); return relu0(v)
with mock.patch.object(O.F, "relu", rec):
    cg = O.classifier_grad(sdp, x, vt, m1, D_mu)[0]
print("graph", G, "err vs oracle", float((ours - cg).abs().max()), "max|cg|", float(cg.abs().max()))
cands = []
for k, v in enumerate(pre):
    rel = (v.abs() / v.abs().max()).flatten()
    idx = torch.nonzero(rel < 2e-5).flatten().tolist()
    for i in idx:
        cands.append((k, i, float(v.flatten()[i]), float(rel[i])))
print("near-zero pre-activations (call, flat index, value, |v|/max):", cands)
torch.set_default_dtype(torch.float64)
pre64 = []
def rec64(v):
    pre64.append(v.detach().clone()); return relu0(v)
with mock.patch.object(O.F, "relu", rec64):
    cg64 = O.classifier_grad({k: v.double() for k, v in sdp.items()}, x.double(), vt.double(), m1.double(), D_mu.double())[0]
torch.set_default_dtype(torch.float32)
print("fp64 values of the candidates:", [(k, i, float(pre64[k].flatten()[i])) for k, i, _, _ in cands])
print("err vs fp64 oracle", float((ours.double() - cg64).abs().max()))
best = None
for n in (1, 2, 3):
    for sub in itertools.combinations(range(len(cands)), n):
        calls = [0]
        def relu(v, sub=sub, calls=calls):
            on = v > 0
            f = on.flatten()
            for j in sub:
                if cands[j][0] == calls[0]:
                    f[cands[j][1]] = ~f[cands[j][1]]
            calls[0] += 1
            return v * f.view(v.shape).to(v.dtype)
        with mock.patch.object(O.F, "relu", relu):
            c2 = O.classifier_grad(sdp, x, vt, m1, D_mu)[0]
        e = float((ours - c2).abs().max())
        if best is None or e < best[0]:
            best = (e, [cands[j][:2] for j in sub])
print("best flip set:", best)
