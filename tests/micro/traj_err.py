"""Per-step error of a golden trajectory against the reference's record (debug aid; prints instead of
asserting): python tests/micro/traj_err.py <golden name> [repeat]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests import test_gpu_parity as T
from tests.helpers import golden, score_config
from diffusionnag_amd import sampling, sde_lib, synthetic, utils
from diffusionnag_amd.specs import CateDims

name = sys.argv[1]
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = golden(name)
DEV = T.DEV


def run():
    guided = bool(int(g["guided"]))
    B, S, N = int(g["B"]), int(g["n_run"]), int(g["sde_N"])
    cfg = score_config(CateDims(), DEV)
    if "predictor" in g: cfg.sampling["predictor"], cfg.sampling["corrector"] = str(g["predictor"]), str(g["corrector"])
    model = T.make_score(CateDims(), int(g["seed"]), preset=str(g["preset"]))
    bmax = float(g["beta_max"]) if "beta_max" in g else 20.0
    sde = sde_lib.VESDE(0.1, 5.0, N) if ("sde_kind" in g and str(g["sde_kind"]) == "ve") else sde_lib.VPSDE(0.1, bmax, N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), float(g["eps"]), conditional=guided)
    tape = synthetic.noise_tape(S, (B, 8, 7), int(g["tape_seed"]))
    tape["x0"] = torch.from_numpy(g["x0"] if "x0" in g else g["x_in"][0] if g["x_in"].ndim == 4 else g["x_in"])
    fn.noise_tape = {k: v.to(DEV) for k, v in tape.items()}
    fn.max_steps, fn.record = S, True
    mask = torch.from_numpy(g["mask"])
    if guided:
        fn(model, mask, T.make_pred(1), classifier_scale=float(g["classifier_scale"]),
           task=synthetic.synthetic_task(10, 20, seed=2))
    else:
        fn(model, mask.unsqueeze(0))
    return {k: v.detach().cpu().numpy().astype(np.float64) for k, v in fn.last_record.items()}, S


prev = None
for it in range(rep):
    r, S = run()
    for key in ("x_corr", "score_c", "cgrad_c", "score_p", "x_mean", "x_pred"):
        if key not in g:
            continue
        row = []
        for i in range(S):
            a, b = r["rec_" + key][i], np.asarray(g[key][i], np.float64)
            row.append(np.abs(a - b).max() / np.abs(b).max())
        print(f"run{it} {key:8s}", " ".join(f"{e:.1e}" for e in row))
    if prev is not None:
        same = all(np.array_equal(prev[k], r[k]) for k in r)
        print(f"run{it} bitwise equal to previous run: {same}")
    prev = r
