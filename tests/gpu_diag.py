"""Stage-by-stage diagnostic of the HIP path against the committed golden vectors and the oracle.
Prints error tables and never asserts: `python tests/gpu_diag.py` on a GPU box."""

import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from diffusionnag_amd import mutils, sampling, sde_lib, synthetic, utils  # noqa: E402
from diffusionnag_amd import cate as _c, digcn_meta as _d  # noqa: E402,F401
from diffusionnag_amd.specs import CateDims, PredictorDims  # noqa: E402
from tests.helpers import Cfg, score_config, predictor_config, golden  # noqa: E402
from oracle import dnag_oracle as O  # noqa: E402

dev = torch.device("cuda:0")


def err(name, a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    e = np.abs(a - b)
    print(f"  {name:28s} max|err| {e.max():.3e}  ref max {np.abs(b).max():.3e}  rel {e.max() / (np.abs(b).max() + 1e-30):.3e}"
          f"  nan {int(np.isnan(a).sum())}", flush=True)


def main():
    torch.manual_seed(0)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    # ---- CATE forward, NB201 -------------------------------------------------------------------
    for mname in ("raw", "closure"):
        g = golden(f"cate_fwd_nb201_{mname}")
        print(f"[cate_fwd_nb201_{mname}]")
        model = mutils.create_model(score_config(CateDims(), dev))
        model.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), perturb_norms=True))
        model.eval()
        X = torch.from_numpy(g["X"]).to(dev)
        tc = torch.from_numpy(g["time_cond"]).to(dev)
        m = torch.from_numpy(g["mask"]).to(dev)
        out, concat, attn0 = model(X, tc, m.unsqueeze(0).repeat(X.shape[0], 1, 1), return_concat=True, return_attn0=True)
        torch.cuda.synchronize()
        err("attn0", attn0, g["attn0"])
        for l in range(12):
            err(f"concat layer {l}", concat[..., l * 64:(l + 1) * 64], g["concat"][..., l * 64:(l + 1) * 64])
        err("out", out, g["out"])
    # ---- CATE forward, OFA ----------------------------------------------------------------------
    g = golden("cate_fwd_ofa_chain")
    print("[cate_fwd_ofa_chain]")
    dims = CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
    model = mutils.create_model(score_config(dims, dev))
    model.load_state_dict(synthetic.cate_state_dict(dims, seed=int(g["seed"]), perturb_norms=True))
    model.eval()
    X = torch.from_numpy(g["X"]).to(dev)
    out = model(X, torch.from_numpy(g["time_cond"]).to(dev), torch.from_numpy(g["mask"]).to(dev).unsqueeze(0).repeat(3, 1, 1))
    torch.cuda.synchronize()
    err("out", out, g["out"])
    # ---- surrogate ------------------------------------------------------------------------------
    for mname in ("raw", "closure"):
        g = golden(f"predictor_nb201_{mname}")
        print(f"[predictor_nb201_{mname}]")
        clf = mutils.create_model(predictor_config(PredictorDims(), dev))
        clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), seed=int(g["seed"])))
        clf.eval()
        task = synthetic.synthetic_task(10, 20, seed=int(g["task_seed"])).to(dev)
        X = torch.from_numpy(g["X"]).to(dev)
        m = torch.from_numpy(g["mask"]).to(dev).unsqueeze(0).repeat(X.shape[0], 1, 1)
        clf.sample_state, clf.D_mu = True, None
        for i, tval in enumerate(g["t"]):
            t = torch.ones(X.shape[0], device=dev) * float(tval)
            y = clf(X, t * 999, m, task=task.repeat(X.shape[0], 1, 1))
            gr = clf.input_grad(X, t * 999, m, task=task)
            torch.cuda.synchronize()
            if i == 0:
                err("D_mu", clf.D_mu[0], g["D_mu"])
            err(f"y t={tval:.3f}", y, g["y"][i])
            err(f"grad t={tval:.3f}", gr, g["grad"][i])
        clf.sample_state = False
    # ---- sampler, first steps ---------------------------------------------------------------------
    for name, guided in (("pc_uncond_nb201_n1000_first8", False), ("pc_guided_nb201_n1000_first8", True),
                         ("pc_uncond_nb201_contractive_n100", False), ("pc_guided_nb201_contractive_n100", True)):
        g = golden(name)
        print(f"[{name}]")
        B, S = int(g["B"]), int(g["n_run"])
        cfg = score_config(CateDims(), dev)
        model = mutils.create_model(cfg)
        model.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), preset=str(g["preset"])))
        sde = sde_lib.VPSDE(0.1, float(g["beta_max"]), int(g["sde_N"]))
        fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), float(g["eps"]), conditional=guided)
        tape = synthetic.noise_tape(S, (B, 8, 7), int(g["tape_seed"]))
        fn.noise_tape = {k: v.to(dev) for k, v in tape.items()}
        fn.max_steps = S
        fn.record = True
        mask = torch.from_numpy(g["mask"])
        t0 = time.time()
        if guided:
            clf = mutils.create_model(predictor_config(PredictorDims(), dev))
            clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), seed=1))
            final = fn(model, mask, clf, classifier_scale=float(g["classifier_scale"]), task=synthetic.synthetic_task(10, 20, seed=2))
        else:
            final = fn(model, mask.unsqueeze(0))[0]
        torch.cuda.synchronize()
        print(f"  {S} steps in {time.time() - t0:.3f} s")
        r = fn.last_record
        for i in sorted(set([0, 1, S // 2, S - 1])):
            err(f"score_c[{i}]", r["rec_score_c"][i], g["score_c"][i])
            if guided:
                err(f"cgrad_c[{i}]", r["rec_cgrad_c"][i], g["cgrad_c"][i])
            err(f"x_corr[{i}]", r["rec_x_corr"][i], g["x_corr"][i])
            err(f"score_p[{i}]", r["rec_score_p"][i], g["score_p"][i])
            err(f"x_mean[{i}]", r["rec_x_mean"][i], g["x_mean"][i])
            err(f"x_pred[{i}]", r["rec_x_pred"][i], g["x_pred"][i])
        print("  step sizes:", r["rec_step_size"][:3].cpu().numpy().tolist())
        if "final" in g:
            err("final", final, g["final"])
            q = np.stack(utils.quantize(final))
            qr = (g["final"] >= 0.5).astype(np.float32)
            print("  quantised mismatches:", int((q != qr).sum()), "of", q.size)
    # ---- random batch vs oracle (ragged sizes, per-graph time and mask) ---------------------------
    print("[random batch vs oracle, B=37]")
    rs = np.random.RandomState(5)
    B = 37
    sd = synthetic.cate_state_dict(CateDims(), seed=4, perturb_norms=True)
    model = mutils.create_model(score_config(CateDims(), dev))
    model.load_state_dict(sd)
    model.eval()
    X = torch.from_numpy(rs.standard_normal((B, 8, 7)).astype(np.float32))
    tc = torch.from_numpy((rs.uniform(0.001, 1, B) * 999).astype(np.float32))
    masks = torch.from_numpy((rs.uniform(0, 1, (B, 8, 8)) < 0.4).astype(np.float32))
    ref = O.cate_forward(sd, X, tc, masks)
    out = model(X.to(dev), tc.to(dev), masks.to(dev))
    err("out (per-graph t, mask)", out, ref)
    # ---- timing of one full-size evaluation ---------------------------------------------------------
    print("[timing, B=4096 forward]")
    B = 4096
    X = torch.randn(B, 8, 7, device=dev)
    tcs = torch.full((1,), 500.0, device=dev)
    m = utils.nb201_adj().to(dev).unsqueeze(0)
    for _ in range(2):
        model(X, tcs, m)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        model(X, tcs, m)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    print(f"  {dt * 1e3:.3f} ms / eval  -> {63.36e6 * B / dt / 1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
