"""Capture golden vectors of the sampler VARIANTS from the REFERENCE implementation (build container
only):  VESDE (the reference's shipped default SDE), EulerMaruyamaPredictor (shipped default
predictor), NonePredictor and NoneCorrector.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ve.py

Same method as make_golden.py: imports ``/root/reference/NAS-Bench-201`` unmodified, loads the seeded
synthetic weights into the reference's own modules, drives the reference's ``get_sampling_fn``
closures with an injected noise tape and records per-step tensors by wrapping -- in this process
only -- ``torch.randn_like``, ``get_score_fn`` / ``get_classifier_grad_fn`` and the ``update_fn`` of
the registered predictor / corrector classes.  Only inputs, seeds and outputs are stored.
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
from diffusionnag_amd.specs import CateDims, PredictorDims  # noqa: E402
from make_golden import predictor_config, score_config  # noqa: E402


def main():
    scratch = tempfile.mkdtemp(prefix="dnag_golden_")
    os.makedirs(os.path.join(scratch, "data", "transfer_nag"))
    torch.save([[torch.randn(30, 512)] for _ in range(10)],
               os.path.join(scratch, "data", "transfer_nag", "cifar10bylabel.pt"))
    os.chdir(scratch)
    sys.path.insert(0, REF)

    import sde_lib                              # reference
    import sampling                             # reference
    import models.utils as mutils               # reference
    from models import cate, digcn_meta         # noqa: F401
    import datasets_nas                         # reference

    nb, pd = CateDims(), PredictorDims()
    adj = synthetic.nb201_adjacency()
    closure = synthetic.transitive_closure(adj)
    clf = mutils.create_model(predictor_config(pd))
    clf.load_state_dict(synthetic.predictor_state_dict(pd, seed=1), strict=True)
    clf.eval()
    task = synthetic.synthetic_task(10, 20, seed=2)
    pred_cls = {"reverse_diffusion": sampling.ReverseDiffusionPredictor, "euler_maruyama": sampling.EulerMaruyamaPredictor,
                "none": sampling.NonePredictor}
    corr_cls = {"langevin": sampling.LangevinCorrector, "none": sampling.NoneCorrector}

    def run(tag, *, sde_kind, predictor, corrector, sde_N, n_run, B, guided, mask, preset, tape_seed, eps,
            classifier_scale=100.0):
        sde = (sde_lib.VESDE(sigma_min=0.1, sigma_max=5.0, N=sde_N) if sde_kind == "ve"
               else sde_lib.VPSDE(beta_min=0.1, beta_max=20.0, N=sde_N))
        cfg = score_config(nb)
        cfg.sampling["predictor"], cfg.sampling["corrector"] = predictor, corrector
        score_model = mutils.create_model(cfg)
        score_model.load_state_dict(synthetic.cate_state_dict(nb, seed=0, preset=preset), strict=True)
        tape = synthetic.noise_tape(n_run, (B, 8, 7), tape_seed)
        x0 = tape["x0"] * (5.0 if sde_kind == "ve" else 1.0)           # prior_sampling: randn * sigma_max (sde_lib.py:285)
        queue = []
        for i in range(n_run):
            if corrector != "none":
                queue.append(tape["corrector"][i])
            if predictor != "none":
                queue.append(tape["predictor"][i])
        calls = {"score": [], "cgrad": [], "corr": [], "pred": []}
        P, Cc = pred_cls[predictor], corr_cls[corrector]
        orig = (torch.randn_like, mutils.get_score_fn, mutils.get_classifier_grad_fn, Cc.update_fn, P.update_fn)

        class Stop(Exception):
            pass

        def tape_randn_like(x):
            if not queue:
                raise Stop()
            return queue.pop(0).clone()

        def wrap(store, f0):
            def outer(*a, **k):
                f = f0(*a, **k)

                def g(x, t, *aa, **kk):
                    r = f(x, t, *aa, **kk)
                    calls[store].append(r.detach().clone())
                    return r
                return g
            return outer

        def rec_c(self, x, t, *a, **k):
            r = orig[3](self, x, t, *a, **k)
            calls["corr"].append((r[0].clone(), r[1].clone()))
            return r

        n_pred = [0]

        def rec_p(self, x, t, *a, **k):
            if predictor == "none" and n_pred[0] >= n_run:
                raise Stop()                 # no randn_like call would ever stop a NonePredictor + NoneCorrector loop
            r = orig[4](self, x, t, *a, **k)
            n_pred[0] += 1
            calls["pred"].append((r[0].clone(), r[1].clone()))
            return r

        torch.randn_like = tape_randn_like
        mutils.get_score_fn, mutils.get_classifier_grad_fn = wrap("score", orig[1]), wrap("cgrad", orig[2])
        Cc.update_fn, P.update_fn = rec_c, rec_p
        sde.prior_sampling = lambda shape: x0.clone()
        final = None
        try:
            fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), datasets_nas.get_data_inverse_scaler(cfg), eps,
                                          conditional=guided, data_name="cifar10", num_sample=20)
            try:
                if guided:
                    final = fn(score_model, mask, clf, classifier_scale=classifier_scale, task=task)
                else:
                    final = fn(score_model, mask.unsqueeze(0))[0]
            except Stop:
                pass
        finally:
            torch.randn_like, mutils.get_score_fn, mutils.get_classifier_grad_fn = orig[0], orig[1], orig[2]
            Cc.update_fn, P.update_fn = orig[3], orig[4]
            clf.sample_state = False
        n_done = len(calls["pred"])
        per = (corrector != "none") + (predictor != "none")       # score evaluations per step
        arrs = dict(seed=0, preset=preset, tape_seed=tape_seed, sde_N=sde_N, sde_kind=sde_kind, predictor=predictor,
                    corrector=corrector, n_run=n_done, B=B, guided=int(guided), classifier_scale=classifier_scale,
                    mask=mask, snr=0.16, eps=eps, x0=x0)
        arrs["x_pred"] = torch.stack([p[0] for p in calls["pred"]])
        arrs["x_mean"] = torch.stack([p[1] for p in calls["pred"]])
        if corrector != "none":
            arrs["x_corr"] = torch.stack([c[0] for c in calls["corr"][:n_done]])
            arrs["score_c"] = torch.stack(calls["score"][0:per * n_done:per])
            if guided:
                arrs["cgrad_c"] = torch.stack(calls["cgrad"][0:per * n_done:per])
        if predictor != "none":
            arrs["score_p"] = torch.stack(calls["score"][per - 1:per * n_done:per])
            if guided:
                arrs["cgrad_p"] = torch.stack(calls["cgrad"][per - 1:per * n_done:per])
        if final is not None:
            arrs["final"] = final
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
        print(f"{tag}.npz  {os.path.getsize(path)/1024:.1f} KiB  steps={n_done} final={'yes' if final is not None else 'no'}")

    # the reference's shipped NB201 default: VESDE + euler_maruyama + langevin, guided, real schedule, first steps
    run("pc_guided_nb201_ve_em_n1000_first8", sde_kind="ve", predictor="euler_maruyama", corrector="langevin", sde_N=1000,
        n_run=8, B=4, guided=True, mask=closure, preset="default", tape_seed=31, eps=1e-5, classifier_scale=10000.0)
    # complete trajectories (contractive preset): final tensors + quantised architectures
    run("pc_uncond_nb201_ve_rd_contractive_n60", sde_kind="ve", predictor="reverse_diffusion", corrector="langevin", sde_N=60,
        n_run=60, B=4, guided=False, mask=adj, preset="contractive", tape_seed=32, eps=1e-5)
    run("pc_guided_nb201_ve_em_contractive_n60", sde_kind="ve", predictor="euler_maruyama", corrector="langevin", sde_N=60,
        n_run=60, B=4, guided=True, mask=closure, preset="contractive", tape_seed=33, eps=1e-5)
    # the OFA tree's shipped default pair: euler_maruyama + none corrector (here on the VPSDE)
    run("pc_uncond_nb201_vp_em_nonecorr_contractive_n60", sde_kind="vp", predictor="euler_maruyama", corrector="none", sde_N=60,
        n_run=60, B=4, guided=False, mask=adj, preset="contractive", tape_seed=34, eps=1e-3)
    # corrector-only sampler
    run("pc_uncond_nb201_vp_nonepred_n1000_first6", sde_kind="vp", predictor="none", corrector="langevin", sde_N=1000,
        n_run=6, B=4, guided=False, mask=adj, preset="default", tape_seed=35, eps=1e-3)


if __name__ == "__main__":
    main()
