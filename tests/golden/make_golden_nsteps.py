"""Capture golden vectors of the sampler with ``n_steps_each = 2`` (two Langevin corrector updates per PC step,
reference ``sampling.py:244``) from the REFERENCE implementation (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nsteps.py

Same method as make_golden_ve.py: imports ``/root/reference/NAS-Bench-201`` unmodified, loads the seeded synthetic
weights into the reference's own modules, drives the reference's ``get_sampling_fn`` closure with an injected noise
tape (``torch.randn_like`` replaced in this process only; per step: one draw per corrector update, then the
predictor's) and records the score / gradient functions' outputs and the corrector's / predictor's results.  Only
inputs, seeds and outputs are stored.
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

CS = 2


def main():
    scratch = tempfile.mkdtemp(prefix="dnag_golden_")
    os.makedirs(os.path.join(scratch, "data", "transfer_nag"))
    torch.save([[torch.randn(30, 512)] for _ in range(10)], os.path.join(scratch, "data", "transfer_nag", "cifar10bylabel.pt"))
    os.chdir(scratch)
    sys.path.insert(0, REF)

    import sde_lib                              # reference
    import sampling                             # reference
    import models.utils as mutils               # reference
    from models import cate, digcn_meta         # noqa: F401
    import datasets_nas                         # reference

    nb, pd = CateDims(), PredictorDims()
    closure = synthetic.transitive_closure(synthetic.nb201_adjacency())
    clf = mutils.create_model(predictor_config(pd))
    clf.load_state_dict(synthetic.predictor_state_dict(pd, seed=1), strict=True)
    clf.eval()
    task = synthetic.synthetic_task(10, 20, seed=2)

    def run(tag, *, sde_N, n_run, B, preset, tape_seed, classifier_scale):
        sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20.0, N=sde_N)
        cfg = score_config(nb)
        cfg.sampling["n_steps_each"] = CS
        score_model = mutils.create_model(cfg)
        score_model.load_state_dict(synthetic.cate_state_dict(nb, seed=0, preset=preset), strict=True)
        tape = synthetic.noise_tape(n_run * CS, (B, 8, 7), tape_seed)       # corrector: n_run * CS draws; predictor: the first n_run
        queue = []
        for i in range(n_run):
            queue += [tape["corrector"][i * CS + k] for k in range(CS)] + [tape["predictor"][i]]
        calls = {"score": [], "cgrad": [], "corr": [], "pred": []}
        Cc, P = sampling.LangevinCorrector, sampling.ReverseDiffusionPredictor
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

        def rec_p(self, x, t, *a, **k):
            r = orig[4](self, x, t, *a, **k)
            calls["pred"].append((r[0].clone(), r[1].clone()))
            return r

        torch.randn_like = tape_randn_like
        mutils.get_score_fn, mutils.get_classifier_grad_fn = wrap("score", orig[1]), wrap("cgrad", orig[2])
        Cc.update_fn, P.update_fn = rec_c, rec_p
        sde.prior_sampling = lambda shape: tape["x0"].clone()
        final = None
        try:
            fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), datasets_nas.get_data_inverse_scaler(cfg), 1e-3,
                                          conditional=True, data_name="cifar10", num_sample=20)
            try:
                final = fn(score_model, closure, clf, classifier_scale=classifier_scale, task=task)
            except Stop:
                pass
        finally:
            torch.randn_like, mutils.get_score_fn, mutils.get_classifier_grad_fn = orig[0], orig[1], orig[2]
            Cc.update_fn, P.update_fn = orig[3], orig[4]
            clf.sample_state = False
        n_done = len(calls["pred"])
        per = CS + 1                                                      # score evaluations per PC step
        sc, cg = calls["score"][:per * n_done], calls["cgrad"][:per * n_done]
        arrs = dict(seed=0, preset=preset, tape_seed=tape_seed, sde_N=sde_N, n_run=n_done, B=B, n_steps_each=CS,
                    classifier_scale=classifier_scale, mask=closure, snr=0.16, eps=1e-3)
        # corrector evaluations in order (entry i * CS + k), predictor evaluations per step
        arrs["score_c"] = torch.stack([sc[i * per + k] for i in range(n_done) for k in range(CS)])
        arrs["cgrad_c"] = torch.stack([cg[i * per + k] for i in range(n_done) for k in range(CS)])
        arrs["score_p"] = torch.stack([sc[i * per + CS] for i in range(n_done)])
        arrs["cgrad_p"] = torch.stack([cg[i * per + CS] for i in range(n_done)])
        arrs["x_corr_last"] = torch.stack([c[0] for c in calls["corr"][:n_done]])      # x after the CS-th update of each step
        arrs["x_pred"] = torch.stack([p[0] for p in calls["pred"]])
        arrs["x_mean"] = torch.stack([p[1] for p in calls["pred"]])
        if final is not None:
            arrs["final"] = final
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
        print(f"{tag}.npz  {os.path.getsize(path)/1024:.1f} KiB  steps={n_done} final={'yes' if final is not None else 'no'}")

    run("pc_guided_nb201_nsteps2_n1000_first6", sde_N=1000, n_run=6, B=4, preset="default", tape_seed=51, classifier_scale=10000.0)
    run("pc_guided_nb201_nsteps2_contractive_n40", sde_N=40, n_run=40, B=4, preset="contractive", tape_seed=52, classifier_scale=100.0)


if __name__ == "__main__":
    main()
