"""Capture golden vectors from the REFERENCE implementation (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py            # every fixture
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py margin     # only the fixtures whose name contains "margin"

Imports ``/root/reference/NAS-Bench-201`` (read-only, unmodified), loads the seeded
synthetic weights of ``diffusionnag_amd.synthetic`` into the reference's own modules
(``load_state_dict(strict=True)`` -- which also proves our parameter layout equals the
reference's), drives the reference's own ``get_sampling_fn`` closures with an injected
noise tape, and writes small ``.npz`` fixtures next to this script.  Only inputs,
seeds and outputs are stored; no reference source travels.

The reference never exposes intermediates, so they are captured by wrapping -- in THIS
process only -- ``torch.randn_like`` (noise tape), ``sde.prior_sampling`` (x0),
``models.utils.get_score_fn`` / ``get_classifier_grad_fn`` (per-call outputs) and the
``update_fn`` of the registered predictor / corrector classes (states).
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
sys.dont_write_bytecode = True

from diffusionnag_amd import synthetic  # noqa: E402
from diffusionnag_amd.specs import CateDims, PredictorDims  # noqa: E402


class Cfg(dict):
    """Minimal stand-in for ml_collections.ConfigDict (attribute access + ``in``)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def score_config(dims: CateDims):
    ge = Cfg(n_layers=dims.n_layers, d_model=dims.d_model, n_head=dims.n_head, d_ff=dims.d_ff,
             dropout=0.1, n_vocab=dims.n_vocab)
    return Cfg(device=torch.device("cpu"),
               data=Cfg(name="ofa" if dims.stagewise_pos else "NASBench201", max_node=dims.max_node,
                        n_vocab=dims.n_vocab, centered=True),
               model=Cfg(name="CATE", nonlinearity="swish", dropout=0.1, pos_enc_type=2, graph_encoder=ge),
               training=Cfg(continuous=True),
               sampling=Cfg(method="pc", predictor="reverse_diffusion", corrector="langevin", snr=0.16,
                            n_steps_each=1, probability_flow=False, noise_removal=True, regress=True,
                            labels="max"))


def predictor_config(d: PredictorDims):
    ge = Cfg(initial_hidden=d.n_vocab, gcn_hidden=d.gcn_hidden, gcn_layers=d.gcn_layers,
             linear_hidden=d.linear_hidden)
    return Cfg(device=torch.device("cpu"),
               model=Cfg(name="MetaNeuralPredictor", nonlinearity="swish", input_type="DA", hs=d.hs, nz=d.nz,
                         num_sample=d.num_sample, graph_encoder=ge))


def main():
    scratch = tempfile.mkdtemp(prefix="dnag_golden_")
    os.makedirs(os.path.join(scratch, "data", "transfer_nag"))
    # MetaTestDataset loads this at sampler-construction time (sampling.py:490-493); synthetic stand-in.
    torch.save([[torch.randn(30, 512)] for _ in range(10)],
               os.path.join(scratch, "data", "transfer_nag", "cifar10bylabel.pt"))
    os.chdir(scratch)
    sys.path.insert(0, REF)

    import sde_lib                              # reference
    import sampling                             # reference
    import models.utils as mutils               # reference
    from models import cate, digcn_meta         # registers the models
    import datasets_nas                         # reference

    out = {}

    only = sys.argv[1] if len(sys.argv) > 1 else None

    def save(name, **arrs):
        if only is not None and only not in name:
            return
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                     for k, v in arrs.items()})
        out[name] = os.path.getsize(path)

    nb = CateDims()
    ofa = CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
    pd = PredictorDims()
    adj = synthetic.nb201_adjacency()
    assert torch.equal(adj, datasets_nas.NASBench201Dataset.get_adj(None))
    closure = synthetic.transitive_closure(adj)

    # ------------------------------------------------------------------ A: CATE forward, NB201
    sd = synthetic.cate_state_dict(nb, seed=0, perturb_norms=True)
    model = mutils.create_model(score_config(nb))
    model.load_state_dict(sd, strict=True)
    model.eval()
    rs = np.random.RandomState(100)
    X = torch.from_numpy(rs.standard_normal((6, 8, 7)).astype(np.float32))
    tc = torch.tensor([1.0, 0.5, 1e-3, 0.25, 0.75, 0.999]) * 999
    rec = {}
    hooks = [model.graph_encoder.encoder_f.layers[0].self_attn.attn.register_forward_hook(
        lambda m, i, o: rec.__setitem__("attn0", o[1].detach())),
        model.graph_encoder.register_forward_hook(lambda m, i, o: rec.__setitem__("concat", o.detach()))]
    with torch.no_grad():
        for mname, m in (("raw", adj), ("closure", closure)):
            y = model(X, tc, m.unsqueeze(0).repeat(6, 1, 1))
            save(f"cate_fwd_nb201_{mname}", seed=0, X=X, time_cond=tc, mask=m, out=y,
                 attn0=rec["attn0"], concat=rec["concat"])
    for h in hooks:
        h.remove()

    # ------------------------------------------------------------------ B: CATE forward, OFA shape
    sd_ofa = synthetic.cate_state_dict(ofa, seed=3, perturb_norms=True)
    model_ofa = mutils.create_model(score_config(ofa))
    model_ofa.load_state_dict(sd_ofa, strict=True)
    model_ofa.eval()
    Xo = torch.from_numpy(rs.standard_normal((3, 20, 9)).astype(np.float32))
    tco = torch.tensor([1.0, 0.4, 1e-3]) * 999
    chain = synthetic.ofa_adjacency(20)
    with torch.no_grad():
        yo = model_ofa(Xo, tco, chain.unsqueeze(0).repeat(3, 1, 1))
    save("cate_fwd_ofa_chain", seed=3, X=Xo, time_cond=tco, mask=chain, out=yo)

    # ------------------------------------------------------------------ C: surrogate forward + guidance gradient
    sdp = synthetic.predictor_state_dict(pd, seed=1)
    clf = mutils.create_model(predictor_config(pd))
    clf.load_state_dict(sdp, strict=True)
    clf.eval()
    task = synthetic.synthetic_task(10, 20, seed=2)
    sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    Xp = torch.from_numpy(rs.standard_normal((5, 8, 7)).astype(np.float32))
    for mname, m in (("raw", adj), ("closure", closure)):
        mB = m.unsqueeze(0).repeat(5, 1, 1)
        clf.sample_state, clf.D_mu = True, None
        ys, gs, ts_ = [], [], []
        for tval in (1.0, 0.37, 1e-3):
            t = torch.ones(5) * tval
            with torch.no_grad():
                y = clf(Xp, t * 999, mB, task=task.repeat(5, 1, 1))
            gfn = mutils.get_classifier_grad_fn(sde, clf, train=False, continuous=True, regress=True, labels="max")
            g = gfn(Xp, t, mB, task=task.repeat(5, 1, 1))
            ys.append(y), gs.append(g), ts_.append(tval)
        save(f"predictor_nb201_{mname}", seed=1, task_seed=2, X=Xp, t=np.array(ts_, np.float32), mask=m,
             D_mu=clf.D_mu[0].detach(), y=torch.stack(ys), grad=torch.stack(gs))
        clf.sample_state = False

    # ------------------------------------------------------------------ D/E: sampler trajectories
    def run_sampler(tag, *, sde_N, n_run, B, guided, mask, preset, tape_seed, beta_max=20.0,
                    classifier_scale=10000.0, full_record=True, double=False, extra=None):
        sde = sde_lib.VPSDE(beta_min=0.1, beta_max=beta_max, N=sde_N)
        cfg = score_config(nb)
        sdm = synthetic.cate_state_dict(nb, seed=0, preset=preset)
        score_model = mutils.create_model(cfg)
        score_model.load_state_dict(sdm, strict=True)
        tape = synthetic.noise_tape(n_run, (B, 8, 7), tape_seed)
        if double:
            # The same run in fp64: the reference's own modules cast to double, the same fp32 inputs (weights, tape,
            # schedule tables, time embedding -- get_timestep_embedding builds it in fp32 by construction,
            # cate.py:363-374, so only its result is cast).  Tie-breaker for trajectories whose fp32 rounding noise is
            # amplified by the dynamics ("Hard parts" (iv)).
            score_model.double()
            clf.double()
            tape = {k: v.double() for k, v in tape.items()}
            orig_te = (cate.get_timestep_embedding, digcn_meta.get_timestep_embedding)
            cate.get_timestep_embedding = lambda *a, **k: orig_te[0](*a, **k).double()
            digcn_meta.get_timestep_embedding = lambda *a, **k: orig_te[1](*a, **k).double()
        queue = []
        for i in range(n_run):
            queue += [tape["corrector"][i], tape["predictor"][i]]
        calls = {"score": [], "cgrad": [], "corr": [], "pred": []}

        orig_randn_like, orig_sfn, orig_cfn = torch.randn_like, mutils.get_score_fn, mutils.get_classifier_grad_fn
        orig_c = sampling.LangevinCorrector.update_fn
        orig_p = sampling.ReverseDiffusionPredictor.update_fn
        real_N = sde.N

        def tape_randn_like(x):
            return queue.pop(0).clone()

        def rec_sfn(*a, **k):
            f = orig_sfn(*a, **k)

            def g(x, t, *aa, **kk):
                r = f(x, t, *aa, **kk)
                calls["score"].append(r.detach().clone())
                return r
            return g

        def rec_cfn(*a, **k):
            f = orig_cfn(*a, **k)

            def g(x, t, *aa, **kk):
                r = f(x, t, *aa, **kk)
                calls["cgrad"].append(r.detach().clone())
                return r
            return g

        def rec_c(self, x, t, *a, **k):
            r = orig_c(self, x, t, *a, **k)
            calls["corr"].append((x.clone(), r[0].clone(), r[1].clone()))
            return r

        def rec_p(self, x, t, *a, **k):
            r = orig_p(self, x, t, *a, **k)
            calls["pred"].append((r[0].clone(), r[1].clone()))
            return r

        class Stop(Exception):
            pass

        # Truncate the loop after n_run steps without touching the time grid: the tape runs dry.
        def tape_or_stop(x):
            if not queue:
                raise Stop()
            return tape_randn_like(x)

        torch.randn_like = tape_or_stop
        mutils.get_score_fn, mutils.get_classifier_grad_fn = rec_sfn, rec_cfn
        sampling.LangevinCorrector.update_fn = rec_c
        sampling.ReverseDiffusionPredictor.update_fn = rec_p
        sde.prior_sampling = lambda shape: tape["x0"].clone()
        final = None
        try:
            fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), datasets_nas.get_data_inverse_scaler(cfg), 1e-3,
                                          conditional=guided, data_name="cifar10", num_sample=20)
            try:
                mk, tk = (mask.double(), task.double()) if double else (mask, task)
                if guided:
                    final = fn(score_model, mk, clf, classifier_scale=classifier_scale, task=tk)
                else:
                    final = fn(score_model, mk.unsqueeze(0))[0]
            except Stop:
                pass
        finally:
            torch.randn_like = orig_randn_like
            mutils.get_score_fn, mutils.get_classifier_grad_fn = orig_sfn, orig_cfn
            sampling.LangevinCorrector.update_fn = orig_c
            sampling.ReverseDiffusionPredictor.update_fn = orig_p
            clf.sample_state = False
            if double:
                clf.float()
                cate.get_timestep_embedding, digcn_meta.get_timestep_embedding = orig_te
        assert real_N == sde_N
        if double:
            assert final.dtype == torch.float64
            return final
        n_done = len(calls["pred"])
        arrs = dict(seed=0, preset=preset, tape_seed=tape_seed, sde_N=sde_N, beta_max=beta_max, n_run=n_done, B=B,
                    guided=int(guided), classifier_scale=classifier_scale, mask=mask, snr=0.16, eps=1e-3)
        if full_record:
            arrs["x_in"] = torch.stack([c[0] for c in calls["corr"][:n_done]])
            arrs["x_corr"] = torch.stack([c[1] for c in calls["corr"][:n_done]])
            arrs["x_pred"] = torch.stack([p[0] for p in calls["pred"]])
            arrs["x_mean"] = torch.stack([p[1] for p in calls["pred"]])
            arrs["score_c"] = torch.stack(calls["score"][0:2 * n_done:2])
            arrs["score_p"] = torch.stack(calls["score"][1:2 * n_done:2])
            if guided:
                arrs["cgrad_c"] = torch.stack(calls["cgrad"][0:2 * n_done:2])
                arrs["cgrad_p"] = torch.stack(calls["cgrad"][1:2 * n_done:2])
        else:
            arrs["x_mean_last"] = calls["pred"][-1][1]
        if final is not None:
            arrs["final"] = final
        if extra:
            arrs.update(extra)
        save(tag, **arrs)

    # random-init weights, real schedule (N=1000), first steps: per-step teacher-forced parity
    run_sampler("pc_uncond_nb201_n1000_first8", sde_N=1000, n_run=8, B=8, guided=False, mask=adj,
                preset="default", tape_seed=11)
    run_sampler("pc_guided_nb201_n1000_first8", sde_N=1000, n_run=8, B=8, guided=True, mask=closure,
                preset="default", tape_seed=12)
    # contractive preset: complete trajectories, final tensors + quantised architectures
    run_sampler("pc_uncond_nb201_contractive_n100", sde_N=100, n_run=100, B=8, guided=False, mask=adj,
                preset="contractive", tape_seed=13)
    run_sampler("pc_guided_nb201_contractive_n100", sde_N=100, n_run=100, B=8, guided=True, mask=closure,
                preset="contractive", tape_seed=14, classifier_scale=100.0)
    run_sampler("pc_guided_nb201_contractive_n1000", sde_N=1000, n_run=1000, B=4, guided=True, mask=closure,
                preset="contractive", tape_seed=15, classifier_scale=100.0, full_record=False, beta_max=5.0)

    # decision margins: B=64 complete guided trajectories on three tapes; the test compares the discretised
    # architectures bit for bit and reports how far the final states sit from the 0.5 threshold
    # Each run is captured twice, in the reference's fp32 and with its modules cast to fp64 (`final64`): these
    # trajectories amplify rounding noise (a 1e-6 relative change of x0 moves the end state by 1e-5 .. 1e-2), so the
    # fp32 reference is itself only one sample of that noise; the fp64 run says which entries fp32 can decide at all.
    for k, tape_seed in enumerate((31, 32, 33)):
        kw = dict(sde_N=100, n_run=100, B=64, guided=True, mask=closure, preset="contractive", tape_seed=tape_seed,
                  classifier_scale=100.0, full_record=False)
        f64 = run_sampler(f"pc_guided_nb201_contractive_margin{k}", double=True, **kw)
        run_sampler(f"pc_guided_nb201_contractive_margin{k}", extra={"final64": f64}, **kw)

    for k, v in out.items():
        print(f"{k}.npz  {v/1024:.1f} KiB")


if __name__ == "__main__":
    main()
