"""Capture golden vectors of the OPTIMISER side of the training step from the REFERENCE implementation (build
container only): ``losses.get_optimizer`` + ``losses.optimization_manager(config)`` (warm-up, clip_grad_norm_, Adam,
reference ``losses.py:10-36``) and ``models.ema.ExponentialMovingAverage.update`` (``models/ema.py:24-42``) driven for
three consecutive steps, as ``step_fn`` does after ``loss.backward()`` (``losses.py:222-226``).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_optim.py

The parameters are a seeded list of small tensors (the optimiser does not care what network they belong to) and the
gradients are seeded too; only inputs, seeds and outputs are stored.  torch 2.10 CPU (the reference pins 1.13.1).
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

from make_golden import Cfg  # noqa: E402

SHAPES = [(64, 7), (64,), (64, 32), (256,), (3, 16, 16), (32, 16), (1,), (7, 96), (37,)]


def optim_inputs(seed, n_steps, grad_scales):
    """Seeded parameters and per-step gradients, flat in SHAPES order."""
    rs = np.random.RandomState(seed)
    n = int(sum(np.prod(s) for s in SHAPES))
    params = (rs.standard_normal(n) * 0.05).astype(np.float32)
    grads = np.stack([(rs.standard_normal(n) * sc).astype(np.float32) for sc in grad_scales[:n_steps]])
    return params, grads


CASES = {
    # the shipped training config (configs/tr_scorenet.py: lr 2e-5, beta1 0.9, eps 1e-8, warmup 1000, grad_clip 1, ema 0.9999);
    # step 400: inside the warm-up; the first gradient is clipped, the other two are not
    "optim_adam_ema_clip": dict(seed=3, lr=2e-5, beta1=0.9, eps=1e-8, weight_decay=0.0, warmup=1000, grad_clip=1.0,
                                ema_decay=0.9999, start_step=400, grad_scales=[0.05, 1e-3, 3e-3]),
    # weight decay, no warm-up, no clipping, a short EMA horizon
    "optim_adam_ema_wd": dict(seed=4, lr=1e-3, beta1=0.8, eps=1e-6, weight_decay=0.01, warmup=0, grad_clip=-1.0,
                              ema_decay=0.95, start_step=0, grad_scales=[1.0, 0.3, 2.0]),
}


def main():
    scratch = tempfile.mkdtemp(prefix="dnag_golden_")
    os.chdir(scratch)
    sys.path.insert(0, REF)
    import losses                               # reference
    from models.ema import ExponentialMovingAverage   # reference

    for name, c in CASES.items():
        p0, grads = optim_inputs(c["seed"], 3, c["grad_scales"])
        params, o = [], 0
        for shp in SHAPES:
            k = int(np.prod(shp))
            params.append(torch.nn.Parameter(torch.from_numpy(p0[o:o + k].copy()).reshape(shp)))
            o += k
        cfg = Cfg(optim=Cfg(optimizer="Adam", lr=c["lr"], beta1=c["beta1"], eps=c["eps"], weight_decay=c["weight_decay"],
                            warmup=c["warmup"], grad_clip=c["grad_clip"]))
        optimizer = losses.get_optimizer(cfg, params)
        optimize_fn = losses.optimization_manager(cfg)
        ema = ExponentialMovingAverage(params, decay=c["ema_decay"])
        step = c["start_step"]
        rec_p, rec_s, norms = [], [], []
        for k in range(3):
            o = 0
            for p in params:
                n = p.numel()
                p.grad = torch.from_numpy(grads[k, o:o + n].copy()).reshape(p.shape)
                o += n
            norms.append(float(torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in params]))))
            optimize_fn(optimizer, params, step=step)          # losses.py:225
            step += 1
            ema.update(params)                                 # losses.py:227
            rec_p.append(torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy())
            rec_s.append(torch.cat([s.reshape(-1) for s in ema.shadow_params]).numpy().copy())
        st = optimizer.state_dict()["state"]
        m = torch.cat([st[i]["exp_avg"].reshape(-1) for i in range(len(params))]).numpy()
        v = torch.cat([st[i]["exp_avg_sq"].reshape(-1) for i in range(len(params))]).numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, params=np.stack(rec_p), shadow=np.stack(rec_s), exp_avg=m, exp_avg_sq=v,
                            grad_norms=np.array(norms, np.float64), ema_num_updates=ema.num_updates,
                            **{k_: np.asarray(v_) for k_, v_ in c.items()})
        print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
