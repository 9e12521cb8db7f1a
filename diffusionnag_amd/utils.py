"""Host mirrors of the few helpers of the reference's ``utils.py`` / ``datasets_nas.py`` that sit
at the two ends of the sampling path (the reference file itself needs ``torch_scatter``)."""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def get_data_scaler(config):
    """Reference ``datasets_nas.py:11-18``."""
    if config.data.centered:
        return lambda x: x * 2. - 1.
    return lambda x: x


def get_data_inverse_scaler(config):
    """Reference ``datasets_nas.py:21-28``.  The returned callable is tagged so that the sampler can
    let the exit kernel apply ``(x + 1) / 2`` itself."""
    if config.data.centered:
        def inverse_scaler(x):
            return (x + 1.) / 2.
        inverse_scaler.dnag_kind = "centered"
    else:
        def inverse_scaler(x):
            return x
        inverse_scaler.dnag_kind = "identity"
    return inverse_scaler


def nb201_adj() -> torch.Tensor:
    """The fixed 8-node NAS-Bench-201 cell DAG (values of reference ``datasets_nas.py:182-194``)."""
    a = torch.zeros(8, 8)
    for i, j in [(0, 1), (0, 2), (0, 3), (1, 4), (1, 5), (2, 6), (3, 7), (4, 6), (5, 7), (6, 7)]:
        a[i, j] = 1.0
    return a


def ofa_adj(n: int = 20) -> torch.Tensor:
    """OFA chain DAG (reference ``MobileNetV3/main_exp/diffusion/run_lib.py:147-153``)."""
    a = torch.zeros(n, n)
    for i in range(n - 1):
        a[i, i + 1] = 1.0
    return a


def floyed(r):
    """Reachability closure, reference ``utils.py:123-139``."""
    r = np.array(r, dtype=np.float32, copy=True)
    n = r.shape[0]
    for k in range(n):
        for i in range(n):
            for j in range(n):
                if r[i, k] > 0 and r[k, j] > 0:
                    r[i, j] = 1
    return r


def aug_mask(adj, algo='floyed', data='NASBench201'):
    """Reference ``utils.py:142-156`` for the ``nasbench201`` / ``ofa`` branch: the mask of ``adj[0]``
    is replicated over the batch; only the literal ``'floyed'`` closes it, any other string
    (including the configs' ``'floyd'``) keeps the raw adjacency."""
    adj = torch.as_tensor(adj, dtype=torch.float32)
    if len(adj.shape) == 2:
        adj = adj.unsqueeze(0)
    if data.lower() not in ['nasbench201', 'ofa']:
        raise NotImplementedError(f"aug_mask for data={data!r} is not on the sampling path")
    assert len(adj.shape) == 3
    r = adj[0].clone().detach()
    if algo == 'floyed':
        mask_i = torch.from_numpy(floyed(r.cpu().numpy())).float().to(adj.device)
    elif algo == 'long_range':
        raise NotImplementedError("algo='long_range' is not used by any sampling caller")
    else:
        mask_i = r
    return torch.stack([mask_i] * adj.size(0))


class MetaTestDataset:
    """Reference ``datasets_nas.py:437-468``: a task = ``num_sample`` random image features per class.
    The feature file is one of the reference's data blobs (``{data_name}bylabel.pt``), written by the
    reference's own preprocessing; it is only opened when a sampler is called without ``task=``."""

    def __init__(self, data_path, data_name, num_sample, num_class=None):
        import os
        self.num_sample = num_sample
        self.data_name = data_name
        num_class_dict = {'cifar100': 100, 'cifar10': 10, 'aircraft': 30, 'pets': 37}
        self.num_class = num_class if num_class is not None else num_class_dict[data_name]
        self.path = os.path.join(data_path, f'{data_name}bylabel.pt')
        self.x = None

    def __len__(self):
        return 1000000

    def __getitem__(self, index):
        if self.x is None:
            self.x = torch.load(self.path, weights_only=True)
        data = []
        for cls in range(self.num_class):
            cx = self.x[cls][0]
            ridx = torch.randperm(len(cx))
            data.append(cx[ridx[:self.num_sample]])
        return torch.cat(data)


def quantize(x):
    """Reference ``utils.py:244-262``: threshold at 0.5 and hand back a list of ``[N, V]`` numpy arrays.
    The threshold runs on the device (``dnag_quantize``); one device-to-host copy replaces the
    reference's per-sample ``.cpu()``."""
    lib = _lib.load()
    dev = _lib.require_device(x)
    xc = x.detach().to(torch.float32).contiguous()
    q = torch.empty_like(xc)
    with torch.cuda.device(dev):
        _lib.check(lib.dnag_quantize(xc.data_ptr(), q.data_ptr(), xc.numel(), _lib.current_stream(dev)), "dnag_quantize")
    host = q.cpu().numpy()
    return [host[i] for i in range(host.shape[0])]


# ---------------------------------------------------------------------------------------------------
# Checkpoints (reference ``utils.py:84-121``; layout in SURVEY.md Appendix C)
# ---------------------------------------------------------------------------------------------------
def _safe_load(path, device):
    """``torch.load`` that executes nothing from the file.  The reference stores its ``ml_collections.ConfigDict``
    in the checkpoint; ``weights_only=True`` refuses such a class unless the caller has allow-listed it with
    ``torch.serialization.add_safe_globals`` -- the error says so instead of falling back to an unsafe load."""
    try:
        return torch.load(path, map_location=device, weights_only=True)
    except Exception as e:   # noqa: BLE001 - re-raised with advice
        raise RuntimeError(
            f"{path}: refused by the safe loader ({type(e).__name__}: {e}).  Checkpoints written by the reference hold a "
            "ConfigDict under 'config': allow-list its classes with torch.serialization.add_safe_globals([...]) or re-save "
            "the checkpoint with only 'model', 'ema', 'optimizer' and 'step'.") from e


def restore_checkpoint(ckpt_dir, state, device, resume=False):
    """Reference ``utils.py:84-101``: same arguments and behaviour, safe loader."""
    import logging
    import os
    if not resume:
        os.makedirs(os.path.dirname(ckpt_dir), exist_ok=True)
        return state
    if not os.path.exists(ckpt_dir):
        os.makedirs(os.path.dirname(ckpt_dir), exist_ok=True)
        logging.warning(f"No checkpoint found at {ckpt_dir}. Returned the same state as input")
        return state
    loaded_state = _safe_load(ckpt_dir, device)
    for k in state:
        if k in ['optimizer', 'model', 'ema']:
            state[k].load_state_dict(loaded_state[k])
        else:
            state[k] = loaded_state[k]
    return state


def save_checkpoint(ckpt_dir, state, step, save_step, is_best, remove_except_best=False):
    """Reference ``utils.py:104-121``: ``checkpoint_{step}_{save_step}.pth.tar`` (+ ``model_best.pth.tar``)."""
    import os
    import shutil
    saved_state = {k: (state[k].state_dict() if k in ['optimizer', 'model', 'ema'] else state[k]) for k in state}
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f'checkpoint_{step}_{save_step}.pth.tar')
    torch.save(saved_state, path)
    if is_best:
        shutil.copy(path, os.path.join(ckpt_dir, 'model_best.pth.tar'))
    if remove_except_best:
        for f in sorted(os.listdir(ckpt_dir)):
            if f.startswith('checkpoint') and os.path.join(ckpt_dir, f) != os.path.join(ckpt_dir, 'model_best.pth.tar'):
                os.remove(os.path.join(ckpt_dir, f))


def load_score_model(ckpt_path, config, device=None):
    """Reference ``run_lib.get_score_model`` (``main_exp/diffusion/run_lib.py:61-79``): create the model, load
    ``'model'``, then copy the EMA shadow parameters into it (``ema.copy_to``), which is what every sampling caller
    does before using the network."""
    from . import mutils
    from .ema import ExponentialMovingAverage
    model = mutils.create_model(config)
    loaded = _safe_load(ckpt_path, device or config.device)
    model.load_state_dict(loaded['model'], strict=False)
    ema = ExponentialMovingAverage(model.parameters(), decay=loaded['ema']['decay'] if 'ema' in loaded else 0.9999)
    if 'ema' in loaded:
        ema.load_state_dict(loaded['ema'])
        ema.shadow_params = [p.to(next(model.parameters()).device) for p in ema.shadow_params]
        ema.copy_to(model.parameters())
    return model, ema
