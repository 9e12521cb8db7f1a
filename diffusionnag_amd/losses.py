"""Host mirror of the reference's ``losses.py`` for the score network's DSM training step
(reference ``losses.py:10-36``, ``:110-168``, ``:171-237``; BASELINE config 3).

``get_sde_loss_fn_nas`` returns a ``loss_fn(model, batch)`` with the reference's signature.  The loss and
the gradient of every ``CATE`` parameter come from ONE call of ``dnag_cate_dsm_step`` in ``libdnag.so``
(forward with dropout, loss, hand-derived backward); the returned tensor is attached to the autograd graph
through a ``torch.autograd.Function`` that only hands those gradients over, so the reference's
``optimizer.zero_grad(); loss = loss_fn(...); loss.backward(); optimize_fn(...)`` sequence works unchanged.
PyTorch runs no part of the network.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, sde_lib
from .flat import chain_view, flatten_, FlatCache
from .cate import CATE, Workspace, positional_table, timestep_frequencies


class FusedAdam:
    """The ``torch.optim.Adam`` surface the reference's training code touches (``param_groups[...]['lr']``,
    ``zero_grad``, ``step``, ``state_dict`` / ``load_state_dict`` in torch's format), on flat device buffers: one call
    of ``dnag_optim_step`` (``include/dnag.h``) computes the global gradient norm, clips, applies Adam and -- when the
    EMA object is handed over -- updates the shadow parameters.  The parameters are re-seated once as views of one flat
    buffer (``flat.flatten_``), so ``model.parameters()`` and ``state_dict()`` keep working on the same memory.
    The Python-double scalars of ``torch.optim.adam._single_tensor_adam`` (bias corrections, step size) are computed
    here exactly as torch does and handed over rounded to fp32, as torch rounds them when they meet an fp32 tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.t = 0
        self._m = self._v = self._ws = None
        self.last_grad_norm = None

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    def _state(self, flat):
        if self._m is not None and self._m.numel() != flat.numel():
            raise RuntimeError("FusedAdam: the parameter list changed size after the first step")
        if self._m is None:
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        elif self._m.device != flat.device:      # model moved after load_state_dict: the moments move with it (self.t is kept)
            self._m, self._v = self._m.to(flat.device), self._v.to(flat.device)
        if self._ws is None or self._ws.device != flat.device:
            self._ws = torch.empty(_lib.load().dnag_optim_workspace_bytes() // 4, dtype=torch.float32, device=flat.device)
            self.last_grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.device)

    def step(self, grad_clip: float = -1.0, ema=None, flat_grads=None):
        """``flat_grads``: the gradients as one flat buffer in parameter order (the fused training step's own output),
        instead of the parameters' ``.grad``."""
        lib = _lib.load()
        if getattr(self, "_flat_cache", None) is None:
            self._flat_cache = FlatCache()
        flat = self._flat_cache.get(self.params, flatten_)
        dev = _lib.require_device(flat)
        if flat_grads is not None:
            grads = flat_grads
            if grads.numel() != flat.numel() or grads.device != flat.device or grads.dtype != torch.float32:
                raise RuntimeError("flat_grads does not match the parameters")
        else:
            missing = [p.grad is None for p in self.params]
            if all(missing):       # torch.optim.Adam skips parameters without a gradient: nothing to do
                return
            if any(missing):
                raise RuntimeError("FusedAdam updates one flat buffer: every parameter needs a gradient (torch.optim.Adam would "
                                   "skip the ones without; use it for partially frozen models)")
            grads = chain_view([p.grad for p in self.params])
            if grads is None:      # gradients that did not come from the fused training step: gather them (plumbing)
                grads = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in self.params])
        self._state(flat)
        g = self.param_groups[0]
        beta1, beta2 = g['betas']
        self.t += 1
        bias_correction1 = 1 - beta1 ** self.t                 # torch/optim/adam.py, python floats
        bias_correction2 = 1 - beta2 ** self.t
        step_size = float(g['lr']) / bias_correction1
        a = _lib.OptimArgsC()
        a.n, a.flags = flat.numel(), _lib.OPTIM_ADAM
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = _lib.ptr(flat), _lib.ptr(grads.contiguous()), _lib.ptr(self._m), _lib.ptr(self._v)
        a.beta1, a.beta2, a.one_minus_beta1, a.one_minus_beta2 = beta1, beta2, 1 - beta1, 1 - beta2
        a.eps, a.weight_decay = g['eps'], g['weight_decay']
        a.step_size, a.bc2_sqrt = step_size, bias_correction2 ** 0.5
        a.grad_clip = float(grad_clip)
        a.grad_norm_out = _lib.ptr(self.last_grad_norm)
        if ema is not None:
            a.flags |= _lib.OPTIM_EMA
            shadow = ema.flat_shadow(self.params)
            if shadow.numel() != flat.numel() or shadow.device != flat.device:
                raise RuntimeError(f"EMA shadow holds {shadow.numel()} values on {shadow.device}, the optimiser "
                                   f"{flat.numel()} on {flat.device}")
            a.shadow = _lib.ptr(shadow)
            a.ema_one_minus_decay = ema.next_one_minus_decay()
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel() * 4
        a.stream = _lib.current_stream(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.dnag_optim_step(C.byref(a)), "dnag_optim_step")
        # The kernel wrote through raw pointers: tell torch, as an in-place op would.  Caches keyed on the parameters'
        # versions (the kernel-layout weights of PackedModule.packed) follow from this.
        torch.autograd.graph.increment_version(self.params)
        if ema is not None:
            torch.autograd.graph.increment_version(ema.shadow_params)

    # torch.optim.Adam's checkpoint format (reference utils.py:104-121 saves optimizer.state_dict())
    def state_dict(self):
        state, o = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            if self._m is not None:
                state[i] = dict(step=torch.tensor(float(self.t)), exp_avg=self._m[o:o + n].view(p.shape).clone(),
                                exp_avg_sq=self._v[o:o + n].view(p.shape).clone())
            o += n
        grp = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        grp['params'] = list(range(len(self.params)))
        return dict(state=state, param_groups=[grp])

    def load_state_dict(self, sd):
        grp = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in grp:
                self.param_groups[0][k] = grp[k] if k != 'betas' else tuple(grp[k])
        st = sd['state']
        if st:
            flat = flatten_(self.params)
            self._m = self._v = None
            self._state(flat)
            o = 0
            for i, p in enumerate(self.params):
                n = p.numel()
                e = st.get(i, st.get(str(i)))
                if e is not None:
                    self._m[o:o + n].copy_(e['exp_avg'].reshape(-1))
                    self._v[o:o + n].copy_(e['exp_avg_sq'].reshape(-1))
                    self.t = int(float(e['step']))
                o += n


def get_optimizer(config, params):
    """Reference ``losses.py:10-19``; the Adam it returns runs on the device in one fused launch chain (``FusedAdam``)."""
    if config.optim.optimizer == 'Adam':
        return FusedAdam(params, lr=config.optim.lr, betas=(config.optim.beta1, 0.999), eps=config.optim.eps,
                         weight_decay=config.optim.weight_decay)
    raise NotImplementedError(f'Optimizer {config.optim.optimizer} not supported yet!')


def optimization_manager(config):
    """Reference ``losses.py:22-36``: warm-up, gradient clipping, optimiser step.  With a ``FusedAdam`` the clipping is
    part of the optimiser's launch chain; ``ema`` (an extension of the reference's signature, used by ``step_fn``)
    folds the EMA update of ``models/ema.py:24-42`` into the same pass."""

    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup, grad_clip=config.optim.grad_clip,
                    ema=None, flat_grads=None):
        if warmup > 0:
            for g in optimizer.param_groups:
                g['lr'] = lr * np.minimum(step / warmup, 1.0)
        if isinstance(optimizer, FusedAdam):
            optimizer.step(grad_clip=grad_clip, ema=ema, flat_grads=flat_grads)
            return
        if grad_clip >= 0:                      # a torch optimiser handed in by the caller: the reference's own lines
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        optimizer.step()
        if ema is not None:
            ema.update(params)

    return optimize_fn


def dropout_mask_layout(dims, batch: int):
    """Byte offsets of the dropout sites inside the ``keep`` buffer of ``dnag_cate_dsm_step``
    (``include/dnag.h``): ``{'rows', 'per_layer', 'emb', 'layer0', 'attn', 'sa', 'ffh', 'ffo', 'total'}``."""
    rows = batch * dims.max_node
    per_layer = dims.n_head * dims.max_node + 2 * dims.d_model + dims.d_ff
    return dict(rows=rows, per_layer=per_layer, emb=0, layer0=rows * dims.d_model, attn=0,
                sa=dims.n_head * dims.max_node, ffh=dims.n_head * dims.max_node + dims.d_model,
                ffo=dims.n_head * dims.max_node + dims.d_model + dims.d_ff,
                total=rows * dims.d_model + dims.n_layers * rows * per_layer)


def pack_dropout_masks(masks: Dict[str, torch.Tensor], dims, batch: int) -> torch.Tensor:
    """Keep masks named and shaped like the reference's dropout sites in forward order -- ``emb [B,N,d]``,
    per layer ``attn{l} [B,H,N,N]``, ``sa{l} [B,N,d]``, ``ffh{l} [B,N,d_ff]``, ``ffo{l} [B,N,d]``
    (``cate.py:147``, ``transformer.py:19``, ``:80``, ``:54``, ``:108``) -- as the uint8 buffer the kernel reads."""
    lay = dropout_mask_layout(dims, batch)
    B, N, d, H, dff = batch, dims.max_node, dims.d_model, dims.n_head, dims.d_ff
    parts = [masks["emb"].reshape(B * N, d)]
    for l in range(dims.n_layers):
        attn = masks[f"attn{l}"].reshape(B, H, N, N).permute(0, 2, 1, 3).reshape(B * N, H * N)   # row = (b, query)
        parts.append(torch.cat([attn, masks[f"sa{l}"].reshape(B * N, d), masks[f"ffh{l}"].reshape(B * N, dff),
                                masks[f"ffo{l}"].reshape(B * N, d)], dim=1))
    flat = torch.cat([p.reshape(-1) for p in parts]).to(torch.uint8).contiguous()
    assert flat.numel() == lay["total"]
    return flat


class _DsmStep(torch.autograd.Function):
    """Carries the gradients computed by the HIP step into ``loss.backward()``."""

    @staticmethod
    def forward(ctx, loss, flat_grads, *params):
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(flat_grads)
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        (flat,) = ctx.saved_tensors
        g = flat * grad_out                      # one launch; the parameters' .grad become views of this buffer
        outs, o = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp))
            outs.append(g[o:o + n].view(shp))
            o += n
        return (None, None) + tuple(outs)


def state_tensors(model) -> list:
    """``list(model.state_dict(keep_vars=True).values())``, remembered on the module (``state_dict()`` walks every
    sub-module: 0.4 ms per call, twice per training step).  The tensors' identities do not change while a model lives;
    ``load_state_dict`` copies into them."""
    ts = getattr(model, "_state_tensors", None)
    if ts is None:
        ts = list(model.state_dict(keep_vars=True).values())
        model._state_tensors = ts
        model._state_flat_cache = FlatCache()
    return ts


def dsm_step(model: CATE, sde, x, t, z, mask, *, reduce_mean=True, keep: Optional[torch.Tensor] = None,
             dropout_p: float = 0.0, need_grads: bool = True, out_grads: Optional[torch.Tensor] = None):
    """One call of ``dnag_cate_dsm_step``: returns ``(loss [device scalar], flat_grads or None)`` with
    ``flat_grads`` in ``state_dict()`` order.  ``t`` [B] and ``z`` [B,N,V] are the random draws of
    ``losses.py:139`` and ``:141``; ``keep`` the packed dropout masks (None: dropout off)."""
    lib = _lib.load()
    if not isinstance(model, CATE):
        raise NotImplementedError("the HIP training step is built for the CATE score network")
    dev = _lib.require_device(x, t, z, mask)
    B, N, V = x.shape
    dims = model.dims
    if (N, V) != (dims.max_node, dims.n_vocab):
        raise ValueError(f"x has shape [*, {N}, {V}], model was built for [*, {dims.max_node}, {dims.n_vocab}]")
    xf = x.detach().to(torch.float32).contiguous()
    zf = z.detach().to(torch.float32).contiguous()
    tf = t.detach().to(torch.float32).contiguous()
    # sde.marginal_prob on a unit input gives the per-graph mean coefficient and std with the reference's
    # own torch expressions (sde_lib.py:179-188 / :279-282)
    mean1, std = sde.marginal_prob(torch.ones(B, 1, 1, device=dev), tf)
    mean_coef = mean1.reshape(B).to(torch.float32).contiguous()
    std = std.to(torch.float32).contiguous()
    if isinstance(sde, sde_lib.VESDE):
        labels = std                                                      # models/utils.py:109
    else:
        labels = (tf * 999).contiguous()                                  # models/utils.py:93
    m = mask.detach().to(torch.float32).contiguous()
    if m.dim() == 2:
        m = m.unsqueeze(0)
    if m.shape[1:] != (N, N) or m.shape[0] not in (1, B):
        raise ValueError(f"mask has shape {tuple(m.shape)}, expected [{B}|1, {N}, {N}]")
    params = state_tensors(model)
    flat = model._state_flat_cache.get(params)                 # already one flat buffer once an optimiser step has run
    if flat is None:
        flat = torch.cat([p.detach().reshape(-1) for p in params]).to(torch.float32).contiguous()
    dc = model.dims_c()
    if flat.numel() != lib.dnag_cate_state_floats(C.byref(dc)):
        raise RuntimeError("state_dict size does not match the CATE layout")
    if getattr(model, "_tables_dev", None) is None or model._tables_dev.device != dev:
        model._tables_dev = torch.cat([positional_table(dims.max_node, dims.d_model, dims.stagewise_pos).reshape(-1),
                                       timestep_frequencies(dims.d_model)]).contiguous().to(dev)
    if getattr(model, "_train_ws", None) is None:
        model._train_ws = Workspace()
    ws = model._train_ws.get(lib.dnag_train_workspace_bytes(C.byref(dc), B), dev)
    if keep is not None:
        if keep.dtype != torch.uint8 or not keep.is_contiguous() or keep.device != dev:
            raise ValueError("keep must be a contiguous uint8 tensor on the model's device")
        if keep.numel() != lib.dnag_dropout_mask_bytes(C.byref(dc), B):
            raise ValueError("keep has the wrong size for this batch (see dropout_mask_layout)")
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grads = (out_grads if out_grads is not None else torch.empty_like(flat)) if need_grads else None
    if grads is not None and (grads.numel() != flat.numel() or grads.device != flat.device or grads.dtype != torch.float32):
        raise ValueError("out_grads does not match the parameters")
    a = _lib.TrainArgsC()
    a.cate_dims = C.pointer(dc)
    a.params = _lib.ptr(flat); a.tables = _lib.ptr(model._tables_dev)
    a.batch = B
    a.x = _lib.ptr(xf); a.z = _lib.ptr(zf); a.mean_coef = _lib.ptr(mean_coef); a.std = _lib.ptr(std); a.labels = _lib.ptr(labels)
    a.mask = _lib.ptr(m); a.mask_stride = int(m.shape[0] == B and B > 1)
    a.score_mode = sde.score_mode
    a.reduce_mean = int(bool(reduce_mean))
    a.keep = keep.data_ptr() if keep is not None else None
    a.dropout_p = float(dropout_p) if keep is not None else 0.0
    a.loss = _lib.ptr(loss); a.grads = _lib.ptr(grads)
    a.workspace = ws.data_ptr(); a.workspace_bytes = ws.numel() * 4
    a.stream = _lib.current_stream(dev)
    with torch.cuda.device(dev):
        _lib.check(lib.dnag_cate_dsm_step(C.byref(a)), "dnag_cate_dsm_step")
    return loss.reshape(()), grads


def get_sde_loss_fn_nas(sde, train, reduce_mean=True, continuous=True, likelihood_weighting=True, eps=1e-5):
    """Reference ``losses.py:110-168``.  ``likelihood_weighting=False`` (every shipped config,
    ``configs/tr_scorenet.py:33``) and ``continuous=True`` only.

    Attributes of the returned function, for tests and reproducible runs:
      ``loss_fn.draws = (t, z)``          inject the random draws of ``losses.py:139`` / ``:141`` (consumed once);
      ``loss_fn.dropout_keep = {...}``    inject keep masks (``pack_dropout_masks`` naming; consumed once);
      ``loss_fn.seed``                    base seed of the on-device Philox dropout masks (incremented per call).
    """
    if likelihood_weighting:
        raise NotImplementedError("likelihood_weighting=True is not built (the shipped configs use False)")
    if not continuous:
        raise NotImplementedError("only continuous=True is supported (as in the reference's shipped configs)")

    def run(model, batch, need, out_grads=None):
        x, adj, mask = batch                                              # losses.py:137
        lib = _lib.load()
        dev = _lib.require_device(x, mask)
        B = x.shape[0]
        if loss_fn.draws is not None:
            t, z = loss_fn.draws
            loss_fn.draws = None
        else:
            t = torch.rand(B, device=dev) * (sde.T - eps) + eps          # :139
            z = torch.randn_like(x)                                       # :141
        p = float(getattr(model, "dropout_p", 0.0)) if train else 0.0     # model.train() / .eval(), models/utils.py:63-68
        keep = None
        if p > 0.0:
            if loss_fn.dropout_keep is not None:
                keep = pack_dropout_masks(loss_fn.dropout_keep, model.dims, B).to(dev)
                loss_fn.dropout_keep = None
            else:
                dc = model.dims_c()
                keep = torch.empty(lib.dnag_dropout_mask_bytes(C.byref(dc), B), dtype=torch.uint8, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(lib.dnag_dropout_fill(C.byref(dc), B, p, loss_fn.seed, keep.data_ptr(), _lib.current_stream(dev)),
                               "dnag_dropout_fill")
                loss_fn.seed += 1
        return dsm_step(model, sde, x, t.to(dev), z.to(dev), mask, reduce_mean=reduce_mean, keep=keep, dropout_p=p,
                        need_grads=need, out_grads=out_grads)

    def loss_fn(model, batch):
        need = train and torch.is_grad_enabled()
        loss, grads = run(model, batch, need)
        if not need:
            return loss
        return _DsmStep.apply(loss, grads, *state_tensors(model))

    def fused(model, batch, out_grads=None):
        """``(loss, flat gradients)`` of one training step without the autograd round trip (``loss_fn`` + ``loss.backward()``
        hands the same buffer to the parameters' ``.grad`` through ``torch.autograd``: 1.4 ms of host time per step)."""
        return run(model, batch, True, out_grads)

    loss_fn.fused = fused
    loss_fn.draws = None
    loss_fn.dropout_keep = None
    loss_fn.seed = 0
    return loss_fn


def get_step_fn(sde, train, optimize_fn=None, reduce_mean=False, continuous=True, likelihood_weighting=False, data='NASBench201'):
    """Reference ``losses.py:171-237``: one training / evaluation step on ``state = {'model', 'optimizer',
    'ema', 'step'}``."""
    if continuous:
        if data in ['NASBench201', 'ofa']:
            loss_fn = get_sde_loss_fn_nas(sde, train, reduce_mean=reduce_mean, continuous=True,
                                          likelihood_weighting=likelihood_weighting)
        else:
            raise NotImplementedError(f"Data {data} (search space) is not supported yet.")
    else:
        raise NotImplementedError(f"Discrete training for {sde.__class__.__name__} is not implemented.")

    def step_fn(state, batch):
        model = state['model']
        if train:
            optimizer = state['optimizer']
            if isinstance(optimizer, FusedAdam) and hasattr(state['ema'], 'flat_shadow') and hasattr(loss_fn, 'fused') \
                    and optimizer.params and all(a is b for a, b in zip(optimizer.params, state_tensors(model))) \
                    and len(optimizer.params) == len(state_tensors(model)):
                # The fused path: zero_grad / loss_fn / loss.backward() / optimize_fn / ema.update of the reference's step_fn
                # (losses.py:204-227) as two library calls on flat buffers.  The parameters' .grad are views of the step's
                # gradient buffer, as after loss.backward() (re-attached only when something detached them).
                gbuf = getattr(optimizer, "_grad_buf", None)
                flat_n = sum(p.numel() for p in optimizer.params) if gbuf is None else gbuf.numel()
                if gbuf is None or gbuf.device != optimizer.params[0].device:
                    gbuf = torch.empty(flat_n, dtype=torch.float32, device=optimizer.params[0].device)
                    optimizer._grad_buf = gbuf
                    optimizer._grad_views = None
                loss, grads = loss_fn.fused(model, batch, out_grads=gbuf)
                views = getattr(optimizer, "_grad_views", None)
                if views is None:
                    views, o = [], 0
                    for p in optimizer.params:
                        views.append(gbuf[o:o + p.numel()].view(p.shape))
                        o += p.numel()
                    optimizer._grad_views = views
                if optimizer.params[0].grad is not views[0] or optimizer.params[-1].grad is not views[-1]:
                    for p, v in zip(optimizer.params, views):
                        p.grad = v
                optimize_fn(optimizer, model.parameters(), step=state['step'], ema=state['ema'], flat_grads=grads)
                state['step'] += 1
                return loss
            optimizer.zero_grad()
            loss = loss_fn(model, batch)
            loss.backward()
        else:
            with torch.no_grad():
                ema = state['ema']
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch)
                ema.restore(model.parameters())
        return loss

    step_fn.loss_fn = loss_fn
    return step_fn
