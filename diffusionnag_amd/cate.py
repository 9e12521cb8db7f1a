"""Score network ``CATE`` behind the reference's model surface (reference ``models/cate.py:108-164``).

The module owns parameters with the reference's ``state_dict`` names and shapes (so reference
checkpoints load with ``load_state_dict``); ``forward`` hands the packed weights and the inputs
to ``dnag_cate_forward`` in ``libdnag.so``.  No PyTorch op takes part in the forward pass.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib, mutils
from .specs import CateDims, cate_param_spec


def build_parameters(module: nn.Module, spec) -> None:
    """Register parameters under dotted names, creating plain container modules on the way, in
    ``state_dict()`` order.  Initial values follow the reference constructors' distributions."""
    for name, shape, init in spec:
        parts = name.split(".")
        m = module
        for p in parts[:-1]:
            if not hasattr(m, p):
                m.add_module(p, nn.Module())
            m = getattr(m, p)
        t = torch.empty(*shape)
        kind = init[0]
        if kind in ("lin_w", "lin_b"):
            b = 1.0 / math.sqrt(init[1])
            nn.init.uniform_(t, -b, b)
        elif kind == "ones":
            nn.init.ones_(t)
        elif kind == "zeros":
            nn.init.zeros_(t)
        elif kind == "xavier":
            a = math.sqrt(6.0 / (init[1] + init[2]))
            nn.init.uniform_(t, -a, a)
        else:
            raise ValueError(kind)
        m.register_parameter(parts[-1], nn.Parameter(t))


def timestep_frequencies(dim: int, max_positions: int = 10000) -> torch.Tensor:
    """The fp32 frequency vector of ``get_timestep_embedding`` (reference ``cate.py:363-368``)."""
    half = dim // 2
    emb = math.log(max_positions) / (half - 1)
    return torch.exp(torch.arange(half, dtype=torch.float32) * -emb)


def positional_table(max_node: int, d_model: int, stagewise: bool) -> torch.Tensor:
    """``PositionalEncoding_Cell`` (``cate.py:231-248``) / ``_StageWise`` (``cate.py:211-228``):
    a plain attribute of the reference module, not part of its ``state_dict``."""
    stages = 5 if stagewise else 1
    length = int(max_node / stages)
    enc = torch.zeros(length, d_model)
    pos = torch.arange(0, length).float().unsqueeze(1)
    two_i = torch.arange(0, d_model, step=2).float()
    enc[:, 0::2] = torch.sin(pos / (10000 ** (two_i / d_model)))
    enc[:, 1::2] = torch.cos(pos / (10000 ** (two_i / d_model)))
    return torch.cat([enc] * stages, dim=0)


class PackedModule(nn.Module):
    """Keeps a device copy of the kernel-layout weights, rebuilt when a parameter changes."""

    def __init__(self):
        super().__init__()
        self._pack_key = None
        self._packed = None

    def _state_flat(self) -> torch.Tensor:   # pragma: no cover - abstract
        raise NotImplementedError

    def _pack_host(self, flat: torch.Tensor) -> torch.Tensor:   # pragma: no cover - abstract
        raise NotImplementedError

    def packed(self, device: torch.device) -> torch.Tensor:
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._pack_key != key:
            with torch.no_grad():
                host = self._pack_host(self._state_flat().detach().to("cpu", torch.float32).contiguous())
            self._packed = host.to(device)
            self._pack_key = key
        return self._packed


class Workspace:
    """Grow-only device scratch shared by the calls of one module / sampler."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device: torch.device) -> torch.Tensor:
        if self.buf is None or self.buf.device != device or self.buf.numel() * 4 < nbytes:
            self.buf = None
            self.buf = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)
        return self.buf


@mutils.register_model(name='CATE')
class CATE(PackedModule):
    """Drop-in for the reference's ``CATE`` (``models/cate.py:108-164``): same constructor argument,
    same ``state_dict``, same ``forward(X, time_cond, maskX)`` contract (eval mode)."""

    def __init__(self, config):
        super().__init__()
        ge = config.model.graph_encoder
        if config.model.nonlinearity.lower() != 'swish':               # get_act, cate.py:116
            raise NotImplementedError("the HIP time-embedding MLP implements 'swish' (the shipped configs) only")
        stagewise = False
        if 'pos_enc_type' in config.model and config.model.pos_enc_type == 2:
            stagewise = config.data.name != 'NASBench201'       # cate.py:128-136
        elif 'pos_enc_type' in config.model:
            raise NotImplementedError("only pos_enc_type=2 (the shipped configs) is supported")
        self.dims = CateDims(max_node=config.data.max_node, n_vocab=config.data.n_vocab, d_model=ge.d_model,
                             n_head=ge.n_head, d_ff=ge.d_ff, n_layers=ge.n_layers, stagewise_pos=stagewise)
        self.has_pos = 'pos_enc_type' in config.model
        if not self.has_pos:
            raise NotImplementedError("CATE without positional encoding is not built")
        # dropout rates of the training-mode forward (cate.py:114, transformer.py:32-103); the training
        # kernels take one rate for every site, as every shipped config has (configs/tr_scorenet.py:92, :100)
        self.dropout_p = float(config.model.dropout) if 'dropout' in config.model else 0.0
        ge_p = float(ge.dropout) if 'dropout' in ge else self.dropout_p
        if ge_p != self.dropout_p:
            raise NotImplementedError("model.dropout and model.graph_encoder.dropout must be equal on the HIP path")
        build_parameters(self, cate_param_spec(self.dims))
        self._ws = Workspace()

    # -- packing ---------------------------------------------------------------------------------
    def dims_c(self) -> _lib.CateDimsC:
        d = self.dims
        return _lib.CateDimsC(d.max_node, d.n_vocab, d.d_model, d.n_head, d.d_ff, d.n_layers, int(d.stagewise_pos))

    def _state_flat(self) -> torch.Tensor:
        return torch.cat([p.reshape(-1) for p in self.state_dict().values()])

    def _pack_host(self, flat: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        dc = self.dims_c()
        if flat.numel() != lib.dnag_cate_state_floats(C.byref(dc)):
            raise RuntimeError("state_dict size does not match the CATE layout")
        tables = torch.cat([positional_table(self.dims.max_node, self.dims.d_model, self.dims.stagewise_pos).reshape(-1),
                            timestep_frequencies(self.dims.d_model)]).contiguous()
        out = torch.zeros(lib.dnag_cate_packed_floats(C.byref(dc)), dtype=torch.float32)
        _lib.check(lib.dnag_cate_pack(C.byref(dc), flat.data_ptr(), tables.data_ptr(), out.data_ptr()), "dnag_cate_pack")
        return out

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, X, time_cond, maskX, *, return_concat=False, return_attn0=False):
        if self.training:
            raise NotImplementedError("CATE.forward on the HIP path is eval-only; the training-mode forward lives in the "
                                      "fused DSM step (diffusionnag_amd.losses.get_sde_loss_fn_nas)")
        lib = _lib.load()
        dev = _lib.require_device(X, time_cond, maskX)
        B, N, V = X.shape
        if (N, V) != (self.dims.max_node, self.dims.n_vocab):
            raise ValueError(f"X has shape [*, {N}, {V}], model was built for [*, {self.dims.max_node}, {self.dims.n_vocab}]")
        assert len(time_cond.shape) == 1                               # cate.py:364
        x = X.detach().to(torch.float32).contiguous()
        t = time_cond.detach().to(torch.float32).contiguous()
        m = maskX.detach().to(torch.float32).contiguous()
        if t.numel() not in (1, B):
            raise ValueError(f"time_cond has {t.numel()} entries for a batch of {B}")
        if m.dim() == 2:
            m = m.unsqueeze(0)
        if m.shape[1:] != (N, N) or m.shape[0] not in (1, B):
            raise ValueError(f"maskX has shape {tuple(m.shape)}, expected [{B}|1, {N}, {N}]")
        dc = self.dims_c()
        packed = self.packed(dev)
        nbytes = lib.dnag_workspace_bytes(C.byref(dc), None, B, 1)
        ws = self._ws.get(nbytes, dev)
        out = torch.empty(B, N, V, dtype=torch.float32, device=dev)
        concat = torch.empty(B, N, self.dims.fdim, dtype=torch.float32, device=dev) if return_concat else None
        attn0 = torch.empty(B, self.dims.n_head, N, N, dtype=torch.float32, device=dev) if return_attn0 else None
        with torch.cuda.device(dev):
            rc = lib.dnag_cate_forward(C.byref(dc), _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t), int(t.numel() == B and B > 1),
                                       _lib.ptr(m), int(m.shape[0] == B and B > 1), _lib.ptr(out), _lib.ptr(concat),
                                       _lib.ptr(attn0), B, ws.data_ptr(), ws.numel() * 4, _lib.current_stream(dev))
        _lib.check(rc, "dnag_cate_forward")
        if return_concat or return_attn0:
            return out, concat, attn0
        return out


def __getattr__(name):
    """``MetaPredictorCATE`` lives in the reference's ``models/cate.py``; here it is defined in
    ``predictor_cate`` (it needs the shared dataset branch) and re-exported lazily."""
    if name == "MetaPredictorCATE":
        from .predictor_cate import MetaPredictorCATE
        return MetaPredictorCATE
    raise AttributeError(name)
