"""OFA meta-predictor ``MetaPredictorCATE`` behind the reference's model surface (reference
``NAS-Bench-201/models/cate.py:251-360``; the same class is ``MobileNetV3/models/cate.py:181-289``).

Arch branch (embedding without positional table, 2-layer graph transformer, ``final`` and
``regeress`` MLPs, tanh head) and its input gradient run in ``libdnag.so``
(``dnag_predcate_forward_grad``).  The dataset branch is the shared PyTorch set encoder, evaluated
once per sampling call (``digcn_meta.DatasetBranch``).
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib, mutils
from .cate import PackedModule, Workspace, build_parameters, timestep_frequencies
from .digcn_meta import DatasetBranch
from .specs import PredictorCateDims, predictor_cate_param_spec


@mutils.register_model(name='MetaPredictorCATE')
class MetaPredictorCATE(DatasetBranch, PackedModule):
    """Drop-in for the reference's ``MetaPredictorCATE``: same constructor argument, ``state_dict``,
    ``forward(X, time_cond, maskX, task)`` and ``sample_state`` / ``D_mu`` attributes."""

    def __init__(self, config):
        super().__init__()
        ge = config.model.graph_encoder
        if config.model.nonlinearity.lower() != 'swish':
            raise NotImplementedError("the HIP time-embedding MLP implements 'swish' only")
        if config.model.input_type != 'DA':
            raise NotImplementedError("only input_type='DA' (dataset + architecture, the shipped config) is supported")
        self.dims = PredictorCateDims(max_node=config.data.max_node, n_vocab=config.data.n_vocab, d_model=ge.d_model,
                                      n_head=ge.n_head, d_ff=ge.d_ff, n_layers=ge.n_layers, hs=config.model.hs,
                                      nz=config.model.nz, num_sample=config.model.num_sample)
        self.input_type = config.model.input_type
        self.hs = config.model.hs
        self.num_sample = config.model.num_sample
        self.nz = config.model.nz
        build_parameters(self, predictor_cate_param_spec(self.dims))
        self.sample_state = False
        self.D_mu = None
        self._ws = Workspace()

    # -- packing ---------------------------------------------------------------------------------
    def dims_c(self) -> _lib.PredCateDimsC:
        d = self.dims
        return _lib.PredCateDimsC(d.max_node, d.n_vocab, d.d_model, d.n_head, d.d_ff, d.n_layers, d.hs, d.nz)

    def _state_flat(self) -> torch.Tensor:
        """The parameters used while sampling, in ``state_dict`` order (the set pools and the unused
        ``set_fc`` are skipped)."""
        sd = self.state_dict()
        skip = ("intra_setpool.", "inter_setpool.", "set_fc.")
        return torch.cat([v.reshape(-1) for k, v in sd.items() if not k.startswith(skip)])

    def _pack_host(self, flat: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        dc = self.dims_c()
        if flat.numel() != lib.dnag_predcate_state_floats(C.byref(dc)):
            raise RuntimeError("state_dict size does not match the MetaPredictorCATE layout")
        tables = timestep_frequencies(self.dims.d_model).contiguous()
        out = torch.zeros(lib.dnag_predcate_packed_floats(C.byref(dc)), dtype=torch.float32)
        _lib.check(lib.dnag_predcate_pack(C.byref(dc), flat.data_ptr(), tables.data_ptr(), out.data_ptr()),
                   "dnag_predcate_pack")
        return out

    # -- HIP calls -----------------------------------------------------------------------------------
    def _run(self, X, time_cond, maskX, task, want_y, want_grad):
        if self.training:
            raise NotImplementedError("MetaPredictorCATE on the HIP path is eval-only")
        lib = _lib.load()
        dev = _lib.require_device(X, time_cond, maskX)
        B, N, V = X.shape
        if (N, V) != (self.dims.max_node, self.dims.n_vocab):
            raise ValueError(f"X has shape [*, {N}, {V}], surrogate was built for [*, {self.dims.max_node}, {self.dims.n_vocab}]")
        assert len(time_cond.shape) == 1                               # cate.py:364
        x = X.detach().to(torch.float32).contiguous()
        t = time_cond.detach().to(torch.float32).contiguous()
        m = maskX.detach().to(torch.float32).contiguous()
        if m.dim() == 2:
            m = m.unsqueeze(0)
        if t.numel() not in (1, B) or m.shape[1:] != (N, N) or m.shape[0] not in (1, B):
            raise ValueError("time_cond / maskX do not match the batch")
        d_mu = self._dataset_embedding(task, dev)
        dc = self.dims_c()
        packed = self.packed(dev)
        ws = self._ws.get(lib.dnag_predcate_workspace_bytes(C.byref(dc), B), dev)
        y = torch.empty(B, dtype=torch.float32, device=dev) if want_y else None
        g = torch.empty(B, N, V, dtype=torch.float32, device=dev) if want_grad else None
        with torch.cuda.device(dev):
            rc = lib.dnag_predcate_forward_grad(C.byref(dc), _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t),
                                                int(t.numel() == B and B > 1), _lib.ptr(m), int(m.shape[0] == B and B > 1),
                                                _lib.ptr(d_mu), _lib.ptr(y), _lib.ptr(g), B, ws.data_ptr(), ws.numel() * 4,
                                                _lib.current_stream(dev))
        _lib.check(rc, "dnag_predcate_forward_grad")
        return y, g

    def forward(self, X, time_cond, maskX, task):
        """``y_pred [B, 1]`` (reference ``cate.py:351-360``)."""
        y, _ = self._run(X, time_cond, maskX, task, True, False)
        return y.unsqueeze(1)

    def input_grad(self, X, time_cond, maskX, task):
        """``d(sum_b y_b)/dX`` -- what ``torch.autograd.grad(logit.sum(), x_in)`` returns at reference
        ``models/utils.py:136-144``, computed by the hand-derived backward pass."""
        _, g = self._run(X, time_cond, maskX, task, False, True)
        return g
