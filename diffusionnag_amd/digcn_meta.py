"""Meta-predictor ``MetaNeuralPredictor`` behind the reference's model surface
(reference ``models/digcn_meta.py:56-159``).

Arch branch (4-layer DiGCN, pooling, time embedding, head) and its input gradient run in
``libdnag.so``.  The dataset branch (``set_encode``, reference ``digcn_meta.py:131-139``, two
SetTransformer pools of ``models/set_encoder``) is evaluated once per sampling call and cached in
``D_mu`` exactly as the reference does; it stays in PyTorch.
"""

from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn.functional as F

from . import _lib, mutils
from .cate import PackedModule, Workspace, build_parameters, timestep_frequencies
from .specs import PredictorDims, predictor_param_spec


def _linear(m, x):
    return F.linear(x, m.weight, getattr(m, "bias", None))


def _mab(m, Q, K, heads):
    """Multihead attention block, reference ``set_encoder/setenc_modules.py:23-37`` (ln=False)."""
    Q = _linear(m.fc_q, Q)
    Kp, V = _linear(m.fc_k, K), _linear(m.fc_v, K)
    dim_v = Q.shape[-1]
    split = dim_v // heads
    Q_ = torch.cat(Q.split(split, 2), 0)
    K_ = torch.cat(Kp.split(split, 2), 0)
    V_ = torch.cat(V.split(split, 2), 0)
    A = torch.softmax(Q_.bmm(K_.transpose(1, 2)) / math.sqrt(dim_v), 2)
    O = torch.cat((Q_ + A.bmm(V_)).split(Q.size(0), 0), 2)
    return O + F.relu(_linear(m.fc_o, O))


def _setpool(m, X, heads):
    """``SetPool(mode='sabPF')``, reference ``set_encoder/setenc_models.py:12-38``."""
    enc, dec = m.enc, m.dec
    h = _mab(getattr(enc, "0").mab, X, X, heads)
    h = _mab(getattr(enc, "1").mab, h, h, heads)
    pma = getattr(dec, "0")
    h = _mab(pma.mab, pma.S.repeat(h.size(0), 1, 1), h, heads)
    return _linear(getattr(dec, "1"), h)


class DatasetBranch:
    """Dataset branch shared by both surrogates (``digcn_meta.py:131-139`` = ``cate.py:330-338``) and the
    ``sample_state`` / ``D_mu`` cache of their ``forward`` (``digcn_meta.py:150-156``, ``cate.py:351-357``).
    Needs ``self.dims.set_heads``, ``self.num_sample``, ``self.intra_setpool``, ``self.inter_setpool``."""

    def set_encode(self, task):
        """``task`` is ``[B, num_class*num_sample, 512]``; the reference encodes every (identical) row of
        the batch, one encoding of each distinct row is enough, so rows are encoded one by one only when
        they differ."""
        heads = self.dims.set_heads
        if task.dim() == 2:
            task = task.unsqueeze(0)
        first = task[0]
        same = task.shape[0] == 1 or bool((task == first.unsqueeze(0)).all())
        rows = [first] if same else list(task)
        out = []
        for x in rows:
            cls_protos = _setpool(self.intra_setpool, x.view(-1, self.num_sample, 512), heads).squeeze(1)
            out.append(_setpool(self.inter_setpool, cls_protos.unsqueeze(0), heads).reshape(-1))
        v = torch.stack(out)
        return v.expand(task.shape[0], -1) if same else v

    def _dataset_embedding(self, task, dev) -> torch.Tensor:
        """One ``[nz]`` dataset embedding for the whole batch, cached like ``digcn_meta.py:151-155``."""
        if self.sample_state and self.D_mu is not None:
            D = self.D_mu
        else:
            with torch.no_grad():
                D = self.set_encode(task)
            if self.sample_state:
                self.D_mu = D
        if D.dim() == 2:
            if D.shape[0] > 1 and not bool((D == D[:1]).all()):
                raise NotImplementedError("per-graph dataset embeddings (different tasks inside one batch) are not built")
            D = D[0]
        return D.detach().to(dev, torch.float32).contiguous()


@mutils.register_model(name='MetaNeuralPredictor')
class MetaNeuralPredictor(DatasetBranch, PackedModule):
    """Drop-in for the reference's ``MetaeuralPredictor`` [sic] (``digcn_meta.py:56-159``): same
    constructor argument, ``state_dict``, ``forward(X, time_cond, maskX, task)`` and the
    ``sample_state`` / ``D_mu`` caching attributes the sampler toggles (``sampling.py:561-562``)."""

    def __init__(self, config):
        super().__init__()
        ge = config.model.graph_encoder
        if config.model.nonlinearity.lower() != 'swish':
            raise NotImplementedError("the HIP time-embedding MLP implements 'swish' only")
        if config.model.input_type != 'DA':
            raise NotImplementedError("only input_type='DA' (dataset + architecture, the shipped config) is supported")
        self.dims = PredictorDims(max_node=8, n_vocab=ge.initial_hidden, gcn_hidden=ge.gcn_hidden,
                                  gcn_layers=ge.gcn_layers, linear_hidden=ge.linear_hidden, hs=config.model.hs,
                                  nz=config.model.nz, num_sample=config.model.num_sample)
        self.num_sample = config.model.num_sample
        self.nz = config.model.nz
        build_parameters(self, predictor_param_spec(self.dims))
        self.sample_state = False
        self.D_mu = None
        self._ws = Workspace()

    # -- packing ---------------------------------------------------------------------------------
    def dims_c(self) -> _lib.PredDimsC:
        d = self.dims
        return _lib.PredDimsC(d.max_node, d.n_vocab, d.gcn_hidden, d.gcn_layers, d.linear_hidden, d.hs, d.nz)

    def _state_flat(self) -> torch.Tensor:
        sd = self.state_dict()
        keys = []
        for l in range(self.dims.gcn_layers):
            keys += [f"gcn.{l}.weight1", f"gcn.{l}.weight2"]
        keys += ["fc1.weight", "timeEmb1.weight", "timeEmb1.bias", "timeEmb2.weight", "timeEmb2.bias",
                 "pred_fc.0.weight", "pred_fc.0.bias", "pred_fc.2.weight", "pred_fc.2.bias"]
        return torch.cat([sd[k].reshape(-1) for k in keys])

    def _pack_host(self, flat: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        dc = self.dims_c()
        if flat.numel() != lib.dnag_pred_state_floats(C.byref(dc)):
            raise RuntimeError("state_dict size does not match the MetaNeuralPredictor layout")
        tables = timestep_frequencies(self.dims.gcn_hidden).contiguous()
        out = torch.zeros(lib.dnag_pred_packed_floats(C.byref(dc)), dtype=torch.float32)
        _lib.check(lib.dnag_pred_pack(C.byref(dc), flat.data_ptr(), tables.data_ptr(), out.data_ptr()), "dnag_pred_pack")
        return out

    # -- HIP calls -----------------------------------------------------------------------------------
    def _run(self, X, time_cond, maskX, task, want_y, want_grad):
        if self.training:
            raise NotImplementedError("MetaNeuralPredictor on the HIP path is eval-only")
        lib = _lib.load()
        dev = _lib.require_device(X, time_cond, maskX)
        B, N, V = X.shape
        if (N, V) != (self.dims.max_node, self.dims.n_vocab):
            raise ValueError(f"X has shape [*, {N}, {V}], surrogate was built for [*, {self.dims.max_node}, {self.dims.n_vocab}]")
        assert len(time_cond.shape) == 1                               # digcn_meta.py:163
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
        ws = self._ws.get(lib.dnag_pred_workspace_bytes(C.byref(dc), max(B, 1)), dev)
        y = torch.empty(B, dtype=torch.float32, device=dev) if want_y else None
        g = torch.empty(B, N, V, dtype=torch.float32, device=dev) if want_grad else None
        with torch.cuda.device(dev):
            rc = lib.dnag_pred_forward_grad(C.byref(dc), _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t),
                                            int(t.numel() == B and B > 1), _lib.ptr(m), int(m.shape[0] == B and B > 1),
                                            _lib.ptr(d_mu), _lib.ptr(y), _lib.ptr(g), B, ws.data_ptr(), ws.numel() * 4,
                                            _lib.current_stream(dev))
        _lib.check(rc, "dnag_pred_forward_grad")
        return y, g

    def forward(self, X, time_cond, maskX, task):
        """``y_pred [B, 1]`` (reference ``digcn_meta.py:150-159``)."""
        y, _ = self._run(X, time_cond, maskX, task, True, False)
        return y.unsqueeze(1)

    def input_grad(self, X, time_cond, maskX, task):
        """``d(sum_b y_b)/dX`` -- what ``torch.autograd.grad(logit.sum(), x_in)`` returns at reference
        ``models/utils.py:136-144``, computed by the kernel's hand-derived backward pass."""
        _, g = self._run(X, time_cond, maskX, task, False, True)
        return g
