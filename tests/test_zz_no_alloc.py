"""dnag_pred_forward_grad_noalloc (include/dnag.h): the allocation-free variant of dnag_pred_forward_grad, called through
the C ABI and compared bit for bit with the existing entry point on two calls in a row, the second with a smaller
input (the library's scratch array keeps the first call's contents)."""

import ctypes as C

import numpy as np
import pytest
import torch

from diffusionnag_amd import _lib, digcn_meta as _dm, mutils, synthetic  # noqa: F401
from diffusionnag_amd.specs import PredictorDims
from tests.helpers import predictor_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, seed):
    rs = np.random.RandomState(seed)
    X = torch.from_numpy(rs.standard_normal((B, 8, 7)).astype(np.float32)).to(DEV)
    t = torch.from_numpy((rs.uniform(1e-3, 1, B) * 999).astype(np.float32)).to(DEV)      # one time condition per graph
    m = torch.from_numpy((rs.uniform(0, 1, (B, 8, 8)) < 0.4).astype(np.float32)).to(DEV)
    return X, t, m


def test_noalloc_entry_point_matches_workspace_entry_point():
    lib = _lib.load()
    d = PredictorDims()
    clf = mutils.create_model(predictor_config(d, DEV))
    clf.load_state_dict(synthetic.predictor_state_dict(d, seed=1))
    clf.eval()
    dev = torch.device(DEV)
    packed = clf.packed(dev)
    dc = clf.dims_c()
    d_mu = torch.from_numpy(np.random.RandomState(3).standard_normal(d.nz).astype(np.float32)).to(DEV)
    st = _lib.current_stream(dev)
    for B, seed in ((200, 11), (37, 12)):       # the second call is smaller: stale rows of the first stay in the scratch
        X, t, m = _inputs(B, seed)
        ws = torch.empty(lib.dnag_pred_workspace_bytes(C.byref(dc), B) // 4, dtype=torch.float32, device=DEV)
        y0, g0 = torch.empty(B, device=DEV), torch.empty(B, 8, 7, device=DEV)
        y1, g1 = torch.full((B,), float("nan"), device=DEV), torch.full((B, 8, 7), float("nan"), device=DEV)
        _lib.check(lib.dnag_pred_forward_grad(C.byref(dc), packed.data_ptr(), X.data_ptr(), t.data_ptr(), 1, m.data_ptr(), 1,
                                              d_mu.data_ptr(), y0.data_ptr(), g0.data_ptr(), B, ws.data_ptr(), ws.numel() * 4, st),
                   "dnag_pred_forward_grad")
        before = torch.cuda.memory_allocated(dev)
        _lib.check(lib.dnag_pred_forward_grad_noalloc(C.byref(dc), packed.data_ptr(), X.data_ptr(), t.data_ptr(), 1, m.data_ptr(), 1,
                                                      d_mu.data_ptr(), y1.data_ptr(), g1.data_ptr(), B, st),
                   "dnag_pred_forward_grad_noalloc")
        assert torch.cuda.memory_allocated(dev) == before
        torch.cuda.synchronize()
        assert torch.isfinite(y1).all() and torch.isfinite(g1).all()
        assert torch.equal(y0, y1), f"B={B}: y differs, max {float((y0 - y1).abs().max()):.3e}"
        assert torch.equal(g0, g1), f"B={B}: grad differs, max {float((g0 - g1).abs().max()):.3e}"
    # one shared time condition (time_stride 0), y only
    X, t, m = _inputs(64, 13)
    ws = torch.empty(lib.dnag_pred_workspace_bytes(C.byref(dc), 1) // 4, dtype=torch.float32, device=DEV)
    y0, y1 = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    _lib.check(lib.dnag_pred_forward_grad(C.byref(dc), packed.data_ptr(), X.data_ptr(), t.data_ptr(), 0, m.data_ptr(), 0,
                                          d_mu.data_ptr(), y0.data_ptr(), None, 64, ws.data_ptr(), ws.numel() * 4, st), "ws")
    _lib.check(lib.dnag_pred_forward_grad_noalloc(C.byref(dc), packed.data_ptr(), X.data_ptr(), t.data_ptr(), 0, m.data_ptr(), 0,
                                                  d_mu.data_ptr(), y1.data_ptr(), None, 64, st), "noalloc")
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


def test_noalloc_entry_point_rejects_oversized_batch():
    lib = _lib.load()
    d = PredictorDims()
    dc = _lib.PredDimsC(d.max_node, d.n_vocab, d.gcn_hidden, d.gcn_layers, d.linear_hidden, d.hs, d.nz)
    one = torch.zeros(8, device=DEV)
    rc = lib.dnag_pred_forward_grad_noalloc(C.byref(dc), one.data_ptr(), one.data_ptr(), one.data_ptr(), 0, one.data_ptr(), 0,
                                            one.data_ptr(), None, None, 4097, None)
    assert rc == -1
    assert b"DNAG_NOALLOC_MAX_BATCH" in lib.dnag_last_error()
