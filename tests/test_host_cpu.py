"""CPU-side tests: the C-ABI library loads and exports every declared symbol, host-side weight packing,
the schedule table, the reference-shaped host interface and its error behaviour, and the
world_size-2 shard/gather path on gloo.  No kernel is launched here."""

import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from diffusionnag_amd import _lib, mutils, sampling, sde_lib, synthetic, utils
from diffusionnag_amd import cate as _cate, digcn_meta as _dm, predictor_cate as _pcm  # noqa: F401
from diffusionnag_amd import distributed as D
from diffusionnag_amd.specs import CateDims, PredictorDims, cate_param_spec, n_params, predictor_param_spec
from oracle import dnag_oracle as O
from diffusionnag_amd.specs import PredictorCateDims, predictor_cate_param_spec
from tests.helpers import predictor_cate_config, predictor_config, score_config

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unfrag(flat, rows, K):
    """Inverse of frag_pack (csrc/dnag_common.h): fragment-major floats -> W[rows][K]."""
    nt_all = (rows + 31) // 32
    a = np.asarray(flat[: nt_all * 32 * K]).reshape(nt_all, K // 8, 2, 32, 4)      # [nt][q][h][r][e]
    W = a.transpose(0, 3, 2, 1, 4).reshape(nt_all * 32, K)                          # row = nt*32+r ; k = h*K/2 + 4q + e
    assert not W[rows:].any()
    return W[:rows]


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    assert lib.dnag_abi_version() == 1
    header = open(os.path.join(REPO, "include", "dnag.h")).read()
    declared = set(re.findall(r"\b(dnag_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layout_matches_header_field_order():
    header = open(os.path.join(REPO, "include", "dnag.h")).read()
    body = header[header.index("typedef struct dnag_pc_args {"):header.index("} dnag_pc_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = body[body.index("{") + 1:]
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        for part in decl.split(","):
            names.append(re.sub(r"[\*\s]", " ", part).split()[-1])
    assert names == [f[0] for f in _lib.PcArgsC._fields_]


def test_sizes_and_arg_errors():
    lib = _lib.load()
    d = _lib.CateDimsC(8, 7, 64, 8, 128, 12, 0)
    p = _lib.PredDimsC(8, 7, 144, 4, 128, 32, 56)
    assert lib.dnag_cate_state_floats(C.byref(d)) == n_params(cate_param_spec(CateDims())) == 3989575
    spec = {k: s for k, s, _ in predictor_param_spec(PredictorDims())}
    used = [k for k in spec if k.startswith(("gcn.", "fc1.", "timeEmb", "pred_fc."))]
    assert lib.dnag_pred_state_floats(C.byref(p)) == sum(int(np.prod(spec[k])) for k in used)
    assert lib.dnag_workspace_bytes(C.byref(d), C.byref(p), 4096, 1000) > 4096 * 8 * (768 + 2 * 1536) * 4
    bad = _lib.CateDimsC(9, 7, 64, 8, 128, 12, 0)
    buf = torch.zeros(16)
    rc = lib.dnag_cate_pack(C.byref(bad), buf.data_ptr(), buf.data_ptr(), buf.data_ptr())
    assert rc == -1 and b"unsupported graph shape" in lib.dnag_last_error()
    with pytest.raises(RuntimeError, match="unsupported graph shape"):
        _lib.check(rc, "dnag_cate_pack")
    args = _lib.PcArgsC()
    assert lib.dnag_pc_sample(C.byref(args)) == -1          # NULL dims rejected before any launch


def test_cate_pack_folds_layernorm_into_linear():
    dims = CateDims()
    model = mutils.create_model(score_config(dims, "cpu"))
    sd = synthetic.cate_state_dict(dims, seed=0, perturb_norms=True)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    packed = model._pack_host(model._state_flat().contiguous()).numpy().astype(np.float64)
    # layout constants restated from csrc/dnag_common.h
    d, V, dff = 64, 7, 128
    layer0 = d * V + d + 4 * d * d + 4 * d + 4 * d * d + d
    stride = 3 * d * d + 3 * d + d * d + d + dff * d + dff + d * dff + d + 2 * d + 8 + (7 * d + dff)   # + scales + vector copy
    rs = np.random.RandomState(0)
    h = rs.standard_normal((5, d))
    for l in (0, 7):
        base = layer0 + l * stride
        Wqkv = unfrag(packed[base: base + 3 * d * d], 3 * d, d)
        bqkv = packed[base + 3 * d * d: base + 3 * d * d + 3 * d]
        p = f"graph_encoder.encoder_f.layers.{l}.self_attn"
        g, b = sd[p + ".norm.weight"].double().numpy(), sd[p + ".norm.bias"].double().numpy()
        xhat = (h - h.mean(1, keepdims=True)) / np.sqrt(h.var(1, keepdims=True) + 1e-5)
        for j in range(3):
            W = sd[p + f".attn.linears.{j}.weight"].double().numpy()
            bb = sd[p + f".attn.linears.{j}.bias"].double().numpy()
            ref = (xhat * g + b) @ W.T + bb
            got = xhat @ Wqkv[j * d:(j + 1) * d].T + bqkv[j * d:(j + 1) * d]
            assert np.abs(ref - got).max() < 1e-6
    # head weights land verbatim
    f0 = layer0 + 12 * stride
    assert np.array_equal(packed[f0: f0 + 1536 * 768].astype(np.float32), sd["final.linears.0.weight"].numpy().ravel())
    # derived tables = the oracle's restatement of the reference constructors
    n_split = 1536 * 768 + 1536 * 1536 + 1536          # split-pair copies of final.linears.{0,1} and the folded bias of .0 ...
    n_l16 = 12 * stride                                # ... and the fp16-fragment copy of the layer blocks close the blob
    n_hsc = 16                                         # head GEMM operand scales, between the two
    tail = packed[-(8 * 64 + 32) - n_split - n_hsc - n_l16: -n_split - n_hsc - n_l16]
    assert np.allclose(tail[: 8 * 64], O.positional_table(8, 64, False).numpy().ravel(), atol=0)
    assert np.allclose(tail[8 * 64:], _cate.timestep_frequencies(64).numpy(), atol=0)
    # split-pair rows (csrc/dnag_gemm16.hip.h): groups of 8 values = 8 fp16 hi then 8 fp16 lo of 256 w; hi + lo
    # reproduces the fp32 weight to 22 significant bits
    raw_all = model._pack_host(model._state_flat().contiguous()).numpy()
    raw = raw_all[-n_split - n_hsc - n_l16: -n_hsc - n_l16]
    # ordinary weights: every operand scale is the default (16 for activations, 256 for weights)
    hsc = raw_all[-n_hsc - n_l16: -n_l16]
    assert hsc[0] == 16.0 and hsc[4] == 1.0 / 4096 and hsc[5] == 16.0
    assert list(hsc[8:12]) == [1.0 / 4096, 16.0, 256.0, 1.0 / 4096]
    for l in range(12):
        sc = raw_all[layer0 + l * stride + stride - 8 - 576: layer0 + (l + 1) * stride - 576]
        assert list(sc[:6]) == [1.0 / 4096, 16.0, 1.0 / 4096, 1.0 / 4096, 16.0, 1.0 / 4096], (l, sc)
    # ... stored k-tile blocked (kblock_rows): [k tile of 32][row][32 dwords]
    rows_major = raw[: 1536 * 768].reshape(768 // 32, 1536, 32).transpose(1, 0, 2).reshape(-1).copy()
    halves = rows_major.view(np.float16).reshape(1536, 768 // 8, 2, 8).astype(np.float64)
    # final.linears.0 carries the affine of the per-layer output norms (the encoders write LN(h) itself as the concat):
    # W0' = W0 diag(gamma), b0' = b0 + W0 beta  (transformer.py:177, cate.py:83-86, :55-73)
    gam = np.concatenate([sd[f"graph_encoder.encoder_f.norms.{l}.weight"].double().numpy() for l in range(12)])
    bet = np.concatenate([sd[f"graph_encoder.encoder_f.norms.{l}.bias"].double().numpy() for l in range(12)])
    w0_plain = sd["final.linears.0.weight"].double().numpy()
    w0 = (sd["final.linears.0.weight"].numpy() * gam.astype(np.float32)[None, :]).astype(np.float64)
    b0f = raw[1536 * 768 + 1536 * 1536:].astype(np.float64)
    assert np.abs(b0f - (sd["final.linears.0.bias"].double().numpy() + w0_plain @ bet)).max() < 1e-6
    hrow = rs.standard_normal((3, 768))
    assert np.abs((hrow * gam + bet) @ w0_plain.T + sd["final.linears.0.bias"].double().numpy() - (hrow @ w0.T + b0f)).max() < 1e-5
    rec = (halves[:, :, 0, :] + halves[:, :, 1, :]).reshape(1536, 768) / 256.0
    assert np.abs(rec - w0).max() <= 2.0 ** -21 * np.abs(w0).max()
    assert np.array_equal(halves[:, :, 0, :].reshape(1536, 768), (w0 * 256).astype(np.float16).astype(np.float64))
    # fp16 fragment copy of layer 7's Wo (csrc/dnag_common.h: frag16_pack): chunk q of lane l holds the hi (q even) or lo
    # (q odd) halves of 256 W[nt*32 + r][h*32 + 8(q>>1) .. +7]
    l16 = raw_all[-n_l16:]
    wo_off = 7 * stride + 3 * d * d + 3 * d
    fr = l16[wo_off: wo_off + d * d].view(np.float16).reshape(2, 8, 64, 8).astype(np.float64)      # [nt][q][lane][j]
    Wo = sd["graph_encoder.encoder_f.layers.7.self_attn.attn.linears.3.weight"].double().numpy()
    rec = np.zeros((64, 64))
    for nt in range(2):
        for q in range(8):
            for lane in range(64):
                r_, h_ = lane & 31, lane >> 5
                rec[nt * 32 + r_, h_ * 32 + 8 * (q >> 1): h_ * 32 + 8 * (q >> 1) + 8] += fr[nt, q, lane]
    assert np.abs(rec / 256.0 - Wo).max() <= 2.0 ** -21 * np.abs(Wo).max()


def test_cate_pack_scales_follow_large_weights():
    """Operand scales of the split arithmetic (csrc/dnag_common.h "Operand scales"): powers of two derived from the
    weights so that scale * magnitude <= 60000 for every weight and every reachable activation."""
    dims = CateDims()
    model = mutils.create_model(score_config(dims, "cpu"))
    sd = synthetic.cate_state_dict(dims, seed=0)
    sd["final.linears.1.weight"][5, 7] = 300.0                                          # 256 * 300 > 65504
    sd["graph_encoder.encoder_f.layers.2.feed_forward.feed_forward.w_1.weight"] *= 4000.0   # FFN hidden beyond 4094
    sd["graph_encoder.encoder_f.norms.4.weight"] *= 900.0                               # folded into final.linears.0: W0 diag(gamma)
    model.load_state_dict(sd)
    raw = model._pack_host(model._state_flat().contiguous()).numpy()
    d, V, dff = 64, 7, 128
    layer0 = d * V + d + 4 * d * d + 4 * d + 4 * d * d + d
    stride = 3 * d * d + 3 * d + d * d + d + dff * d + dff + d * dff + d + 2 * d + 8 + (7 * d + dff)
    n_l16 = 12 * stride
    hsc = raw[-16 - n_l16: -n_l16].astype(np.float64)
    sa_c, sa1, sa2, sw3 = hsc[0], hsc[5], hsc[9], hsc[10]
    sw0, sw1 = 1 / (hsc[4] * sa_c), 1 / (hsc[8] * sa1)
    for v in (sa_c, sa1, sa2, sw0, sw1, sw3):
        assert v > 0 and np.log2(v) == int(np.log2(v))                                   # powers of two
    assert sw3 == 256 and sw1 == 128 and sw1 * 300 <= 60000 < 2 * sw1 * 300
    cbound = 8.0                                                                         # the concat is LN(h) itself: |LN(h)| < 8
    assert sa_c == 16
    gam = np.concatenate([sd[f"graph_encoder.encoder_f.norms.{l}.weight"].double().numpy() for l in range(12)])
    bet = np.concatenate([sd[f"graph_encoder.encoder_f.norms.{l}.bias"].double().numpy() for l in range(12)])
    w0_plain = sd["final.linears.0.weight"].double().numpy()
    w0 = w0_plain * gam[None, :]
    b0f = sd["final.linears.0.bias"].double().numpy() + w0_plain @ bet
    assert sw0 * np.abs(w0).max() <= 60000 < 2 * sw0 * np.abs(w0).max() or sw0 == 256
    h1b = max(1.0, (np.abs(w0).sum(1) * cbound + np.abs(b0f)).max())
    assert sa1 * h1b <= 60000 * 1.001 and (sa1 == 16 or 2 * sa1 * h1b > 60000)
    w1 = sd["final.linears.1.weight"].double().numpy()
    h2b = max(1.0, (np.abs(w1).sum(1) * h1b + np.abs(sd["final.linears.1.bias"].double().numpy())).max())
    assert sa2 * h2b <= 60000 * 1.001 and (sa2 == 16 or 2 * sa2 * h2b > 60000)
    # the split-pair copy of final.linears.1 carries sw1 and reproduces the weights to 22 bits, 300 included
    n_split = 1536 * 768 + 1536 * 1536 + 1536                                          # f0_s | f1_s | folded bias of final.linears.0
    f1s = raw[-n_split - 16 - n_l16 + 1536 * 768: -16 - n_l16 - 1536]
    f1s = f1s.reshape(1536 // 32, 1536, 32).transpose(1, 0, 2).reshape(-1).copy()       # k-tile blocked -> rows
    halves = f1s.view(np.float16).reshape(1536, 1536 // 8, 2, 8).astype(np.float64)
    assert np.isfinite(halves).all()
    rec = (halves[:, :, 0, :] + halves[:, :, 1, :]).reshape(1536, 1536) / sw1
    assert np.abs(rec - w1).max() <= 2.0 ** -21 * np.abs(w1).max() and abs(rec[5, 7] - 300.0) < 1e-4
    # layer 2: the FFN hidden's scale follows the bound max_c (8 ||W1'_c||_2 + |b1'_c|), the W1 fragments their own scale
    sc = raw[layer0 + 2 * stride + stride - 8 - 576: layer0 + 3 * stride - 576].astype(np.float64)
    W1 = sd["graph_encoder.encoder_f.layers.2.feed_forward.feed_forward.w_1.weight"].double().numpy()
    b1 = sd["graph_encoder.encoder_f.layers.2.feed_forward.feed_forward.w_1.bias"].double().numpy()
    fb = (8 * np.sqrt((W1 ** 2).sum(1)) + np.abs(b1)).max()                               # gamma = 1, beta = 0
    sa_f2 = sc[4]
    assert sa_f2 < 16 and sa_f2 * fb <= 60000 * 1.001 < 2 * sa_f2 * fb * 1.001
    sw_1 = 1 / (16 * sc[3])
    assert sw_1 * np.abs(W1).max() <= 60000 < 2 * sw_1 * np.abs(W1).max()
    l16 = raw[-n_l16:]
    assert np.isfinite(l16[2 * stride: 3 * stride - 8 - 576].view(np.float16).astype(np.float64)).all()
    # non-finite weights are refused
    sd["final.linears.0.weight"][0, 0] = float("nan")
    model.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="not finite"):
        model._pack_host(model._state_flat().contiguous())


def test_predictor_pack_transposes():
    d = PredictorDims()
    clf = mutils.create_model(predictor_config(d, "cpu"))
    sdp = synthetic.predictor_state_dict(d, seed=1)
    assert list(clf.state_dict().keys()) == list(sdp.keys())
    clf.load_state_dict(sdp)
    packed = clf._pack_host(clf._state_flat().contiguous()).numpy()
    g = 144
    w1_0 = packed[: 7 * g].reshape(7, g)
    assert np.array_equal(w1_0, sdp["gcn.0.weight1"].numpy())
    off = 2 * 7 * g                                   # w1_0, w2_0 plain; then per layer w1b, w2b, (w1f, w2f)
    assert np.array_equal(unfrag(packed[off:], 7, g), sdp["gcn.0.weight1"].numpy())       # layer 0 backward tile
    off += 2 * 32 * g                                 # layer 0: two one-tile matrices
    w1b_1 = unfrag(packed[off:], g, g)
    w1f_1 = unfrag(packed[off + 2 * 160 * g:], g, g)
    assert np.array_equal(w1b_1, sdp["gcn.1.weight1"].numpy())
    assert np.array_equal(w1f_1, sdp["gcn.1.weight1"].numpy().T)
    # set encoder (PyTorch, once per call) equals the oracle's restatement
    task = synthetic.synthetic_task(10, 20, seed=2)
    with torch.no_grad():
        assert torch.allclose(clf.set_encode(task)[0], O.set_encode(sdp, task), atol=1e-6)


def test_predictor_cate_pack_layout():
    """MetaPredictorCATE: state_dict names of the reference, forward weights, zero padding and the
    transposed operands of the backward pass (layout restated from csrc/dnag_common.h)."""
    d = PredictorCateDims()
    clf = mutils.create_model(predictor_cate_config(d, "cpu"))
    sdp = synthetic.predictor_cate_state_dict(d, seed=5)
    assert list(clf.state_dict().keys()) == list(sdp.keys())
    assert n_params(predictor_cate_param_spec(d)) == 774866
    assert _cate.MetaPredictorCATE is _pcm.MetaPredictorCATE
    clf.load_state_dict(sdp)
    lib = _lib.load()
    dc = clf.dims_c()
    packed = clf._pack_host(clf._state_flat().contiguous()).numpy()
    assert packed.size == lib.dnag_predcate_packed_floats(C.byref(dc))
    D_, V, dff, L, F, Hd, R, RK, R2, R2P, RNP, hs, nz = 64, 9, 32, 2, 128, 256, 180, 192, 360, 384, 256, 512, 56
    al4 = lambda x: (x + 3) // 4 * 4
    layer0 = al4(D_ * V) + D_ + 4 * D_ * D_ + 4 * D_ + 4 * D_ * D_ + D_
    stride = 3 * D_ * D_ + 3 * D_ + D_ * D_ + D_ + dff * D_ + dff + D_ * dff + D_ + 2 * D_ + 8 + (7 * D_ + dff)    # + operand scales + vector copy
    f0 = layer0 + L * stride
    assert np.array_equal(packed[f0: f0 + Hd * F], sdp["final.linears.0.weight"].numpy().ravel())
    tab_end = f0 + Hd * F + Hd + Hd * Hd + Hd + al4(V * Hd) + al4(V) + al4(20 * D_) + al4(D_ // 2)
    assert not packed[tab_end - al4(D_ // 2) - al4(20 * D_): tab_end - al4(D_ // 2)].any()      # no positional table
    enc_total = tab_end + Hd * F + Hd * Hd + Hd + 16 + L * stride   # split-pair copies + folded bias, head scales (unused by this surrogate), fp16 fragments
    # backward operands of layer 1: Wo^T and W2^T, fragment-major
    per_layer = 32 * 2 * 3 * D_ + 32 * 2 * D_ + 32 * 2 * dff + 32 * D_      # frag_floats(rows, K) = ceil(rows/32)*32*K
    b1 = enc_total + per_layer
    p = "graph_encoder.encoder_f.layers.1"
    woT = unfrag(packed[b1 + 64 * 3 * D_:], D_, D_)
    assert np.array_equal(woT, sdp[p + ".self_attn.attn.linears.3.weight"].numpy().T)
    w2T = unfrag(packed[b1 + 64 * 3 * D_ + 64 * D_ + 64 * dff:], dff, D_)
    assert np.array_equal(w2T, sdp[p + ".feed_forward.feed_forward.w_2.weight"].numpy().T)
    # Wqkv'^T: gamma of self_attn.norm folded in
    wqkvT = unfrag(packed[b1:], D_, 3 * D_)
    gam = sdp[p + ".self_attn.norm.weight"].numpy()
    Wk = sdp[p + ".self_attn.attn.linears.1.weight"].numpy()
    assert np.allclose(wqkvT[:, D_:2 * D_], (Wk * gam[None, :]).T, atol=0)
    o = enc_total + L * per_layer
    f0T = packed[o: o + F * Hd].reshape(F, Hd); o += F * Hd
    assert np.array_equal(f0T, sdp["final.linears.0.weight"].numpy().T)
    o += Hd * Hd
    r0 = packed[o: o + R2P * RK].reshape(R2P, RK); o += R2P * RK
    assert np.array_equal(r0[:R2, :R], sdp["regeress.linears.0.weight"].numpy()) and not r0[R2:].any() and not r0[:, R:].any()
    r0b = packed[o: o + R2P]; o += R2P
    assert np.array_equal(r0b[:R2], sdp["regeress.linears.0.bias"].numpy()) and not r0b[R2:].any()
    r0T = packed[o: o + RNP * R2P].reshape(RNP, R2P); o += RNP * R2P
    assert np.array_equal(r0T[:R, :R2], sdp["regeress.linears.0.weight"].numpy().T) and not r0T[R:].any()
    o += R2P * R2P + R2P + R2P * R2P
    wd = packed[o: o + hs * nz].reshape(hs, nz); o += hs * nz
    wa = packed[o: o + hs * R2P].reshape(hs, R2P); o += hs * R2P
    W0 = sdp["pred_fc.0.weight"].numpy()
    assert np.array_equal(wd, W0[:, :nz]) and np.array_equal(wa[:, :R2], W0[:, nz:]) and not wa[:, R2:].any()
    o += hs
    p0T = packed[o: o + R2P * hs].reshape(R2P, hs); o += R2P * hs
    assert np.array_equal(p0T[:R2], W0[:, nz:].T)
    assert np.array_equal(packed[o: o + hs], sdp["pred_fc.2.weight"].numpy().ravel())
    assert packed[o + hs] == sdp["pred_fc.2.bias"].numpy()[0] and o + hs + 4 == packed.size
    bad = _lib.PredCateDimsC(20, 9, 64, 2, 32, 2, 500, 56)
    buf = torch.zeros(16)
    assert lib.dnag_predcate_pack(C.byref(bad), buf.data_ptr(), buf.data_ptr(), buf.data_ptr()) == -1
    assert b"unsupported hs" in lib.dnag_last_error()


def test_schedule_table_matches_oracle_vpsde():
    for beta_max, N in ((20.0, 1000), (5.0, 100)):
        sde = sde_lib.VPSDE(0.1, beta_max, N)
        osde = O.VPSDE(0.1, beta_max, N)
        tab = sde.schedule_table(1e-3)
        ts = O.timesteps(osde, 1e-3)
        k = osde.timestep_index(ts)
        assert tab.shape == (N, _lib.SCHED_COLS) and tab.dtype == np.float32
        assert np.array_equal(tab[:, _lib.SCHED_T], ts.numpy())
        assert np.array_equal(tab[:, _lib.SCHED_LABEL], (ts * 999).numpy())
        assert np.array_equal(tab[:, _lib.SCHED_STD], osde.marginal_std(ts).numpy())
        assert np.array_equal(tab[:, _lib.SCHED_ALPHA], osde.alphas[k].numpy())
        _, G = osde.discretize(torch.zeros(N, 1, 1), ts)
        assert np.array_equal(tab[:, _lib.SCHED_G], G.numpy())
        assert np.array_equal(tab[:, _lib.SCHED_G2], (G ** 2).numpy())
        assert int(k[0]) == N - 1 and int(k[-1]) == 0
    assert sde_lib.VPSDE(0.1, 20, 1000).schedule_table(1e-3, 8).shape == (8, 8)
    with pytest.raises(ValueError):
        sde_lib.VPSDE(0.1, 20, 10).schedule_table(1e-3, 11)


def test_schedule_table_vesde_and_euler_maruyama_match_oracle():
    """VESDE (the reference's shipped default SDE) and the Euler-Maruyama scalars, against the oracle's
    restatement of sde_lib.py:253-300, :168-177 and sampling.py:194-200."""
    N = 1000
    sde, osde = sde_lib.VESDE(0.1, 5.0, N), O.VESDE(0.1, 5.0, N)
    ts = O.timesteps(osde, 1e-5)
    tab = sde.schedule_table(1e-5)
    assert np.array_equal(tab[:, _lib.SCHED_LABEL], osde.marginal_std(ts).numpy())
    assert np.all(tab[:, _lib.SCHED_ALPHA] == 1.0) and np.all(tab[:, _lib.SCHED_SQRTA] == 1.0)
    _, G = osde.discretize(torch.zeros(N, 1, 1), ts)
    assert np.array_equal(tab[:, _lib.SCHED_G], G.numpy()) and np.array_equal(tab[:, _lib.SCHED_G2], (G ** 2).numpy())
    assert tab[-1, _lib.SCHED_G] == osde.discrete_sigmas[0]                       # k = 0: adjacent sigma is 0
    assert sde.score_mode == _lib.SCORE_VE and sde.prior_scale == 5.0
    for s_, o_ in ((sde, osde), (sde_lib.VPSDE(0.1, 20.0, N), O.VPSDE(0.1, 20.0, N))):
        eps = 1e-5 if s_ is sde else 1e-3
        tab = s_.schedule_table(eps, predictor="euler_maruyama")
        ts = O.timesteps(o_, eps)
        x = torch.ones(N, 1, 1)
        drift, diffusion = o_.sde(x, ts)
        dt = -1.0 / N
        assert np.array_equal(tab[:, _lib.SCHED_SQRTA], drift.reshape(-1).numpy())
        assert np.array_equal(tab[:, _lib.SCHED_G2], (diffusion ** 2).numpy())
        assert np.array_equal(tab[:, _lib.SCHED_G], (diffusion[:, None, None] * np.sqrt(-dt)).reshape(-1).numpy())
        assert np.all(tab[:, _lib.SCHED_DT] == np.float32(dt))
    torch.manual_seed(0)
    a = sde.prior_sampling((2, 8, 7))
    torch.manual_seed(0)
    assert torch.equal(a, torch.randn(2, 8, 7) * 5.0)


def test_registries_and_factory_errors_mirror_the_reference():
    assert sampling.get_predictor("reverse_diffusion") is sampling.ReverseDiffusionPredictor
    assert sampling.get_corrector("langevin") is sampling.LangevinCorrector
    assert set(sampling._PREDICTORS) == {"euler_maruyama", "reverse_diffusion", "none"}
    assert set(sampling._CORRECTORS) == {"langevin", "none"}
    with pytest.raises(ValueError, match="Already registered"):
        sampling.register_predictor(name="none")(object)
    with pytest.raises(ValueError, match="Already registered model"):
        mutils.register_model(name="CATE")(object)
    assert mutils.get_model("CATE") is _cate.CATE and mutils.get_model("MetaNeuralPredictor") is _dm.MetaNeuralPredictor
    cfg = score_config(CateDims(), "cpu")
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    inv = utils.get_data_inverse_scaler(cfg)
    cfg.sampling["method"] = "ode"
    with pytest.raises(NotImplementedError, match="Sampler name ode unknown"):
        sampling.get_sampling_fn(cfg, sde, (4, 8, 7), inv, 1e-3)
    cfg.sampling["method"] = "pc"
    for pname in ("euler_maruyama", "none", "reverse_diffusion"):        # every registered predictor is built
        cfg.sampling["predictor"] = pname
        for s_ in (sde, sde_lib.VESDE(0.1, 5.0, 1000)):
            assert callable(sampling.get_sampling_fn(cfg, s_, (4, 8, 7), inv, 1e-3))
    cfg.sampling["probability_flow"] = True
    with pytest.raises(NotImplementedError):
        sampling.get_sampling_fn(cfg, sde, (4, 8, 7), inv, 1e-3)
    cfg.sampling["probability_flow"] = False

    class SubVP(sde_lib.VPSDE):
        pass
    with pytest.raises(NotImplementedError, match="not yet supported"):
        sampling.get_sampling_fn(cfg, object(), (4, 8, 7), inv, 1e-3)
    fn = sampling.get_sampling_fn(cfg, sde, (4, 8, 7), inv, 1e-3)
    with pytest.raises(RuntimeError, match="MI355X only"):     # no CPU fallback
        fn(mutils.create_model(cfg), utils.nb201_adj().unsqueeze(0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mutils.create_model(cfg).eval()(torch.zeros(2, 8, 7), torch.zeros(2), torch.zeros(2, 8, 8))
    with pytest.raises(NotImplementedError):
        mutils.get_score_fn(sde, None, continuous=False)


def test_mask_and_scaler_helpers():
    adj = utils.nb201_adj()
    assert torch.equal(adj, O.nb201_adj())
    assert torch.equal(utils.aug_mask(adj, algo="floyd")[0], adj)          # the reference's typo path
    assert torch.equal(utils.aug_mask(adj)[0], O.floyed(adj))
    assert utils.aug_mask(adj.unsqueeze(0).repeat(3, 1, 1)).shape == (3, 8, 8)
    cfg = score_config(CateDims(), "cpu")
    x = torch.tensor([0.0, 0.25, 1.0])
    assert torch.equal(utils.get_data_inverse_scaler(cfg)(utils.get_data_scaler(cfg)(x)), x)


def test_shard_batch_partitions():
    for total, world in ((65536, 8), (4096, 3), (5, 8), (7, 2)):
        spans = [D.shard_batch(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DNAG_REPO"])
from diffusionnag_amd import distributed as D
rank, world, local = D.init_from_env("gloo")
total = 7
start, count = D.shard_batch(total, rank, world)
x = (torch.arange(start, start + count, dtype=torch.float32)[:, None, None] + torch.zeros(count, 8, 7))
full = D.gather_samples(x, total)
assert full.shape == (total, 8, 7), full.shape
assert torch.equal(full[:, 0, 0], torch.arange(total, dtype=torch.float32)), full[:, 0, 0]
sys.stdout.write(f"rank {rank} ok\n"); sys.stdout.flush()      # one write: two ranks share the pipe
"""


def test_gather_world_size_2_gloo(tmp_path):
    import socket
    with socket.socket() as sk:          # a free port: a fixed one can still be in TIME_WAIT from the previous run
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, DNAG_REPO=REPO, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


_WORKER_LANGEVIN = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["DNAG_REPO"])
from diffusionnag_amd import distributed as D
rank, world, local = D.init_from_env("gloo")
# per-block sums of ||grad_b|| and ||noise_b|| of a 64-block batch (8 graphs per block), as the exit kernel writes
# them: [2][blocks]; each rank holds its contiguous half
rs = np.random.RandomState(3)
gn, nn = rs.uniform(1, 1e4, 64).astype(np.float32), rs.uniform(50, 70, 64).astype(np.float32)
nb = 64 // world
local_p = torch.from_numpy(np.concatenate([gn[rank * nb:(rank + 1) * nb], nn[rank * nb:(rank + 1) * nb]]))
glob = D.gather_partials(local_p)                               # [world][2][nb], rank order
ex = D.LangevinExchange(2 * nb, torch.device("cpu"))            # the object dnag_pc_sample's callback drives
ex.local.copy_(local_p)
ex()
assert torch.equal(ex.global_, glob) and ex.calls == 1


def reduce_like_kernel(parts, nseg, batch):
    # corrector_update_kernel (csrc/dnag_head.hip.h): thread t sums blocks t, t + 256, ... in order, then a pairwise tree
    seg = parts.numel() // (2 * nseg)
    p = parts.view(nseg, 2, seg)
    out = []
    for which in (0, 1):
        flat = p[:, which, :].reshape(-1)                        # block k of the whole batch = segment k // seg, entry k % seg
        lanes = np.zeros(256, np.float32)
        for k in range(flat.numel()):
            lanes[k % 256] = np.float32(lanes[k % 256] + np.float32(flat[k]))
        s = 128
        while s > 0:
            lanes[:s] = (lanes[:s] + lanes[s:2 * s]).astype(np.float32)
            s >>= 1
        out.append(np.float32(lanes[0]) / np.float32(batch))
    return out


whole = torch.from_numpy(np.concatenate([gn, nn]))              # the unsharded batch's buffer: [1][2][64]
a = reduce_like_kernel(whole, 1, 512)
b = reduce_like_kernel(glob, world, 512)
assert a[0] == b[0] and a[1] == b[1], (a, b)                     # bit for bit
mine = reduce_like_kernel(local_p, 1, 512 // world)             # the per-shard mode differs (its own batch mean)
assert mine[0] != a[0]
sys.stdout.write(f"rank {rank} ok\n"); sys.stdout.flush()
"""


def test_langevin_exchange_world_size_2_gloo(tmp_path):
    """Exact-global-batch Langevin mode: two shards exchange their per-block norm sums and reduce the
    gathered list in the kernel's fixed order -- the same bits as the unsharded batch.  (The kernels' side of it is
    tests/test_gpu_shard.py::test_two_shards_reproduce_the_single_batch_step_size.)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker_langevin.py"
    script.write_text(_WORKER_LANGEVIN)
    env = dict(os.environ, DNAG_REPO=REPO, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


# ----------------------------------------------------------------------------------------------------
# DSM training step: host logic (the kernels are covered by tests/test_gpu_train.py)
# ----------------------------------------------------------------------------------------------------
def test_train_args_struct_matches_header():
    header = open(os.path.join(REPO, "include", "dnag.h")).read()
    body = header[header.index("typedef struct dnag_train_args {"):header.index("} dnag_train_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = body[body.index("{") + 1:]
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            for part in decl.split(","):
                names.append(re.sub(r"[\*\s]", " ", part).split()[-1])
    assert names == [f[0] for f in _lib.TrainArgsC._fields_]


def test_dropout_mask_packing_layout():
    from diffusionnag_amd import losses
    lib = _lib.load()
    dims = CateDims()
    B = 3
    d = _lib.CateDimsC(8, 7, 64, 8, 128, 12, 0)
    lay = losses.dropout_mask_layout(dims, B)
    assert lay["total"] == lib.dnag_dropout_mask_bytes(C.byref(d), B)
    # every site element gets a distinct tag -> find it where include/dnag.h says it lives
    sites = O.dropout_sites(B, 8)
    rs = np.random.RandomState(0)
    masks = {k: torch.from_numpy(rs.randint(0, 2, size=s).astype(np.float32)) for k, s in sites.items()}
    flat = losses.pack_dropout_masks(masks, dims, B).numpy()
    rows, pl = lay["rows"], lay["per_layer"]
    b, n, l = 2, 5, 7
    row = b * 8 + n
    assert (flat[row * 64:(row + 1) * 64] == masks["emb"][b, n].numpy()).all()
    base = lay["layer0"] + l * rows * pl + row * pl
    for h in range(8):
        assert (flat[base + h * 8: base + h * 8 + 8] == masks[f"attn{l}"][b, h, n].numpy()).all()
    assert (flat[base + lay["sa"]: base + lay["sa"] + 64] == masks[f"sa{l}"][b, n].numpy()).all()
    assert (flat[base + lay["ffh"]: base + lay["ffh"] + 128] == masks[f"ffh{l}"][b, n].numpy()).all()
    assert (flat[base + lay["ffo"]: base + lay["ffo"] + 64] == masks[f"ffo{l}"][b, n].numpy()).all()


def test_training_step_has_no_cpu_fallback_and_rejects_bad_arguments():
    from diffusionnag_amd import losses
    lib = _lib.load()
    assert lib.dnag_cate_dsm_step(C.byref(_lib.TrainArgsC())) == -1          # NULL dims rejected before any launch
    ofa = _lib.CateDimsC(20, 9, 64, 8, 128, 12, 1)
    a = _lib.TrainArgsC()
    a.cate_dims = C.pointer(ofa)
    assert lib.dnag_cate_dsm_step(C.byref(a)) == -1 and b"NAS-Bench-201" in lib.dnag_last_error()
    d = _lib.CateDimsC(8, 7, 64, 8, 128, 12, 0)
    assert lib.dnag_train_workspace_bytes(C.byref(d), 1024) > 1024 * 8 * (768 + 4 * 1536) * 4
    with pytest.raises(NotImplementedError):
        losses.get_sde_loss_fn_nas(sde_lib.VPSDE(), train=True, likelihood_weighting=True)
    model = mutils.create_model(score_config(CateDims(), "cpu"))
    fn = losses.get_sde_loss_fn_nas(sde_lib.VPSDE(), train=True, reduce_mean=True, continuous=True, likelihood_weighting=False)
    x = torch.zeros(2, 8, 7)
    m = torch.zeros(2, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fn(model, (x, m, m))


def test_ema_mirror_matches_reference_formula():
    from diffusionnag_amd.ema import ExponentialMovingAverage
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    ema = ExponentialMovingAverage(ps, decay=0.9999)
    shadow = [p.detach().clone() for p in ps]
    for step in range(1, 4):
        with torch.no_grad():
            for p in ps:
                p.add_(torch.randn_like(p))
        ema.update(ps)
        decay = min(0.9999, (1 + step) / (10 + step))                 # models/ema.py:37
        shadow = [s - (1.0 - decay) * (s - p.detach()) for s, p in zip(shadow, ps)]
    for s, e in zip(shadow, ema.shadow_params):
        assert torch.equal(s, e)
    ema.store(ps)
    ema.copy_to(ps)
    assert all(torch.equal(p.data, s) for p, s in zip(ps, ema.shadow_params))
    ema.restore(ps)
    assert not torch.equal(ps[0].data, ema.shadow_params[0])
    assert set(ema.state_dict()) == {"decay", "num_updates", "shadow_params"}


def test_checkpoint_round_trip_and_safe_loader(tmp_path):
    """save_checkpoint / restore_checkpoint / load_score_model mirrors (reference utils.py:84-121,
    main_exp/diffusion/run_lib.py:61-79) on a checkpoint written here; a file holding an arbitrary object is refused."""
    from diffusionnag_amd.ema import ExponentialMovingAverage
    cfg = score_config(CateDims(), "cpu")
    model = mutils.create_model(cfg)
    model.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=0))
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01)
    ema.update(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    state = dict(optimizer=opt, model=model, ema=ema, step=7)
    ckdir = str(tmp_path / "ck")
    utils.save_checkpoint(ckdir, state, 7, 1, is_best=True)
    path = os.path.join(ckdir, "model_best.pth.tar")
    assert os.path.exists(path) and os.path.exists(os.path.join(ckdir, "checkpoint_7_1.pth.tar"))
    model2 = mutils.create_model(cfg)
    ema2 = ExponentialMovingAverage(model2.parameters(), decay=0.5)
    state2 = utils.restore_checkpoint(path, dict(optimizer=torch.optim.Adam(model2.parameters(), lr=1e-3), model=model2, ema=ema2, step=0),
                                      "cpu", resume=True)
    assert state2["step"] == 7 and ema2.decay == 0.999 and ema2.num_updates == 1
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
    m3, e3 = utils.load_score_model(path, cfg)
    assert all(torch.equal(p, s) for p, s in zip(m3.parameters(), ema.shadow_params))      # the EMA weights are what samples
    bad = str(tmp_path / "bad.pth.tar")
    torch.save({"model": model.state_dict(), "config": Cfg_like()}, bad)
    with pytest.raises(RuntimeError, match="safe loader"):
        utils.restore_checkpoint(bad, dict(model=model2), "cpu", resume=True)


class Cfg_like:
    """Stands for the reference's ConfigDict inside a checkpoint: an arbitrary class the safe loader must refuse."""
    def __init__(self):
        self.x = 1
