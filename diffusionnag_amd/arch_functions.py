"""Host mirror of the reference's ``analysis/arch_functions.py`` for the exit of a sampling round
(reference ``analysis/arch_functions.py:85-125``, ``:209-329``): validity, architecture strings, uniqueness.

The per-sample work -- threshold, validity rules, argmax decode -- runs in ``dnag_nb201_validity`` on the
device for the whole batch; the host only formats strings from the returned integer codes.
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

OPS_DECODER = ['input', 'output', 'none', 'skip_connect', 'nor_conv_1x1', 'nor_conv_3x3', 'avg_pool_3x3']   # :88
NCODES = 15625
UNSEEN = 0x7fffffff
ERORR_NB201 = {'MULTIPLE_NODE_TYPES': 1, 'No_START_END': 2, 'INTERM_START_END': 3, 'NO_ERROR': -1}        # :308-313


def code_to_string(code: int, ops: Sequence[str] = OPS_DECODER) -> str:
    """``decode_x_to_NAS_BENCH_201_string`` (``:240-256``) from the kernel's integer code."""
    t = [ops[(code // 5 ** i) % 5 + 2] for i in range(6)]
    return '|{}~0|+|{}~0|{}~1|+|{}~0|{}~1|{}~2|'.format(*t)


def device_validity(x: torch.Tensor, first_seen: Optional[torch.Tensor] = None, index_base: int = 0):
    """``x`` [B, 8, 7] on the device (sampler output or already quantised).  Returns device int32 tensors
    ``status [B]``, ``code [B]``, ``counts [2]`` (valid in this batch, distinct codes in ``first_seen``)."""
    lib = _lib.load()
    dev = _lib.require_device(x, first_seen)
    if x.dim() != 3 or tuple(x.shape[1:]) != (8, 7):
        raise ValueError(f"expected [B, 8, 7], got {tuple(x.shape)}")
    xc = x.detach().to(torch.float32).contiguous()
    B = xc.shape[0]
    status = torch.empty(B, dtype=torch.int32, device=dev)
    code = torch.empty(B, dtype=torch.int32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    if first_seen is not None and (first_seen.dtype != torch.int32 or first_seen.numel() != NCODES or not first_seen.is_contiguous()):
        raise ValueError("first_seen must be a contiguous int32 tensor of NCODES entries")
    with torch.cuda.device(dev):
        rc = lib.dnag_nb201_validity(xc.data_ptr(), B, status.data_ptr(), code.data_ptr(),
                                     first_seen.data_ptr() if first_seen is not None else None, int(index_base),
                                     counts.data_ptr(), _lib.current_stream(dev))
    _lib.check(rc, "dnag_nb201_validity")
    return status, code, counts


def new_seen_table(device) -> torch.Tensor:
    """A first-occurrence table with every code unseen."""
    lib = _lib.load()
    dev = torch.device(device)
    t = torch.empty(NCODES, dtype=torch.int32, device=dev)
    _lib.require_device(t)
    with torch.cuda.device(dev):
        _lib.check(lib.dnag_nb201_seen_reset(t.data_ptr(), _lib.current_stream(dev)), "dnag_nb201_seen_reset")
    return t


class BasicArchMetricsMeta(object):
    """Reference ``analysis/arch_functions.py:85-125`` (constructed without a training set, as
    ``generate_archs_meta`` does, ``main_exp/diffusion/run_lib.py:115``)."""

    def __init__(self, train_ds=None, train_arch_str_list=None, device="cuda:0"):
        if train_ds is not None:
            raise NotImplementedError("novelty against the NAS-Bench-201 table needs the reference's data blobs")
        self.ops_decoder = list(OPS_DECODER)
        self.train_arch_str_list = train_arch_str_list
        self.device = torch.device(device)

    def compute_validity(self, generated):
        """``generated``: the sampler's device tensor [B, 8, 7] or the list ``quantize`` returns.  Returns
        ``(valid, validity, valid_arch_str, all_arch_str)`` like ``:95-118``."""
        if torch.is_tensor(generated):
            x = generated
        else:
            if len(generated) == 0:
                return [], 0, [], []
            x = torch.from_numpy(np.stack([np.asarray(g, np.float32) for g in generated])).to(self.device)
        status, code, _ = device_validity(x)
        self.last_error_types = status.cpu().numpy()
        codes = code.cpu().numpy()
        q = (x.detach() >= 0.5).to(torch.float32).cpu().numpy()
        valid, valid_arch_str, all_arch_str = [], [], []
        for i, c in enumerate(codes):
            if c >= 0:
                s = code_to_string(int(c), self.ops_decoder)
                valid.append(q[i])
                valid_arch_str.append(s)
                all_arch_str.append(s)
            else:
                all_arch_str.append(None)
        validity = 0 if len(codes) == 0 or len(valid) == 0 else len(valid) / len(codes)
        return valid, validity, valid_arch_str, all_arch_str

    def compute_uniqueness(self, valid_arch_str):
        return list(set(valid_arch_str)), len(set(valid_arch_str)) / len(valid_arch_str)


# ----------------------------------------------------------------------------------------------
# MobileNetV3 / OFA search space (reference MobileNetV3/analysis/arch_functions.py)
# ----------------------------------------------------------------------------------------------
OFA_NUM_STAGE, OFA_MAX_LAYER_PER_STAGE = 5, 4                                                              # :11-12
OFA_OPS2STR = {0: '3-3', 1: '3-4', 2: '3-6', 3: '5-3', 4: '5-4', 5: '5-6', 6: '7-3', 7: '7-4', 8: '7-6'}  # :20-24
ERORR_OFA = {'MULIPLE_NODES': 1, 'INVALID_1OR2_LAYERS': 2, 'INVALID_3AND4_LAYERS': 3, 'NO_ERROR': -1}      # :414-419


def device_validity_ofa(x: torch.Tensor):
    """``x`` [B, 20, 9] on the device (sampler output or already quantised).  Returns device tensors ``status [B]``
    int32 (``is_valid_OFA_x`` codes), ``types [B, 20]`` int8 (-1: skipped layer), ``counts [4]`` int32 (graphs with
    status -1, 1, 2, 3)."""
    lib = _lib.load()
    dev = _lib.require_device(x)
    if x.dim() != 3 or tuple(x.shape[1:]) != (20, 9):
        raise ValueError(f"expected [B, 20, 9], got {tuple(x.shape)}")
    xc = x.detach().to(torch.float32).contiguous()
    B = xc.shape[0]
    status = torch.empty(B, dtype=torch.int32, device=dev)
    types = torch.empty(B, 20, dtype=torch.int8, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dnag_ofa_validity(xc.data_ptr(), B, status.data_ptr(), types.data_ptr(), counts.data_ptr(),
                                   _lib.current_stream(dev))
    _lib.check(rc, "dnag_ofa_validity")
    return status, types, counts


def ofa_types_to_string(types: Sequence[int]) -> str:
    """``get_string_from_onehot_x`` (``:135-144``) from the kernel's per-slot types: ``<stage depth>-<kernel>-<expand>``
    per layer slot, ``0-0-0`` for a skipped one (valid graphs have at most one type per slot, so the reference's stage
    sum of ones is the number of used slots)."""
    t = np.asarray(types).reshape(OFA_NUM_STAGE, OFA_MAX_LAYER_PER_STAGE)
    ds = (t >= 0).sum(axis=1)
    return '_'.join('0-0-0' if v < 0 else f'{int(ds[i])}-{OFA_OPS2STR[int(v)]}' for i in range(OFA_NUM_STAGE) for v in t[i])


class BasicArchMetricsMetaOFA(object):
    """Reference ``MobileNetV3/analysis/arch_functions.py:125-216`` (constructed without the training-set blob, as the
    sampling loop does).  Validity runs on the device for the whole batch; strings, uniqueness and novelty are host
    work on the 20 bytes per graph the kernel returns."""

    def __init__(self, train_ds=None, train_arch_str_list=None, except_inout=False, data_root=None, device="cuda:0"):
        if data_root is not None:
            raise NotImplementedError("the training-set blob is a pickle of the reference's data pipeline (out of scope)")
        self.ofa = None
        self.train_arch_list = None
        self.device = torch.device(device)

    def get_string_from_onehot_x(self, x):
        x = np.asarray(x)
        types = np.where(x.sum(axis=1) == 0, -1, x.argmax(axis=1))
        return ofa_types_to_string(types)

    def compute_validity(self, generated, adj=None, mask=None):
        """Returns ``(valid, validity, valid_arch_str, all_arch_str, error_types)`` like ``:146-163``."""
        if torch.is_tensor(generated):
            x = generated
        else:
            if len(generated) == 0:
                return [], 0, [], [], []
            x = torch.from_numpy(np.stack([np.asarray(g, np.float32) for g in generated])).to(self.device)
        status, types, counts = device_validity_ofa(x)
        st = status.cpu().numpy()
        ty = types.cpu().numpy()
        self.last_counts = counts.cpu().numpy()
        q = (x.detach() >= 0.5).to(torch.int64).cpu()
        valid, valid_arch_str, all_arch_str, error_types = [], [], [], []
        for i, s in enumerate(st):
            if s == -1:
                a = ofa_types_to_string(ty[i])
                valid.append(q[i])
                valid_arch_str.append(a)
            else:
                a = None
                error_types.append(int(s))
            all_arch_str.append(a)
        validity = 0 if len(st) == 0 else len(valid) / len(st)
        return valid, validity, valid_arch_str, all_arch_str, error_types

    def compute_uniqueness(self, valid_arch):
        """``:165-170``: first occurrences, in order (tensor equality there; a hash of the bytes here)."""
        seen, unique = set(), []
        for x in valid_arch:
            k = x.numpy().tobytes()
            if k not in seen:
                seen.add(k)
                unique.append(x)
        return unique, len(unique) / len(valid_arch)

    def compute_novelty(self, unique):
        if self.train_arch_list is None:                              # :175-177
            print("Dataset arch_str is None, novelty computation skipped")
            return 1, 1
        raise NotImplementedError

    def evaluate(self, generated, adj=None, mask=None, check_dataname='imagenet1k'):
        """``:184-216``: ``([validity, uniqueness, novelty, error_1, error_2, error_3], unique, dict of dummy property
        lists, None)``."""
        valid_arch, validity, _, _, error_types = self.compute_validity(generated, adj, mask)
        n = len(generated)
        e = np.asarray(error_types)
        error_1, error_2, error_3 = (float((e == k).sum()) / n for k in (1, 2, 3))
        if validity > 0:
            unique, uniqueness = self.compute_uniqueness(valid_arch)
            novelty = -1.0
        else:
            novelty, uniqueness, unique = -1.0, 0.0, []
        return ([validity, uniqueness, novelty, error_1, error_2, error_3], unique,
                dict(test_acc_list=[0], flops_list=[0], params_list=[0], latency_list=[0]), None)
