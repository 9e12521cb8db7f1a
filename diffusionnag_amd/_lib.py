"""ctypes binding of ``libdnag.so`` (C ABI declared in ``include/dnag.h``).

There is no CPU fallback: if the library has not been built (``python __graft_entry__.py`` or
``make -C diffusionnag_amd/csrc``) every product call raises.  PyTorch is used only for device
memory and streams; every pointer handed to the library comes from ``Tensor.data_ptr()``.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdnag.so")

SCHED_COLS = 8
SCHED_LABEL, SCHED_STD, SCHED_ALPHA, SCHED_SQRTA, SCHED_G, SCHED_G2, SCHED_T, SCHED_DT = range(8)
PREDICTOR_NONE, PREDICTOR_REVERSE_DIFFUSION, PREDICTOR_EULER_MARUYAMA = range(3)
SCORE_VP, SCORE_VE = range(2)

# every symbol include/dnag.h declares (tests check that the built library exports them all)
EXPORTS = (
    "dnag_abi_version", "dnag_last_error", "dnag_pc_args_bytes", "dnag_train_args_bytes", "dnag_ofa_validity",
    "dnag_cate_state_floats", "dnag_cate_table_floats", "dnag_cate_packed_floats", "dnag_cate_pack",
    "dnag_pred_state_floats", "dnag_pred_table_floats", "dnag_pred_packed_floats", "dnag_pred_pack",
    "dnag_predcate_state_floats", "dnag_predcate_table_floats", "dnag_predcate_packed_floats", "dnag_predcate_pack",
    "dnag_predcate_workspace_bytes", "dnag_workspace_bytes", "dnag_workspace_bytes_predcate",
    "dnag_cate_forward", "dnag_pred_forward_grad", "dnag_predcate_forward_grad", "dnag_pc_sample",
    "dnag_quantize", "dnag_timing_begin", "dnag_timing_read", "dnag_timing_end",
    "dnag_nb201_seen_reset", "dnag_nb201_validity",
    "dnag_train_workspace_bytes", "dnag_dropout_mask_bytes", "dnag_dropout_fill", "dnag_cate_dsm_step",
    "dnag_pred_forward_grad_noalloc", "dnag_optim_workspace_bytes", "dnag_optim_args_bytes", "dnag_optim_step",
    "dnag_langevin_partials_floats", "dnag_pred_workspace_bytes",
)


class CateDimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("max_node", "n_vocab", "d_model", "n_head", "d_ff", "n_layers", "stagewise_pos")]


class PredDimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("max_node", "n_vocab", "gcn_hidden", "gcn_layers", "linear_hidden", "hs", "nz")]


class PredCateDimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("max_node", "n_vocab", "d_model", "n_head", "d_ff", "n_layers", "hs", "nz")]


_fp = C.c_void_p   # float* (device or host address)


class PcArgsC(C.Structure):
    _fields_ = [
        ("cate_dims", C.POINTER(CateDimsC)), ("cate_packed", _fp),
        ("pred_dims", C.POINTER(PredDimsC)), ("pred_packed", _fp),
        ("predcate_dims", C.POINTER(PredCateDimsC)), ("predcate_packed", _fp), ("d_mu", _fp),
        ("classifier_scale", C.c_float), ("label_sign", C.c_float),
        ("batch", C.c_int32), ("mask", _fp), ("sched", _fp), ("n_steps", C.c_int32),
        ("snr", C.c_float), ("corrector", C.c_int32), ("predictor", C.c_int32), ("score_mode", C.c_int32),
        ("prior_scale", C.c_float),
        ("x0", _fp), ("noise_corr", _fp), ("noise_pred", _fp),
        ("seed", C.c_uint64), ("stream_offset", C.c_uint64),
        ("x_out", _fp), ("x_last", _fp), ("x_mean_last", _fp),
        ("rec_score_c", _fp), ("rec_total_c", _fp), ("rec_x_corr", _fp), ("rec_score_p", _fp),
        ("rec_total_p", _fp), ("rec_x_mean", _fp), ("rec_x_pred", _fp),
        ("rec_cgrad_c", _fp), ("rec_cgrad_p", _fp), ("rec_step_size", _fp),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("cscale_steps", _fp), ("guide_steps", C.c_void_p), ("weight_ratio", C.c_int32), ("weight_ratio_p", C.c_float),
        ("prod_w", C.c_int32),
        ("chain_x", _fp), ("chain_steps", C.c_int32), ("chain_keep", C.c_int32), ("chain_n", C.c_int32),
        ("chain_denoise", C.c_int32),
        ("corrector_steps", C.c_int32),
        ("langevin_exchange", C.c_void_p), ("langevin_user", C.c_void_p), ("langevin_local", _fp), ("langevin_global", _fp),
        ("langevin_world", C.c_int32),
    ]


LANGEVIN_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class TrainArgsC(C.Structure):
    _fields_ = [
        ("cate_dims", C.POINTER(CateDimsC)), ("params", _fp), ("tables", _fp), ("batch", C.c_int32),
        ("x", _fp), ("z", _fp), ("mean_coef", _fp), ("std", _fp), ("labels", _fp), ("mask", _fp),
        ("mask_stride", C.c_int32), ("score_mode", C.c_int32), ("reduce_mean", C.c_int32),
        ("keep", C.c_void_p), ("dropout_p", C.c_float), ("loss", _fp), ("grads", _fp),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
    ]


class OptimArgsC(C.Structure):
    _fields_ = [
        ("n", C.c_size_t), ("flags", C.c_int32), ("params", _fp), ("grads", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp),
        ("shadow", _fp), ("beta1", C.c_float), ("beta2", C.c_float), ("one_minus_beta1", C.c_float),
        ("one_minus_beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("step_size", C.c_float),
        ("bc2_sqrt", C.c_float), ("grad_clip", C.c_float), ("ema_one_minus_decay", C.c_float), ("grad_norm_out", _fp),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
    ]


OPTIM_ADAM, OPTIM_EMA = 1, 2

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the library (once).  Raises with build instructions when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is not built.  Run `python __graft_entry__.py` "
            "(or `make -C diffusionnag_amd/csrc`) first; diffusionnag_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    sz, i32, vp = C.c_size_t, C.c_int, C.c_void_p
    lib.dnag_abi_version.restype = i32
    lib.dnag_last_error.restype = C.c_char_p
    for n in ("dnag_cate_state_floats", "dnag_cate_table_floats", "dnag_cate_packed_floats"):
        getattr(lib, n).restype = sz
        getattr(lib, n).argtypes = [C.POINTER(CateDimsC)]
    for n in ("dnag_pred_state_floats", "dnag_pred_table_floats", "dnag_pred_packed_floats"):
        getattr(lib, n).restype = sz
        getattr(lib, n).argtypes = [C.POINTER(PredDimsC)]
    for n in ("dnag_predcate_state_floats", "dnag_predcate_table_floats", "dnag_predcate_packed_floats"):
        getattr(lib, n).restype = sz
        getattr(lib, n).argtypes = [C.POINTER(PredCateDimsC)]
    lib.dnag_predcate_pack.restype = i32
    lib.dnag_predcate_pack.argtypes = [C.POINTER(PredCateDimsC), vp, vp, vp]
    lib.dnag_predcate_workspace_bytes.restype = sz
    lib.dnag_predcate_workspace_bytes.argtypes = [C.POINTER(PredCateDimsC), i32]
    lib.dnag_workspace_bytes_predcate.restype = sz
    lib.dnag_workspace_bytes_predcate.argtypes = [C.POINTER(CateDimsC), C.POINTER(PredCateDimsC), i32, i32]
    lib.dnag_predcate_forward_grad.restype = i32
    lib.dnag_predcate_forward_grad.argtypes = [C.POINTER(PredCateDimsC), vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, sz, vp]
    lib.dnag_cate_pack.restype = i32
    lib.dnag_cate_pack.argtypes = [C.POINTER(CateDimsC), vp, vp, vp]
    lib.dnag_pred_pack.restype = i32
    lib.dnag_pred_pack.argtypes = [C.POINTER(PredDimsC), vp, vp, vp]
    lib.dnag_workspace_bytes.restype = sz
    lib.dnag_workspace_bytes.argtypes = [C.POINTER(CateDimsC), C.POINTER(PredDimsC), i32, i32]
    lib.dnag_cate_forward.restype = i32
    lib.dnag_cate_forward.argtypes = [C.POINTER(CateDimsC), vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, sz, vp]
    lib.dnag_pred_forward_grad.restype = i32
    lib.dnag_pred_forward_grad.argtypes = [C.POINTER(PredDimsC), vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, sz, vp]
    lib.dnag_pred_forward_grad_noalloc.restype = i32
    lib.dnag_pred_forward_grad_noalloc.argtypes = [C.POINTER(PredDimsC), vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp]
    lib.dnag_pc_sample.restype = i32
    lib.dnag_pc_sample.argtypes = [C.POINTER(PcArgsC)]
    lib.dnag_pred_workspace_bytes.restype = sz
    lib.dnag_pred_workspace_bytes.argtypes = [C.POINTER(PredDimsC), i32]
    lib.dnag_langevin_partials_floats.restype = sz
    lib.dnag_langevin_partials_floats.argtypes = [C.POINTER(CateDimsC), i32]
    lib.dnag_quantize.restype = i32
    lib.dnag_quantize.argtypes = [vp, vp, sz, vp]
    lib.dnag_timing_begin.restype = i32
    lib.dnag_timing_begin.argtypes = [i32, i32]
    lib.dnag_timing_read.restype = i32
    lib.dnag_timing_read.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.dnag_timing_end.restype = i32
    lib.dnag_timing_end.argtypes = []
    lib.dnag_nb201_seen_reset.restype = i32
    lib.dnag_nb201_seen_reset.argtypes = [vp, vp]
    lib.dnag_nb201_validity.restype = i32
    lib.dnag_nb201_validity.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp]
    lib.dnag_train_workspace_bytes.restype = sz
    lib.dnag_train_workspace_bytes.argtypes = [C.POINTER(CateDimsC), i32]
    lib.dnag_dropout_mask_bytes.restype = sz
    lib.dnag_dropout_mask_bytes.argtypes = [C.POINTER(CateDimsC), i32]
    lib.dnag_dropout_fill.restype = i32
    lib.dnag_dropout_fill.argtypes = [C.POINTER(CateDimsC), i32, C.c_float, C.c_uint64, vp, vp]
    lib.dnag_cate_dsm_step.restype = i32
    lib.dnag_cate_dsm_step.argtypes = [C.POINTER(TrainArgsC)]
    if lib.dnag_abi_version() != 1:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.dnag_abi_version()} != 1; rebuild the library")
    lib.dnag_ofa_validity.restype = i32
    lib.dnag_ofa_validity.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.dnag_optim_workspace_bytes.restype = sz
    lib.dnag_optim_args_bytes.restype = sz
    lib.dnag_optim_step.restype = i32
    lib.dnag_optim_step.argtypes = [C.POINTER(OptimArgsC)]
    if lib.dnag_optim_args_bytes() != C.sizeof(OptimArgsC):
        raise ImportError(f"{LIB_PATH}: dnag_optim_args is {lib.dnag_optim_args_bytes()} bytes, this binding has "
                          f"{C.sizeof(OptimArgsC)}; rebuild the library")
    lib.dnag_pc_args_bytes.restype = sz
    lib.dnag_train_args_bytes.restype = sz
    if lib.dnag_pc_args_bytes() != C.sizeof(PcArgsC) or lib.dnag_train_args_bytes() != C.sizeof(TrainArgsC):
        raise ImportError(f"{LIB_PATH}: argument structs ({lib.dnag_pc_args_bytes()}, {lib.dnag_train_args_bytes()} bytes) do not "
                          f"match this binding ({C.sizeof(PcArgsC)}, {C.sizeof(TrainArgsC)}); rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    """Turn a negative return code into RuntimeError carrying dnag_last_error() (no exceptions cross the ABI)."""
    if rc != 0:
        msg = load().dnag_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Address of a contiguous fp32 tensor, or NULL."""
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("dnag buffers must be contiguous float32 tensors")
    return t.data_ptr()


def require_device(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on one HIP device; the product path has no CPU implementation."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("diffusionnag_amd runs on an MI355X only: got a CPU tensor (no CPU fallback exists)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
    return dev


def current_stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream
