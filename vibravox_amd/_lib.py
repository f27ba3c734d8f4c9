"""ctypes binding of ``libeben_hip.so`` (C ABI declared in ``include/eben_hip.h``).

There is deliberately NO fallback: if the shared library is missing, or a tensor handed to an
op is not a contiguous float32 tensor on a HIP device, the call raises.  The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeben_hip.so")


class EbenError(RuntimeError):
    pass


class EbenConv1dDesc(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32),
        ("c_in", c_int32),
        ("c_out", c_int32),
        ("l_in", c_int32),
        ("l_out", c_int32),
        ("ksize", c_int32),
        ("stride", c_int32),
        ("dilation", c_int32),
        ("groups", c_int32),
        ("pad_l", c_int32),
        ("pad_r", c_int32),
        ("pad_mode", c_int32),
        ("transposed", c_int32),
        ("in_slope", c_float),
        ("out_slope", c_float),
        ("math", c_int32),
    ]


class EbenPackJob(ctypes.Structure):
    _fields_ = [("desc", EbenConv1dDesc), ("v", c_void_p), ("scale", c_void_p), ("wp_fwd", c_void_p), ("wp_bwd", c_void_p)]


class EbenRuPackJob(ctypes.Structure):
    _fields_ = [("channels", c_int32), ("math", c_int32), ("which", c_int32), ("pad_", c_int32),
                ("v_dil", c_void_p), ("scale_dil", c_void_p), ("v_pw", c_void_p), ("scale_pw", c_void_p), ("wimg", c_void_p)]


class EbenWnScaleItem(ctypes.Structure):
    _fields_ = [("g", c_void_p), ("v", c_void_p), ("scale", c_void_p), ("norm", c_void_p), ("rows", c_int32), ("cols", c_int32)]


class EbenWnBwdItem(ctypes.Structure):
    _fields_ = [("slabs", c_void_p), ("g", c_void_p), ("v", c_void_p), ("norm", c_void_p), ("dg", c_void_p), ("dv", c_void_p),
                ("dbias", c_void_p), ("slab_stride", c_int64), ("nslab", c_int32), ("rows", c_int32), ("cols", c_int32),
                ("row_stride", c_int32), ("col_perm_k", c_int32), ("pad_", c_int32)]


class EbenBlHeadJob(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("v", c_void_p), ("scale", c_void_p), ("bias", c_void_p), ("y_hi", c_void_p), ("y_lo", c_void_p),
                ("c_in", c_int32), ("c_out", c_int32), ("l_in", c_int32), ("l_out", c_int32), ("ksize", c_int32), ("dilation", c_int32),
                ("pad", c_int32), ("reflect_pad", c_int32), ("out_slope", c_float), ("pad_", c_int32)]


class EbenAdamTensor(ctypes.Structure):
    _fields_ = [
        ("param", c_void_p),
        ("grad", c_void_p),
        ("exp_avg", c_void_p),
        ("exp_avg_sq", c_void_p),
        ("numel", c_int64),
    ]


class EbenCollateItem(ctypes.Structure):
    _fields_ = [("speech", c_void_p), ("airborne", c_void_p), ("noise", c_void_p), ("length", c_int64), ("noise_start", c_int64),
                ("shift", c_int64)]


_P = c_void_p
_D = POINTER(EbenConv1dDesc)

# name -> (restype, argtypes); mirrors include/eben_hip.h one to one
SIGNATURES = {
    "eben_last_error": (c_char_p, []),
    "eben_version": (c_int, []),
    "eben_device_info": (c_int, [ctypes.c_char_p, c_size_t]),
    "eben_wn_scale": (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    "eben_wn_bwd": (c_int, [_P, c_int, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "eben_wn_scale_multi": (c_int, [POINTER(EbenWnScaleItem), c_int, _P]),
    "eben_wn_bwd_multi": (c_int, [POINTER(EbenWnBwdItem), c_int, _P]),
    "eben_conv1d_packed_floats": (c_size_t, [_D, c_int]),
    "eben_conv1d_kernel_generation": (c_int, [_D, c_int]),
    "eben_conv1d_pack": (c_int, [_D, _P, _P, _P, _P, _P]),
    "eben_conv1d_pack_multi": (c_int, [POINTER(EbenPackJob), c_int, _P]),
    "eben_conv1d_fwd": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "eben_conv1d_fwd_res": (c_int, [_D, _P, _P, _P, _P, c_float, _P, _P]),
    "eben_conv1d_bwd_dx_res": (c_int, [_D, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "eben_conv1d_bwd_dx_workspace": (c_size_t, [_D]),
    "eben_conv1d_bwd_dw_workspace": (c_size_t, [_D, POINTER(c_int), POINTER(c_int)]),
    "eben_conv1d_bwd_dx": (c_int, [_D, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "eben_conv1d_bwd_dx_ex": (c_int, [_D, _P, _P, _P, c_int, _P, c_float, c_int, POINTER(c_int), _P, _P]),
    "eben_conv1d_bwd_dx_fm": (c_int, [_D, _P, _P, _P, c_int, _P, c_float, _P, c_float, c_int, POINTER(c_int), _P, _P]),
    "eben_conv1d_bwd_dw": (c_int, [_D, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "eben_bl_from_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "eben_bl_to_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "eben_bl_conv1d_fwd": (c_int, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "eben_bl_conv1d_bwd_dx": (c_int, [_D, _P, _P, _P, _P, c_float, c_int, POINTER(c_int), c_int, c_int, _P, c_float, _P, _P, _P]),
    "eben_bl_dx_pr_desc": (c_int, [_D, _D]),
    "eben_bl_dx_pr_weights": (c_int, [_D, _P, _P, _P, _P]),
    "eben_bl_conv1d_bwd_dx_pr": (c_int, [_D, _P, _P, _P, _P, c_float, c_int, POINTER(c_int), c_int, c_int, _P, c_float, _P, _P, _P]),
    "eben_bl_conv1d_bwd_dx_c": (c_int, [_D, _P, _P, _P, _P, _P, c_float, c_int, POINTER(c_int), c_int, c_int, _P, c_float, _P, _P, _P]),
    "eben_bl_conv1d_bwd_dx_pr_c": (c_int, [_D, _P, _P, _P, _P, _P, c_float, c_int, POINTER(c_int), c_int, c_int, _P, c_float, _P, _P, _P]),
    "eben_bl_conv1d_bwd_dw_workspace": (c_size_t, [_D, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "eben_bl_conv1d_bwd_dw": (c_int, [_D, _P, _P, c_int, _P, c_size_t, _P]),
    "eben_bl_conv1d_bwd_dw_multi": (c_int, [POINTER(_D), POINTER(_P), POINTER(_P), c_int, POINTER(_P), POINTER(c_size_t), c_int, _P]),
    "eben_bl_head_fwd": (c_int, [POINTER(EbenBlHeadJob), c_int, c_int, _P]),
    "eben_bl_head_dx": (c_int, [POINTER(EbenBlHeadJob), c_int, c_int, _P, _P]),
    "eben_bl_head_dw_workspace": (c_size_t, [POINTER(EbenBlHeadJob), c_int, POINTER(c_int), POINTER(c_int)]),
    "eben_bl_head_dw": (c_int, [POINTER(EbenBlHeadJob), c_int, _P, c_size_t, _P]),
    "eben_bl_tail_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P]),
    "eben_bl_tail_dx": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_int, POINTER(c_int), c_int, c_int, _P, c_float,
                                _P, _P, _P]),
    "eben_bl_tail_dw_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "eben_bl_tail_dw": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "eben_bl_fm_sums_workspace": (c_size_t, [c_int]),
    "eben_bl_fm_sums": (c_int, [POINTER(c_void_p), POINTER(c_int64), c_int, _P, c_size_t, _P, _P]),
    "eben_bl_fm_sums_codes": (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), c_int, _P, c_size_t, _P, _P]),
    "eben_ru_packed_floats": (c_size_t, [c_int]),
    "eben_ru_pack": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "eben_ru_fwd": (c_int, [c_int, c_int, c_int, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P]),
    "eben_ru_pack_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "eben_ru_bwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_float, _P, c_float, _P, _P, _P, _P, _P]),
    "eben_ru_packed_floats_ex": (c_size_t, [c_int, c_int]),
    "eben_ru_supported": (c_int, [c_int, c_int, c_int]),
    "eben_ru_pack_ex": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "eben_ru_pack_multi": (c_int, [POINTER(EbenRuPackJob), c_int, _P]),
    "eben_ru_fwd_ex": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P]),
    "eben_ru_bwd_ex": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, c_float, _P, _P, _P, _P, _P]),
    "eben_ru_dw_slabs": (c_int, [c_int, c_int, c_int]),
    "eben_ru_dw": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, c_float, _P, _P, _P]),
    "eben_rubl_supported": (c_int, [c_int, c_int]),
    "eben_rubl_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "eben_rubl_bwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_float, _P, c_float, _P, _P, _P, _P, _P, _P]),
    "eben_rubl_dw_slabs": (c_int, [c_int, c_int, c_int]),
    "eben_rubl_dw": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "eben_fir_decimate": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_fir_interp_sum": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_lrelu_fwd": (c_int, [_P, _P, c_size_t, c_float, _P]),
    "eben_lrelu_bwd": (c_int, [_P, _P, _P, c_size_t, c_float, _P]),
    "eben_space_to_depth": (c_int, [_P, _P, c_float, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_gemm_packed_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "eben_gemm_pack": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P]),
    "eben_gemm_fwd": (c_int, [c_int, c_int, c_int, c_int, ctypes.c_longlong, _P, _P, _P, _P]),
    "eben_add": (c_int, [_P, _P, _P, c_size_t, _P]),
    "eben_axpby": (c_int, [_P, c_float, _P, c_float, _P, c_size_t, _P]),
    "eben_tanh_lift_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "eben_tanh_bwd": (c_int, [_P, _P, _P, c_size_t, _P]),
    "eben_reflect_pad_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "eben_reflect_pad_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "eben_fm_sums": (c_int, [POINTER(c_void_p), POINTER(c_int64), c_int, _P, c_size_t, _P, _P]),
    "eben_fm_sums_workspace": (c_size_t, [c_int]),
    "eben_fm_bwd": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int, _P, _P, c_float, _P]),
    "eben_hinge_fwd": (c_int, [_P, c_size_t, c_float, _P, _P]),
    "eben_hinge_fwd_multi": (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_float), c_int, _P, _P]),
    "eben_disc_losses": (c_int, [_P, c_int, c_float, _P, c_int, _P, _P]),
    "eben_stft_loss_total": (c_int, [POINTER(c_void_p), POINTER(c_float), c_int, c_int, _P, _P]),
    "eben_last_conv_norms_workspace": (c_size_t, [c_int]),
    "eben_last_conv_norms": (c_int, [POINTER(c_void_p), c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "eben_balance": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, _P, c_int, c_int, c_float, c_float, _P, _P, _P]),
    "eben_weighted_sum": (c_int, [POINTER(c_void_p), _P, c_int, c_size_t, _P, _P]),
    "eben_hinge_bwd": (c_int, [_P, c_size_t, c_float, _P, c_float, _P, _P]),
    "eben_hinge_bwd_stacked": (c_int, [_P, _P, c_size_t, _P, c_float, c_float, c_float, _P, _P]),
    "eben_stft_loss_sums": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, c_size_t, _P, _P]),
    "eben_stft_loss_sums_workspace": (c_size_t, [c_int]),
    "eben_stft_loss_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, c_float, _P, _P]),
    "eben_overlap_add": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_time_mask": (c_int, [_P, c_int64, c_int, c_int, c_int, _P]),
    "eben_phase_vocoder": (c_int, [_P, _P, c_int, c_int, c_int, c_int, ctypes.c_double, c_float, _P]),
    "eben_resample": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_stft_frames": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_stft_frames_folded": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "eben_split3": (c_int, [_P, _P, c_int, c_int, c_int64, _P]),
    "eben_overlap_add_folded": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, _P]),
    "eben_stft_loss_sums_ex": (c_int, [_P, _P, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_float, _P, c_size_t, _P, _P]),
    "eben_stft_loss_bwd_ex": (c_int, [_P, _P, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_float, _P, _P, c_float, _P, c_int64, c_int64,
                                      c_int64, _P]),
    "eben_overlap_add_ex": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, _P]),
    "eben_adam_step": (c_int, [POINTER(EbenAdamTensor), c_int, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P]),
    "eben_noisy_collate": (c_int, [POINTER(EbenCollateItem), c_int, c_int, _P, _P, _P]),
    "eben_l2norm": (c_int, [_P, c_size_t, _P, _P]),
}

_lib: Optional[ctypes.CDLL] = None


#: include/eben_hip.h EBEN_ABI_VERSION this module's structures and signatures were written against
ABI_VERSION = 3


def load(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the library and attach the prototypes.  Works without a GPU (symbols only)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("EBEN_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise EbenError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `vibravox_amd/csrc/build.sh` (there is no CPU fallback)."
        )
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    have = lib.eben_version()
    if have != ABI_VERSION:   # a stale .so (or a caller built against another header) would mis-stride the POD tables
        raise EbenError(f"{p} reports ABI version {have}, this package binds version {ABI_VERSION} (include/eben_hip.h EBEN_ABI_VERSION): rebuild it "
                        f"with vibravox_amd/csrc/build.sh")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().eben_last_error()
        raise EbenError(f"{what or 'libeben_hip'} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor the kernels may touch (contiguous float32 on a HIP device)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EbenError(
            f"vibravox_amd ops run only on an MI355X HIP device (got a tensor on '{t.device}'); "
            "there is no CPU path -- the CPU oracle lives under oracle/ and is test-only."
        )
    if t.dtype is not torch.float32 or not t.is_contiguous():
        raise EbenError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def stream() -> int:
    """Raw handle of torch's current HIP stream (two C calls: ~20x cheaper than torch.cuda.current_stream().cuda_stream,
    which builds a Python Stream object -- it was 2.6 ms of CPU per train step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
