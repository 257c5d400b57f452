"""ctypes binding of libasr_hip.so (the C ABI declared in include/asr_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.
PyTorch is used only as the owner of device memory and streams (``tensor.data_ptr()``,
``torch.cuda.current_stream()``).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# ASR_LIB selects another build of the same ABI (a diagnostic variant of csrc/build.py); default: the product library
LIB_PATH = os.environ.get("ASR_LIB") or os.path.join(_PKG_DIR, "libasr_hip.so")

ASR_OK = 0


class AsrError(RuntimeError):
    """A libasr_hip.so entry point returned a negative status."""


class AsrLibraryMissing(ImportError):
    """libasr_hip.so has not been built (run __graft_entry__.build() or csrc/build.py)."""


_f = C.POINTER(C.c_float)
_vp = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_fl = C.c_float
_sz = C.c_size_t

OPT_ADAM, OPT_SGD, OPT_ADAGRAD, OPT_ADADELTA, OPT_ADAMAX = 0, 1, 2, 3, 4
PRIOR_TV, PRIOR_BTV = 0, 1


class SrConfig(C.Structure):
    """struct asr_sr_config (include/asr_hip.h): update rule + prior of the *_cfg SR entry points."""
    _fields_ = [("optimizer", C.c_int), ("flag", C.c_int), ("c0", C.c_float), ("c1", C.c_float), ("c2", C.c_float),
                ("prior", C.c_int), ("btv_alpha", C.c_float), ("btv_shift", C.c_int), ("plane_chunk", C.c_int)]


_cfg = C.POINTER(SrConfig)
ABI_VERSION = 3          # ASR_ABI_VERSION of include/asr_hip.h this table mirrors

# name -> (restype, argtypes).  Order and types mirror include/asr_hip.h exactly.
SIGNATURES = {
    "asr_last_error": (C.c_char_p, []),
    "asr_abi_version": (_i, []),
    "asr_target_arch": (C.c_char_p, []),
    "asr_warp_affine_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_warp_affine_nearest_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_augment_copies_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "asr_sr_init_target_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_sr_forward_residual_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_sr_backward_adam_f32": (_i, [_vp] * 10 + [_i] * 6 + [_fl] * 7 + [_i, _vp]),
    "asr_sr_loss_terms_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_sr_solve_workspace_bytes": (_sz, [_i] * 6),
    "asr_sr_solve_workspace_bytes_cfg": (_sz, [_i] * 6 + [_cfg]),
    "asr_sr_solve_f32": (_i, [_vp] * 10 + [_i, _vp, _vp, _sz] + [_i] * 6 + [_fl] * 7 + [_i, _vp]),
    "asr_sr_backward_cfg_f32": (_i, [_vp] * 10 + [_i] * 6 + [_fl] * 4 + [_cfg, _vp]),
    "asr_sr_loss_terms_cfg_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _cfg, _vp]),
    "asr_sr_solve_cfg_f32": (_i, [_vp] * 10 + [_i, _vp, _vp, _sz] + [_i] * 6 + [_fl] * 4 + [_cfg, _vp]),
    "asr_realign_max_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_realign_mean_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_realign_max_mean_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_minmax_f32": (_i, [_vp, _vp, _i64, _i, _vp]),
    "asr_class_activation_f32": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "asr_argmax_i32": (_i, [_vp, _vp, _i64, _i, _vp]),
    "asr_opm_argmax_f32": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "asr_opm_slice_max_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "asr_opm_slice_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _fl, _fl, _vp]),
    "asr_threshold_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _fl, _i, _vp]),
    "asr_iou_counts_i32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "asr_iou_counts_shared_truth_i32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "asr_minmax_normalize_f32": (_i, [_vp, _vp, _vp, _i64, _i, _fl, _fl, _vp]),
    "asr_standard_mask_i32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_class_counts_i32": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "asr_pwconv_packed_floats": (_sz, [_i, _i]),
    "asr_pwconv_pack_weights_f32": (_i, [_vp, _vp, _i, _i, _vp]),
    "asr_pwconv_mfma_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_pwconv_packed_floats_f16x3": (_sz, [_i, _i]),
    "asr_pwconv_pack_weights_f16x3": (_i, [_vp, _vp, _i, _i, _vp]),
    "asr_pwconv_mfma_f16x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_conv3x3_mfma_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "asr_conv3x3_mfma_f16x3": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "asr_dwconv3x3_nhwc_split_f16": (_i, [_vp, _vp, _vp, _vp] + [_i] * 14 + [_vp]),
    "asr_pwconv_mfma_f16x3_presplit": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "asr_conv3x3_direct_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "asr_conv3x3_stem_f16x3": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "asr_entry_stem_f16x3": (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    "asr_sepconv_fused_f16x3": (_i, [_vp] * 6 + [_i] * 10 + [_vp]),
    "asr_dwconv3x3_nhwc_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "asr_aspp_dwconv3_nhwc_f32": (_i, [_vp] * 6 + [_i] * 11 + [_vp]),
    "asr_aspp_dwconv3_nhwc_split_f16": (_i, [_vp] * 6 + [_i] * 11 + [_vp]),
    "asr_aspp_dwconv3_supported": (_i, [_i] * 5),
    "asr_gap_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "asr_resize_bilinear_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle, with argtypes/restype set for every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsrLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`python deeplabv3plus-augmented-superresolution_amd/csrc/build.py`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.asr_abi_version() != ABI_VERSION:
        raise AsrError(f"libasr_hip.so ABI version {lib.asr_abi_version()} != {ABI_VERSION} (stale build? python "
                       f"deeplabv3plus-augmented-superresolution_amd/csrc/build.py --force)")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != ASR_OK:
        msg = load().asr_last_error().decode("utf-8", "replace")
        raise AsrError(f"{what or 'libasr_hip call'} failed (status {rc}): {msg}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32, allow_none=False):
    """Device pointer of a contiguous CUDA(ROCm) tensor; raises instead of copying silently."""
    if t is None:
        if allow_none:
            return None
        raise AsrError("null tensor passed to libasr_hip")
    if not isinstance(t, torch.Tensor):
        raise AsrError(f"expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise AsrError("libasr_hip needs device memory: tensor is on the CPU (no CPU fallback exists)")
    if dtype is not None and t.dtype != dtype:
        raise AsrError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise AsrError("tensor must be contiguous")
    return t.data_ptr()


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def require_gpu():
    if not torch.cuda.is_available():
        raise AsrError("no ROCm device visible: the asr_amd product path runs only on the GPU "
                       "(hand-written gfx950 kernels, no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())
