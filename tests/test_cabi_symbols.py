"""The C-ABI library loads without a GPU and exports every symbol include/asr_hip.h declares
(no compute calls here)."""
import os
import re

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "asr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_expected_surface():
    names = _declared_functions()
    for must in ("asr_warp_affine_f32", "asr_augment_copies_f32", "asr_sr_forward_residual_f32",
                 "asr_sr_backward_adam_f32", "asr_sr_solve_f32", "asr_realign_max_f32", "asr_realign_mean_f32",
                 "asr_threshold_f32", "asr_iou_counts_i32", "asr_opm_argmax_f32", "asr_opm_slice_f32",
                 "asr_opm_slice_max_f32", "asr_pwconv_mfma_f32", "asr_dwconv3x3_nhwc_f32", "asr_conv3x3_mfma_f32",
                 "asr_conv3x3_direct_f32", "asr_gap_f32", "asr_resize_bilinear_f32"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    from asr_amd import _lib
    declared = _declared_functions()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/asr_hip.h but not exported"
    # and the ctypes signature table covers the header one-to-one
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.asr_abi_version() == _lib.ABI_VERSION == 3
    assert lib.asr_target_arch() == b"gfx950"


def test_argument_validation_needs_no_gpu(lib):
    """Host-side checks reject bad arguments before any launch (safe on a CPU-only box)."""
    from asr_amd import _lib
    rc = lib.asr_warp_affine_f32(None, None, None, 1, 0, 0, 4, 4, 4, 4, 1, None)
    assert rc == -1 and b"null pointer" in lib.asr_last_error()
    rc = lib.asr_pwconv_packed_floats(728, 728)
    assert rc == 736 * 768
    # the pre-split GEMM addresses a tile's operands by 32-bit offsets: operands beyond that are refused, not wrapped
    fake = 1 << 20                                               # non-null, 128-byte aligned; never dereferenced on the host
    rc = lib.asr_pwconv_mfma_f16x3_presplit(fake, fake, None, None, fake, 1024, 40960, 40960, 1280, 40960, 0, 0, None)
    assert rc == -2 and b"32-bit tile offsets" in lib.asr_last_error()
    # residuals + ping-pong x + running data-term sum + bordered planes (one chunk of copies + x per image) + one flag per image
    assert lib.asr_sr_solve_workspace_bytes(2, 3, 8, 8, 4, 4) == 4 * (2 * 3 * 16 + 2 * 2 * 64 + (2 * 3 + 2) * (8 + 4) * (8 + 64) + 2)
    import ctypes as C
    from asr_amd import ops
    per = lambda chunk: 4 * (100 * 16 + 2 * 64 + (chunk + 1) * (8 + 4) * (8 + 64) + 1)
    assert lib.asr_sr_solve_workspace_bytes(1, 100, 8, 8, 4, 4) == per(100)           # default: all copies at once (< 1 GiB)
    assert lib.asr_sr_solve_workspace_bytes_cfg(1, 100, 8, 8, 4, 4, C.byref(ops.sr_config(plane_chunk=100))) == per(100)
    assert lib.asr_sr_solve_workspace_bytes_cfg(1, 100, 8, 8, 4, 4, C.byref(ops.sr_config(plane_chunk=7))) == per(7)
    assert lib.asr_sr_solve_workspace_bytes_cfg(1, 200, 8, 8, 4, 4, C.byref(ops.sr_config(plane_chunk=29))) == 4 * (200 * 16 + 2 * 64 + (29 + 1) * 12 * 72 + 1)
    # beyond 1 GiB of planes the default splits evenly: 2000 copies of 516 x 576 floats = 2.4 GB -> 3 chunks of 667
    big = lambda chunk: 4 * (2000 * 128 * 128 + 2 * 512 * 512 + (chunk + 1) * 516 * 576 + 1)
    assert lib.asr_sr_solve_workspace_bytes(1, 2000, 512, 512, 128, 128) == big(667)


def test_product_refuses_cpu_tensors(lib):
    """No CPU fallback: a host tensor is an error, not a silent slow path."""
    import pytest
    import torch
    from asr_amd import ops, _lib
    with pytest.raises(_lib.AsrError):
        ops.minmax(torch.zeros(16))


def test_every_translation_unit_is_built_without_packed_f32_by_default():
    """csrc/build.py: packed-f32 instructions are off for every device translation unit; kernels opt back in one by one with
    ASR_PK_F32 (asr_common.h), and only kernels too large to share a SIMD with the fused entry-flow kernels may (DESIGN.md 4.1;
    tests/test_isa_guard.py checks the machine code).  The warp / SR / reduce units also keep -ffp-contract=off."""
    import importlib.util, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "asr_build", os.path.join(here, "deeplabv3plus-augmented-superresolution_amd", "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    flags = dict(b.SOURCES)
    for src in ("sr.hip", "warp.hip", "reduce.hip"):
        assert "-ffp-contract=off" in flags[src], src
    for src, fl in flags.items():
        if src.endswith(".hip"):       # off by default; a unit of POSTPASS keeps it on and has its op_sel:[0,1] instructions split instead
            assert ("-packed-fp32-ops" in fl) != (src in b.POSTPASS), src
    assert b.POSTPASS == {"sr.hip"}


def test_aspp_geometry_query_is_host_arithmetic(lib):
    """asr_aspp_dwconv3_supported needs no GPU: the plan asks it whether the fused ASPP depthwise kernel can stage a plane
    (engine.py), so a plan never meets ASR_ERR_UNSUPPORTED for geometry at run time."""
    assert lib.asr_aspp_dwconv3_supported(32, 32, 6, 12, 18) == 1          # configs[1]: 512 / 16
    assert lib.asr_aspp_dwconv3_supported(64, 64, 6, 12, 18) == 1          # configs[4]: 1024 / 16
    assert lib.asr_aspp_dwconv3_supported(64, 64, 12, 24, 36) == 1         # OS 8
    assert lib.asr_aspp_dwconv3_supported(4096, 4096, 1, 2, 3) == 0        # gcd 1: the whole plane would have to fit
    assert lib.asr_aspp_dwconv3_supported(0, 32, 6, 12, 18) == 0 and lib.asr_aspp_dwconv3_supported(32, 32, 0, 12, 18) == 0
