#!/usr/bin/env python3
"""Interleaved A/B of the product pre-split GEMM (asr_pwconv_mfma_f16x3_presplit: 8 MFMA waves + 4 loader waves) against
round 1's 8-wave form of it (asr_diag_pwconv_presplit_8w, diagnostic library only) in ONE process: the chip's clock drifts by 10-20 % with its thermal / power state, so
variants timed in separate processes (or one after the other) are not comparable.  Per shape the two kernels alternate
launch by launch (A B A B ...), every launch is timed with its own HIP events; medians and the bitwise comparison are
printed.

    ASR_BUILD_VARIANT=diag python <pkg>/csrc/build.py && ASR_LIB=<pkg>/libasr_hip_diag.so python tools/ab_presplit_lw.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, ops  # noqa: E402

lib = _lib.load()
fn_8w = lib.asr_diag_pwconv_presplit_8w
fn_8w.restype = C.c_int
fn_8w.argtypes = _lib.SIGNATURES["asr_pwconv_mfma_f16x3_presplit"][1]
dev = torch.device("cuda")
torch.manual_seed(0)
shapes = [(100, 32, 728, 728, False), (100, 32, 728, 728, True), (100, 32, 1536, 2048, False), (100, 32, 1024, 1536, False),
          (100, 128, 256, 256, False)]
rounds = int(os.environ.get("AB_ROUNDS", "30"))
for b, hw, c, n, res in shapes:
    m = b * hw * hw
    x = torch.randn(b, hw, hw, c, device=dev)
    wd = torch.zeros(3, 3, c, device=dev)
    wd[1, 1] = 1.0
    w16 = ops.pack_pw_weights_f16x3(torch.randn(c, n, device=dev) / c ** 0.5)
    bias = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    xs, _, chunks = ops.dwconv3x3_split(x, wd, torch.zeros(c, device=dev))
    outs = [torch.empty(m, n, device=dev) for _ in range(2)]
    s = _lib.stream_ptr()

    def launch(which):
        f = fn_8w if which == 0 else lib.asr_pwconv_mfma_f16x3_presplit
        _lib.check(f(_lib.ptr(xs), _lib.ptr(w16), _lib.ptr(bias), _lib.ptr(r, allow_none=True), _lib.ptr(outs[which]), m, c, n,
                     chunks, n, n if res else 0, 0, s), "presplit")

    for _ in range(4):
        launch(0)
        launch(1)
    torch.cuda.synchronize()
    ev = [[], []]
    for _ in range(rounds):
        for which in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(which)
            e1.record()
            ev[which].append((e0, e1))
    torch.cuda.synchronize()
    t = [np.array([a.elapsed_time(b_) * 1e3 for a, b_ in ev[w]]) for w in (0, 1)]
    same = bool(torch.equal(outs[0], outs[1]))
    print(f"M={m} K={c} N={n} res={int(res)}: 8-wave (round 1) {np.median(t[0]):8.1f} us (min {t[0].min():.1f})   loader-wave (product) "
          f"{np.median(t[1]):8.1f} us (min {t[1].min():.1f})   ratio {np.median(t[1]) / np.median(t[0]):.3f}   bit-identical {same}",
          flush=True)
