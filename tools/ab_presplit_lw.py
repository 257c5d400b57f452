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
fns = [("8-wave (round 1)", lib.asr_diag_pwconv_presplit_8w)]
for _sym, _label in (("asr_diag_pwconv_presplit_lw", "loader waves, two stages (round 2)"),
                     ("asr_diag_pwconv_presplit_exp", "MOCK fused (timing only, wrong results)")):   # csrc/diag/gemm_diag.hip
    if hasattr(lib, _sym):
        fns.append((_label, getattr(lib, _sym)))
for _n, _f in fns:
    _f.restype = C.c_int
    _f.argtypes = _lib.SIGNATURES["asr_pwconv_mfma_f16x3_presplit"][1]
fns.append(("ring (product)", lib.asr_pwconv_mfma_f16x3_presplit))
NV = len(fns)
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
    outs = [torch.full((m, n), float("nan"), device=dev) for _ in range(NV)]
    s = _lib.stream_ptr()

    def launch(which):
        f = fns[which][1]
        _lib.check(f(_lib.ptr(xs), _lib.ptr(w16), _lib.ptr(bias), _lib.ptr(r, allow_none=True), _lib.ptr(outs[which]), m, c, n,
                     chunks, n, n if res else 0, 0, s), "presplit")

    for _ in range(3):
        for which in range(NV):
            launch(which)
    torch.cuda.synchronize()
    ev = [[] for _ in range(NV)]
    for _ in range(rounds):
        for which in range(NV):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(which)
            e1.record()
            ev[which].append((e0, e1))
    torch.cuda.synchronize()
    t = [np.array([a.elapsed_time(b_) * 1e3 for a, b_ in ev[w]]) for w in range(NV)]
    same = all(bool(torch.equal(outs[0], outs[w])) for w in range(1, NV) if "MOCK" not in fns[w][0])
    med = [float(np.median(v)) for v in t]
    print(f"M={m} K={c} N={n} res={int(res)}: " + "   ".join(f"{fns[w][0]} {med[w]:8.1f} us" for w in range(NV)) +
          "   ratios " + " ".join(f"{med[w] / med[0]:.3f}" for w in range(1, NV)) + f"   bit-identical {same}", flush=True)
