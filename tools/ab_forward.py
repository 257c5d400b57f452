#!/usr/bin/env python3
"""In-situ interleaved A/B of the pre-split GEMM forms: whole forward passes (every kernel in its real place, operands as the
producing kernel left them in the caches), alternating the form behind asr_pwconv_mfma_f16x3_presplit pass by pass in ONE
process.  Per form: the HIP-event sum over the pre-split launches and over the whole pass (medians over the rounds).

    ASR_BUILD_VARIANT=diag python <pkg>/csrc/build.py && ASR_LIB=<pkg>/libasr_hip_diag.so python tools/ab_forward.py
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, weights as W  # noqa: E402
from asr_amd.model import DeeplabModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1, choices=(1, 4))
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
size, batch = (512, 100) if args.config == 1 else (1024, 50)
lib = _lib.load()
sig = _lib.SIGNATURES["asr_pwconv_mfma_f16x3_presplit"][1]
product = lib.asr_pwconv_mfma_f16x3_presplit
forms = [("product", product)]
for sym, label in (("asr_diag_pwconv_presplit_lw", "loader waves (round 2)"), ("asr_diag_pwconv_presplit_8w", "8-wave (round 1)")):
    if hasattr(lib, sym):
        f = getattr(lib, sym)
        f.restype, f.argtypes = C.c_int, sig
        forms.append((label, f))
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (size, size, 3), 21, final_upsample=False, last_activation=None)
x = torch.rand((batch, size, size, 3), device="cuda")
ref = None
res = {lbl: [] for lbl, _ in forms}
for r in range(args.rounds + 1):
    for lbl, fn in forms:
        lib.asr_pwconv_mfma_f16x3_presplit = fn            # the engine looks the entry point up by name at every launch
        prof = {}
        out = model.engine.forward(x, profile=prof).clone()
        if ref is None:
            ref = out
        assert torch.equal(out, ref), f"{lbl}: logits differ"
        pre = sum(ms for kind, label, ms, fl, by in prof["_detail"] if "presplit" in label)
        total = sum(ms for kind, label, ms, fl, by in prof["_detail"])
        if r:
            res[lbl].append((pre, total, prof.get("dw", [0])[0]))
lib.asr_pwconv_mfma_f16x3_presplit = product
print(f"config {args.config}: forward pass of {batch} copies at {size}x{size}, {args.rounds} interleaved rounds, logits bit-identical across forms")
base = None
for lbl, _ in forms:
    a = np.median(np.array(res[lbl]), axis=0)
    base = base if base is not None else a
    print(f"  {lbl:26s} pre-split GEMMs {a[0]:8.3f} ms ({a[0] / base[0]:.3f})   whole pass {a[1]:8.3f} ms ({a[1] / base[1]:.3f})   depthwise {a[2]:7.3f} ms")
