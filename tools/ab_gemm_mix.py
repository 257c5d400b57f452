#!/usr/bin/env python3
"""Interleaved A/B of the pre-split pointwise GEMM forms over the REAL layer mix of a forward pass, in one process.

The 65 asr_pwconv_mfma_f16x3_presplit launches of a 100-copy 512 x 512 forward pass (or the 4 x 65 of a BASELINE
configs[4] image with --config 4: 50 copies of 1024 x 1024) are replayed from the engine's launch plan, with the plan's
own operands, leading dimensions and residuals; per launch the kernel forms alternate (A B C A B C ...), every launch
timed with its own HIP events.  Printed: per distinct shape the medians and ratios, and the launch-weighted sum per
forward pass -- the figure bench.py reports as kernel_time_ms_per_step.pw16 (minus the 8 in-kernel-split launches).

    ASR_BUILD_VARIANT=diag python <pkg>/csrc/build.py && ASR_LIB=<pkg>/libasr_hip_diag.so python tools/ab_gemm_mix.py
"""
import argparse
import collections
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
ap.add_argument("--rounds", type=int, default=9)
args = ap.parse_args()
size, batch = (512, 100) if args.config == 1 else (1024, 50)

lib = _lib.load()
sig = _lib.SIGNATURES["asr_pwconv_mfma_f16x3_presplit"][1]
forms = []
for sym, label in (("asr_diag_pwconv_presplit_8w", "8-wave (round 1)"), ("asr_diag_pwconv_presplit_lw", "loader waves (round 2)")):
    if hasattr(lib, sym):
        f = getattr(lib, sym)
        f.restype, f.argtypes = C.c_int, sig
        forms.append((label, f))
forms.append(("product", lib.asr_pwconv_mfma_f16x3_presplit))
NV = len(forms)

dev = torch.device("cuda")
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (size, size, 3), 21, final_upsample=False, last_activation=None)
x = torch.rand((batch, size, size, 3), device=dev)
model.predict_device(x, batch_size=batch)
torch.cuda.synchronize()
plan = model.engine.plan(batch, size, size)
steps = plan["steps"]
s = _lib.stream_ptr()
PRODUCERS = ("asr_dwconv3x3_nhwc_split_f16", "asr_aspp_dwconv3_nhwc_split_f16")
last_producer = None
per_shape = collections.OrderedDict()
tot = np.zeros(NV)
nl = 0
for idx, (name, a, kind, fl, by, label) in enumerate(steps):
    if name in PRODUCERS:
        last_producer = (name, a)
    if name != "asr_pwconv_mfma_f16x3_presplit":
        continue
    if last_producer is not None:                    # refresh the A operand (its buffer has been recycled since the forward pass)
        _lib.check(getattr(lib, last_producer[0])(*last_producer[1], s), last_producer[0])
    m, k, n, res = a[5], a[6], a[7], a[3] is not None
    outs = []
    ev = [[] for _ in range(NV)]
    for r in range(args.rounds + 2):
        for v, (_lbl, fn) in enumerate(forms):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(fn(*a, s), "presplit")
            e1.record()
            if r >= 2:
                ev[v].append((e0, e1))
            elif r == 1:                             # bitwise comparison of the forms on this layer's real operands
                torch.cuda.synchronize()
                outs.append(plan["outs"][idx].t.clone() if not isinstance(plan["outs"][idx], (list, tuple)) else None)
    torch.cuda.synchronize()
    med = np.array([np.median([p.elapsed_time(q) * 1e3 for p, q in ev[v]]) for v in range(NV)])
    same = all(o is None or outs[0] is None or torch.equal(o, outs[0]) for o in outs[1:])
    key = (m, k, n, res)
    rec = per_shape.setdefault(key, dict(count=0, t=np.zeros(NV), same=True))
    rec["count"] += 1
    rec["t"] += med
    rec["same"] &= same
    tot += med
    nl += 1
print(f"config {args.config}: {nl} presplit launches per forward pass of {batch} copies at {size}x{size}; forms: " + ", ".join(f[0] for f in forms))
for (m, k, n, res), rec in per_shape.items():
    t = rec["t"] / rec["count"]
    print(f"M={m:8d} K={k:5d} N={n:5d} res={int(res)} x{rec['count']:3d}: " + "  ".join(f"{v:8.1f} us" for v in t) +
          "   vs first: " + " ".join(f"{v / t[0]:.3f}" for v in t[1:]) + f"   bit-identical {rec['same']}")
print("sum per forward pass: " + "  ".join(f"{v / 1e3:8.3f} ms" for v in tot) + "   vs first: " + " ".join(f"{v / tot[0]:.3f}" for v in tot[1:]))
