#!/usr/bin/env python3
"""What the fused stem's time is made of: the product launch against ablated builds (tools/build_hazard_variants.py: stem_skip*),
interleaved in one process, HIP events around each launch, medians.

    python tools/bench_stem_ablation.py [copies]
"""
import ctypes as C
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, weights as W  # noqa: E402
from asr_amd.model import DeeplabModel  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NAME = "asr_entry_stem_f16x3"
what = {"1": "no image loads", "2": "stage 2 with 1 tap of 9", "4": "no output stores", "8": "no stage 1", "12": "stage 2 alone, no stores",
        "6": "stage 1 alone (1 tap, no stores)"}
lib = _lib.load()
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (512, 512, 3), 21, final_upsample=False, last_activation=None)
x = torch.rand((copies, 512, 512, 3), device="cuda")
model.engine.forward(x)
torch.cuda.synchronize()
plan = model.engine.plan(copies, 512, 512, 0)
args = next(s[1] for s in plan["steps"] if s[0] == NAME)
forms = [("product", getattr(lib, NAME))]
pkg = os.path.dirname(_lib.LIB_PATH)
for key in what:
    path = os.path.join(pkg, f"libasr_hz_stem_skip{key}.so")
    if os.path.exists(path):
        f = getattr(C.CDLL(path), NAME)
        f.restype, f.argtypes = _lib.SIGNATURES[NAME]
        forms.append((what[key], f))
times = {lbl: [] for lbl, _ in forms}
s = _lib.stream_ptr()
for r in range(9):
    for lbl, f in forms:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert f(*args, s) == 0
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[lbl].append(e0.elapsed_time(e1) * 1e3)
base = np.median(times["product"])
for lbl, _ in forms:
    m = np.median(times[lbl])
    print(f"{lbl:36s} {m:8.1f} us  ({m / base:.2f})")
