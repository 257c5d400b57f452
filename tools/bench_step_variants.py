#!/usr/bin/env python3
"""One launch of a forward plan (picked by a substring of its label) timed against ablated / alternative builds of its entry
point, interleaved in one process, HIP events around each launch, medians.

    python tools/bench_step_variants.py middle_flow_unit_1_separable_conv1_depthwise dw_skip1:"one tap" dw_skip2:"f32 store" ...
    (variant libraries: <pkg>/libasr_hz_<name>.so from tools/build_hazard_variants.py)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, weights as W  # noqa: E402
from asr_amd.model import DeeplabModel  # noqa: E402

label = sys.argv[1]
variants = [a.split(":", 1) for a in sys.argv[2:]]
copies = int(os.environ.get("COPIES", "100"))
lib = _lib.load()
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (512, 512, 3), 21, final_upsample=False, last_activation=None)
x = torch.rand((copies, 512, 512, 3), device="cuda")
model.engine.forward(x)
torch.cuda.synchronize()
plan = model.engine.plan(copies, 512, 512, 0)
step = next(s for s in plan["steps"] if label in s[5])
name, args = step[0], step[1]
print(f"{name}  {step[5]}  ({step[4] / 1e6:.0f} MB algorithmic)")
forms = [("product", getattr(lib, name))]
pkg = os.path.dirname(_lib.LIB_PATH)
for key, what in variants:
    f = getattr(C.CDLL(os.path.join(pkg, f"libasr_hz_{key}.so")), name)
    f.restype, f.argtypes = _lib.SIGNATURES[name]
    forms.append((what, f))
times = {lbl: [] for lbl, _ in forms}
s = _lib.stream_ptr()
for r in range(11):
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
    print(f"{lbl:44s} {m:8.1f} us  ({m / base:.2f})  {step[4] / m / 1e6:6.2f} TB/s on the product's bytes")
