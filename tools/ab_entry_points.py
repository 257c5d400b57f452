#!/usr/bin/env python3
"""Interleaved A/B of two builds of the library inside whole forward passes: the named entry points are taken from library B in
every second pass (the engine looks an entry point up by name at every launch), everything else -- and the order of the
launches, the state of the caches, the chip's clock -- stays what it is in the product.  Reports HIP-event sums per kernel
family and checks that the logits are bit-identical.

    python tools/ab_entry_points.py --lib-b <pkg>/libasr_hz_dwclamped.so --entries asr_dwconv3x3_nhwc_split_f16,asr_aspp_dwconv3_nhwc_split_f16
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
ap.add_argument("--lib-b", required=True)
ap.add_argument("--entries", required=True)
ap.add_argument("--config", type=int, default=1, choices=(1, 4))
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--label-b", default="B")
args = ap.parse_args()
size, batch = (512, 100) if args.config == 1 else (1024, 50)
lib = _lib.load()
libb = C.CDLL(os.path.abspath(args.lib_b))
entries = [e for e in args.entries.split(",") if e]
fa = {e: getattr(lib, e) for e in entries}
fb = {}
for e in entries:
    f = getattr(libb, e)
    f.restype, f.argtypes = _lib.SIGNATURES[e]
    fb[e] = f
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (size, size, 3), 21, final_upsample=False, last_activation=None)
x = torch.rand((batch, size, size, 3), device="cuda")
ref = None
res = {"product": [], args.label_b: []}
kinds = None
for r in range(args.rounds + 1):
    for lbl, table in (("product", fa), (args.label_b, fb)):
        for e, f in table.items():
            setattr(lib, e, f)
        prof = {}
        out = model.engine.forward(x, profile=prof).clone()
        if ref is None:
            ref = out
        same = torch.equal(out, ref)
        kinds = kinds or sorted(k for k in prof if not k.startswith("_"))
        if r:
            res[lbl].append([prof[k][0] for k in kinds] + [sum(prof[k][0] for k in kinds), float(same)])
for e, f in fa.items():
    setattr(lib, e, f)
print(f"config {args.config}: forward pass of {batch} copies at {size}x{size}, {args.rounds} interleaved rounds; entries from B: {entries}")
print(" " * 12 + "  ".join(f"{k:>9}" for k in kinds + ["sum", "identical"]))
base = None
for lbl in res:
    a = np.median(np.array(res[lbl]), axis=0)
    base = a if base is None else base
    print(f"{lbl:12s}" + "  ".join(f"{v:9.3f}" for v in a) + "    x" + "  ".join(f"{v / b:6.3f}" for v, b in zip(a[:-1], base[:-1]) if b))
