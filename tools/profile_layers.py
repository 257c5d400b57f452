#!/usr/bin/env python3
"""Per-layer HIP-event timing of the DeepLabV3+ plan (diagnostic; prints TF/s and GB/s per launch).

    python tools/profile_layers.py [--batch 50] [--size 512] [--reps 3]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--top", type=int, default=200)
    a = ap.parse_args()
    from asr_amd import weights as W
    from asr_amd.model import DeeplabModel
    model = DeeplabModel(W.make_synthetic_weights(1234), (a.size, a.size, 3), 21, False, None)
    x = torch.rand((a.batch, a.size, a.size, 3), device=model.device)
    model.engine.forward(x)
    torch.cuda.synchronize()
    best = None
    for _ in range(a.reps):
        prof = {}
        model.engine.forward(x, profile=prof)
        d = prof["_detail"]
        best = d if best is None else [(k, l, min(m0, m1), f, b) for (k, l, m0, f, b), (_, _, m1, _, _) in zip(best, d)]
    tot = sum(m for _, _, m, _, _ in best)
    print(f"batch {a.batch} size {a.size}: {tot:.2f} ms/forward = {tot / a.batch:.3f} ms/copy; pool {model.engine.plan(a.batch, a.size, a.size)['pool_bytes'] / 2**30:.2f} GiB")
    agg = {}
    for k, l, m, f, b in best:
        g = agg.setdefault(k, [0, 0, 0])
        g[0] += m; g[1] += f; g[2] += b
    for k, (m, f, b) in agg.items():
        print(f"  {k:5s} {m:8.2f} ms  {f / m / 1e9 if m else 0:8.1f} TF/s  {b / m / 1e6 if m else 0:8.1f} GB/s")
    shapes = {}
    for k, l, m, f, b in best:
        if k in ("pw", "pw16", "pw16s", "dw"):
            key = k + " " + " ".join(l.split()[1:])
            g = shapes.setdefault(key, [0, 0.0, 0.0, 0.0])
            g[0] += 1; g[1] += m; g[2] += f; g[3] += b
    print("-- by shape (count, total ms, TF/s, GB/s) --")
    for key, (c, m, f, b) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
        print(f"{key:48s} x{c:3d} {m:8.3f} ms {f / m / 1e9:8.1f} TF/s {b / m / 1e6:8.1f} GB/s")
    for k, l, m, f, b in best[:a.top]:
        print(f"{k:5s} {l:70s} {m * 1e3:9.1f} us {f / m / 1e9:8.1f} TF/s {b / m / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
