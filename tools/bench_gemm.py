#!/usr/bin/env python3
"""Micro-benchmark of the pointwise GEMM kernels on the shapes of the net (f32 MFMA vs split-f16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
dev = torch.device("cuda")
shapes = [(102400, 728, 728, True), (102400, 728, 728, False), (409600, 728, 728, True), (204800, 728, 728, True), (102400, 1536, 2048, False), (102400, 2048, 256, False), (1638400, 256, 256, False),
          (6553600, 128, 128, False), (409600, 728, 728, False), (102400, 1024, 1536, False)]
for m, k, n, res in shapes:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(k, n, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    out = torch.empty(m, n, device=dev)
    w32, w16 = ops.pack_pw_weights(w), ops.pack_pw_weights_f16x3(w)
    line = f"M={m} K={k} N={n} res={int(res)}:"
    for name, wp, f16 in (("f32", w32, False), ("f16x3", w16, True)):
        for _ in range(2):
            ops.pwconv(x, wp, b, k, n, out=out, residual=r, f16x3=f16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.pwconv(x, wp, b, k, n, out=out, residual=r, f16x3=f16)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += f"  {name} {ms * 1e3:8.1f} us {2.0 * m * k * n / ms / 1e9:7.1f} TF/s"
    print(line, flush=True)
