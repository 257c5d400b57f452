#!/usr/bin/env python3
"""Split-f16 GEMM: in-kernel split (f32 A, 128x128 tile) against the pre-split LDS-DMA kernel (256x256 tile) on the
separable-conv shapes of the net."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
import hashlib
dev = torch.device("cuda")
torch.manual_seed(0)
shapes = [(100, 32, 728, 728, True), (100, 32, 728, 728, False), (100, 32, 1536, 2048, False), (100, 32, 1024, 1536, False),
          (100, 128, 256, 256, False), (100, 64, 256, 728, False)]
for b, hw, c, n, res in shapes:
    m = b * hw * hw
    x = torch.randn(b, hw, hw, c, device=dev)
    wd = torch.zeros(3, 3, c, device=dev); wd[1, 1] = 1.0
    bd = torch.zeros(c, device=dev)
    w = torch.randn(c, n, device=dev) / c ** 0.5
    bias = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    w16 = ops.pack_pw_weights_f16x3(w)
    xs, _, chunks = ops.dwconv3x3_split(x, wd, bd)
    out = torch.empty(m, n, device=dev)
    line = f"M={m} K={c} N={n} res={int(res)}:"
    for name, fn in (("in-kernel split 128x128", lambda: ops.pwconv(x.reshape(m, c), w16, bias, c, n, out=out, residual=r, f16x3=True)),
                     ("pre-split LDS-DMA 256x256", lambda: ops.pwconv_presplit(xs, w16, bias, c, n, chunks, out=out, residual=r))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += f"  {name} {ms * 1e3:8.1f} us {2.0 * m * c * n / ms / 1e9:7.1f} TF/s sha {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]}"
    print(line, flush=True)
