#!/usr/bin/env python3
"""HBM copy ceiling at the sizes of the depthwise layers (torch's elementwise copy kernel), for comparison with
tools/profile_layers.py's GB/s column (same accounting: bytes read + bytes written)."""
import torch
dev = torch.device("cuda")
for name, shape in (("middle flow 100x32x32x736", (100, 32, 32, 736)), ("decoder 100x128x128x256", (100, 128, 128, 256)),
                    ("entry 100x256x256x128", (100, 256, 256, 128))):
    x = torch.rand(shape, device=dev)
    y = torch.empty_like(x)
    for fn_name, fn in (("copy", lambda: y.copy_(x)), ("relu", lambda: torch.relu(x, out=y) if False else torch.clamp_min(x, 0, out=y)),
                        ("read-only sum", lambda: x.sum())):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        nbytes = x.numel() * 4 * (1 if fn_name.startswith("read") else 2)
        print(f"{name:28s} {fn_name:14s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
