#!/usr/bin/env python3
"""Timing of the fused entry-flow kernels at the bench shapes (100 copies of 512 x 512): asr_sepconv_fused_f16x3 64 -> 128 and
128 -> 128 at 256 x 256, asr_entry_stem_f16x3; us per launch (median of 9), GB/s on layer input + output, checksum."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
b = int(os.environ.get("BATCH", "100"))
def timeit(fn, n=9):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), out
for cin in (64, 128):
    x = torch.randn(b, 256, 256, cin, device=dev)
    wd = torch.randn(3, 3, cin, device=dev) * 0.3
    bd = torch.randn(cin, device=dev) * 0.1
    w16 = ops.pack_pw_weights_f16x3(torch.randn(cin, 128, device=dev) / cin ** 0.5)
    bp = torch.randn(128, device=dev) * 0.1
    t, y = timeit(lambda: ops.sepconv_fused(x, wd, bd, w16, bp, 128, pre_relu=(cin == 128), dw_relu=False, out_relu=False))
    gb = 4.0 * b * 256 * 256 * (cin + 128) / 1e9
    print(f"sepconv_fused {cin:3d} -> 128: {t:8.1f} us  {gb / t * 1e6:7.1f} GB/s  sha {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    del x, y
x = torch.rand(b, 512, 512, 3, device=dev)
w1 = torch.randn(3, 3, 3, 32, device=dev) * 0.3; b1 = torch.randn(32, device=dev) * 0.1
w2 = ops.pack_pw_weights_f16x3(torch.randn(288, 64, device=dev) / 17.0); b2 = torch.randn(64, device=dev) * 0.1
t, y = timeit(lambda: ops.entry_stem_fused(x, w1, b1, w2, b2))
gb = 4.0 * b * (512 * 512 * 3 + 256 * 256 * 64) / 1e9
print(f"entry_stem_fused          : {t:8.1f} us  {gb / t * 1e6:7.1f} GB/s  sha {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
