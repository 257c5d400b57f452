import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops, _lib
torch.manual_seed(0)
dev = torch.device("cuda")
k, n = 2048, 256
w = torch.randn(k, n, device=dev) / k ** 0.5
b = torch.randn(n, device=dev)
wp = ops.pack_pw_weights(w)
for B in (4, 8):
    bad = 0
    for rep in range(30):
        x = torch.randn(B, 16, 16, k, device=dev)
        pooled = torch.full((B, k), 7.0, device=dev)
        out = torch.empty((B, n), device=dev)
        torch.cuda.synchronize()
        _lib.call("asr_gap_f32", x.data_ptr(), pooled.data_ptr(), B, 256, k, k, _lib.stream_ptr())
        _lib.call("asr_pwconv_mfma_f32", pooled.data_ptr(), wp.data_ptr(), b.data_ptr(), None, out.data_ptr(), B, k, n, k, n, 0, 1, 1, 0, 0, _lib.stream_ptr())
        torch.cuda.synchronize()
        ref = (x.double().mean(dim=(1, 2)) @ w.double() + b.double()).relu()
        d = (out.double() - ref).abs().max().item()
        if d > 1e-3: bad += 1
    print("B", B, "bad", bad, "of 30", flush=True)
