import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda")
for m in (4, 8, 50, 3):
    k, n = 2048, 256
    x = torch.randn(m, k, device=dev).abs()
    w = torch.randn(k, n, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    wp = ops.pack_pw_weights(w)
    ref = (x.double() @ w.double() + b.double()).relu()
    worst = 0
    outs = []
    for rep in range(20):
        got = ops.pwconv(x, wp, b, k, n, relu=True)
        outs.append(got.clone())
        worst = max(worst, (got.double() - ref).abs().max().item())
        # disturb LDS / cache state with another GEMM
        ops.pwconv(torch.randn(1024, 728, device=dev), ops.pack_pw_weights(torch.randn(728, 728, device=dev)), None, 728, 728)
    rr = max((o - outs[0]).abs().max().item() for o in outs)
    print("M", m, "worst vs ref", worst, "run-to-run", rr, flush=True)
