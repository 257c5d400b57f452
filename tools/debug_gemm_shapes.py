import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda")
for (m, k, n, res, relu) in [(65536, 256, 21, False, False), (65536, 256, 48, False, True), (65536, 304, 256, False, True),
                            (65536, 64, 128, False, False), (16384, 128, 256, True, False), (4096, 728, 728, True, False),
                            (1024, 728, 728, True, False), (1024, 2048, 256, False, True), (4, 2048, 256, False, True),
                            (262144, 64, 128, False, False), (1024, 1280, 256, False, True)]:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(k, n, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    ref = x.double() @ w.double() + b.double()
    if relu: ref = ref.relu()
    if res: ref = ref + r.double()
    got = ops.pwconv(x, ops.pack_pw_weights(w), b, k, n, relu=relu, residual=r)
    d = (got.double() - ref).abs().max().item()
    print(m, k, n, res, relu, "maxdiff", d, flush=True)
