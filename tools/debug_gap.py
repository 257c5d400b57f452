import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda")
for shape in ((4, 16, 16, 2048), (8, 16, 16, 2048), (4, 32, 32, 2048), (2, 4, 6, 2048)):
    x = torch.randn(*shape, device=dev)
    ref = x.double().mean(dim=(1, 2))
    worst = 0
    for rep in range(10):
        got = ops.gap(x)
        worst = max(worst, (got.double() - ref).abs().max().item())
    print(shape, "worst", worst, flush=True)
