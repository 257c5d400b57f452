#!/usr/bin/env python3
"""Forward-pass time per copy against the forward batch size (100 copies of one image is the reference's unit, but the GEMM
grid of a batch of B copies at 32 x 32 is 4 B row blocks x ceil(N / 256) column tiles: for B = 64 / 128 every launch is a whole
number of rounds of 256 workgroups, for B = 100 the 728-wide layers run 4.69 rounds and pay for 5).

    python tools/ab_batch.py [batch sizes ...]       # default 50 64 100 128
"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from asr_amd import weights as W
from asr_amd.model import DeeplabModel

sizes = [int(a) for a in sys.argv[1:]] or [50, 64, 100, 128]
dev = torch.device("cuda")
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (512, 512, 3), 21, final_upsample=False, last_activation=None)
torch.manual_seed(1)
xs = {b: torch.rand((b, 512, 512, 3), device=dev) for b in sizes}
for b in sizes:
    model.engine.forward(xs[b])
torch.cuda.synchronize()
tot = {b: {} for b in sizes}
wall = {b: 0.0 for b in sizes}
ROUNDS = 4
for r in range(ROUNDS):                  # interleaved: the chip's clock drifts with its thermal state
    for b in sizes:
        prof = {}
        model.engine.forward(xs[b], profile=prof)
        for k, v in prof.items():
            if not k.startswith("_"):
                tot[b][k] = tot[b].get(k, 0.0) + v[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.engine.forward(xs[b])
        e1.record()
        torch.cuda.synchronize()
        wall[b] += e0.elapsed_time(e1)
kinds = sorted({k for b in sizes for k in tot[b]})
print("ms per 100 copies (mean of %d interleaved rounds)" % ROUNDS)
print("batch  " + "  ".join(f"{k:>8}" for k in kinds) + "    sum   unprofiled pass")
for b in sizes:
    row = [tot[b].get(k, 0.0) / ROUNDS * 100.0 / b for k in kinds]
    print(f"{b:5d}  " + "  ".join(f"{v:8.3f}" for v in row) + f"  {sum(row):7.3f}  {wall[b] / ROUNDS * 100.0 / b:7.3f}", flush=True)
