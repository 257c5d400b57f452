#!/usr/bin/env python3
"""SR solver timing at the bench shape (N = 100 copies, 128^2 -> 512^2, 50 iterations): ms per solve and a checksum of the
result."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops, transforms as T
from asr_amd.superresolution_scripts import augmentation_utils as au
dev = torch.device("cuda")
n, H, h, iters = 100, 512, 128, 50
rng = np.random.RandomState(0)
y = rng.rand(1, n, h, h).astype(np.float32)
angles = rng.uniform(-0.15, 0.15, (1, n)).astype(np.float32); angles[0, 0] = 0
shifts = rng.uniform(-30, 30, (1, n, 2)).astype(np.float32); shifts[0, 0] = 0
rot = ops.to_device(T.rotation_transforms(angles.reshape(-1), H, H).reshape(1, n, 8))
irot = ops.to_device(T.rotation_transforms(-angles.reshape(-1), H, H).reshape(1, n, 8))
tr = ops.to_device(T.translation_transforms(shifts.reshape(-1, 2)).reshape(1, n, 8))
itr = ops.to_device(T.translation_transforms(-shifts.reshape(-1, 2)).reshape(1, n, 8))
yd = ops.to_device(y)
b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, it + 1)] for it in range(iters)], np.float32))
lam = (1.0, 0.3, 0.7, 0.0)
def solve():
    x = ops.sr_init_target(yd, (H, H))
    return ops.sr_solve(x, yd, rot, tr, irot, itr, alphas, lam, np.float32(1) - b1, np.float32(1) - b2, eps, True, want_loss=False)[0]
x = solve(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    x = solve()
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 5:.3f} ms per solve ({iters} iterations)  "
      f"sha1(x)={hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:16]}")
