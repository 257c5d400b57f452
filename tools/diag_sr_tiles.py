#!/usr/bin/env python3
"""Bit-identity of the tile (plane-free) SR backward against the plane form over awkward shapes: ragged tiles, large angles
(windows near the LDS cap), f = 2 / 4 / 6, shifts that push windows off the image, a projective copy (direct path)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, ops, transforms as T
b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
lam = (1.0, 0.3, 0.7, 0.05)
bad = 0
for case, (b, n, H, h, amax, smax, proj) in enumerate([(1, 100, 512, 256, 0.15, 30, False), (2, 37, 100, 50, 0.15, 10, False),
                                                        (1, 20, 128, 32, 0.8, 60, False), (2, 9, 96, 16, 3.1, 40, False),
                                                        (1, 12, 72, 12, 0.3, 20, False), (1, 16, 128, 64, 0.2, 200, False),
                                                        (1, 10, 128, 64, 0.2, 20, True)]):
    rng = np.random.RandomState(case)
    y = rng.rand(b, n, h, h).astype(np.float32)
    angles = rng.uniform(-amax, amax, (b, n)).astype(np.float32); angles[:, 0] = 0
    shifts = rng.uniform(-smax, smax, (b, n, 2)).astype(np.float32); shifts[:, 0] = 0
    tfs = [T.rotation_transforms(angles.reshape(-1), H, H), T.rotation_transforms(-angles.reshape(-1), H, H),
           T.translation_transforms(shifts.reshape(-1, 2)), T.translation_transforms(-shifts.reshape(-1, 2))]
    if proj:
        tfs[1] = tfs[1].copy(); tfs[1][3, 6] = 1e-4; tfs[1][5, 7] = -2e-4      # two projective inverse rotations
        tfs[3] = tfs[3].copy(); tfs[3][7, 0] = 1.01                            # one inverse translate that is not a pure translation
    rot, irot, tr, itr = [ops.to_device(np.asarray(t, np.float32).reshape(b, n, 8)) for t in tfs]
    iters = 3
    alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, it + 1)] * b for it in range(iters)], np.float32))
    yd = ops.to_device(y)
    outs = {}
    for chunk in (n, -1, -2):
        cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, eps, plane_chunk=chunk)
        st = {}
        x, _ = ops.sr_solve(ops.sr_init_target(yd, (H, H)), yd, rot, tr, irot, itr, alphas, lam, want_loss=False, cfg=cfg, state=st)
        outs[chunk] = (x, st["m"], st["v"])
    for chunk in (-1, -2):
        same = all(torch.equal(a, r) for a, r in zip(outs[chunk], outs[n]))
        d = (outs[chunk][0] - outs[n][0]).abs().max().item()
        print(f"case {case} b={b} n={n} H={H} h={h} amax={amax} smax={smax} proj={proj} tiles={32 if chunk == -1 else 16}: "
              f"{'bit-identical' if same else 'DIFFERENT'} (max |dx| {d:.3g})")
        bad += not same
sys.exit(1 if bad else 0)
