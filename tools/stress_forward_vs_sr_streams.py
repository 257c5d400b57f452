#!/usr/bin/env python3
"""Reproducibility under stream concurrency: forward passes (100 copies, 512 x 512) on one HIP stream, SR solves on a second,
each compared bit for bit with its first result.  Round 3: the forward passes always stayed identical; 15-20 %
of the SOLVES differed while sr.hip was compiled with packed-f32 instructions (K_fwd next to entry_stem_fused_kernel:
tools/diag_sr_stages_under_stem.py), 0 of 382 x 8 since csrc/build.py compiles the second-lane kernels without them.

    python tools/stress_forward_vs_sr_streams.py
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from asr_amd import _lib, ops, transforms as T, weights as W
from asr_amd.model import DeeplabModel
dev = torch.device("cuda")
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (512, 512, 3), 21, final_upsample=False, last_activation=None)
torch.manual_seed(1)
xin = torch.rand((100, 512, 512, 3), device=dev)
b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
rng = np.random.RandomState(3)
n, H, h, iters = 100, 512, 128, 50
y = ops.to_device((rng.rand(1, n, h, h) > 0.6).astype(np.float32))
angles = rng.uniform(-0.15, 0.15, (1, n)).astype(np.float32); shifts = rng.uniform(-80, 80, (1, n, 2)).astype(np.float32)
tf = lambda a: ops.to_device(a.reshape(1, n, 8))
rot, irot = tf(T.rotation_transforms(angles.reshape(-1), H, H)), tf(T.rotation_transforms(-angles.reshape(-1), H, H))
tr, itr = tf(T.translation_transforms(shifts.reshape(-1, 2))), tf(T.translation_transforms(-shifts.reshape(-1, 2)))
alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, it + 1)] for it in range(iters)], np.float32))
cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, eps)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref_f = ref_s = None
bad_f = bad_s = 0; it = 0; t0 = time.time()
while time.time() - t0 < 30:
    with torch.cuda.stream(sa):
        lo = model.engine.forward(xin, lane=0).clone()
    souts = []
    with torch.cuda.stream(sb):
        for _ in range(8):                       # ~8 solves fit under one forward pass
            x0 = ops.sr_init_target(y, (H, H))
            souts.append(ops.sr_solve(x0, y, rot, tr, irot, itr, alphas, (1.0, 0.3, 0.7, 0.0), want_loss=False, cfg=cfg)[0])
    torch.cuda.synchronize()
    if ref_f is None: ref_f, ref_s = lo.clone(), souts[0].clone()
    if not torch.equal(lo, ref_f):
        bad_f += 1
        if bad_f <= 3: print(f"iter {it}: forward differs in {int((lo != ref_f).sum())} logits, max |d| {float((lo - ref_f).abs().max()):.3g}", flush=True)
    for k, o in enumerate(souts):
        if not torch.equal(o, ref_s):
            bad_s += 1
            if bad_s <= 3: print(f"iter {it} solve {k}: SR differs in {int((o != ref_s).sum())} pixels, max |d| {float((o - ref_s).abs().max()):.3g}", flush=True)
    it += 1
print(f"{it} rounds: forward mismatches {bad_f}, SR mismatches {bad_s}", flush=True)
