#!/usr/bin/env python3
"""SR solver timing at the bench shapes, interleaved over the gradient-plane chunking (asr_sr_config.plane_chunk):
  configs[1]: N = 100 copies, 128^2 -> 512^2 (f = 4), 50 iterations
  configs[4]: N = 200 copies, 256^2 -> 512^2 (f = 2), 50 iterations
ms per solve (median of interleaved rounds), ms per iteration, and a checksum of x (must not depend on the chunking).
    python tools/bench_sr.py [--lanes 2]     # --lanes 2: two solves in flight on two streams (the bench's two lanes)
"""
import argparse, hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, ops, transforms as T
ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=1)
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda")
b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
lam = (1.0, 0.3, 0.7, 0.0)


def problem(n, H, h, iters, seed=0):
    rng = np.random.RandomState(seed)
    y = rng.rand(1, n, h, h).astype(np.float32)
    angles = rng.uniform(-0.15, 0.15, (1, n)).astype(np.float32); angles[0, 0] = 0
    shifts = rng.uniform(-30, 30, (1, n, 2)).astype(np.float32); shifts[0, 0] = 0
    tf = lambda a: ops.to_device(a.reshape(1, n, 8))
    rot, irot = tf(T.rotation_transforms(angles.reshape(-1), H, H)), tf(T.rotation_transforms(-angles.reshape(-1), H, H))
    tr, itr = tf(T.translation_transforms(shifts.reshape(-1, 2))), tf(T.translation_transforms(-shifts.reshape(-1, 2)))
    alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, it + 1)] for it in range(iters)], np.float32))
    return ops.to_device(y), rot, tr, irot, itr, alphas


def solve(p, H, chunk):
    yd, rot, tr, irot, itr, alphas = p
    cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, eps, plane_chunk=chunk)
    x = ops.sr_init_target(yd, (H, H))
    return ops.sr_solve(x, yd, rot, tr, irot, itr, alphas, lam, want_loss=False, cfg=cfg)[0]


for name, n, H, h, iters in (("configs[1]", 100, 512, 128, 50), ("configs[4]", 200, 512, 256, 50)):
    probs = [problem(n, H, h, iters, seed=s) for s in range(args.lanes)]
    streams = [torch.cuda.Stream() for _ in range(args.lanes)]
    variants = [("all copies at once (round 2)", n), ("default (<= 32)", 0), ("16", 16), ("8", 8)]
    times = {v[0]: [] for v in variants}
    sums = {}
    for r in range(args.rounds + 1):
        for label, chunk in variants:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            outs = []
            for s, p in zip(streams, probs):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    outs.append(solve(p, H, chunk))
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
            e1.record(); torch.cuda.synchronize()
            if r:
                times[label].append(e0.elapsed_time(e1))
            sums[label] = hashlib.sha1(outs[0].cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{name}: N = {n}, {h}^2 -> {H}^2, {iters} iterations, {args.lanes} solve(s) in flight")
    for label, _ in variants:
        t = float(np.median(times[label]))
        print(f"  plane_chunk {label:30s} {t:8.3f} ms per solve  {1000 * t / iters:7.1f} us per iteration  sha1(x) {sums[label]}")
    assert len(set(sums.values())) == 1, "the result depends on the chunking"
