#!/usr/bin/env python3
"""Is any kernel of the FORWARD pass disturbed by the fused entry-flow kernels of the other lane?  

The two-lane pipeline runs two forward passes on two streams, so every launch of one lane can share CUs with
entry_stem_fused_kernel / sepconv_fused_kernel of the other.  Round 4 established that a wave executing packed-f32
instructions returns garbage in lanes 48-63 when it shares a SIMD with two waves of those kernels; this tool walks lane 0's
plan launch by launch and, for each launch, compares the output written quietly with the output written while lane 1's stem
(and fused separable convs) are replayed on another stream -- bit for bit.

    python tools/diag_forward_steps_under_stem.py [trials per launch, default 3] [copies, default 100]
    AGGR=synthetic:<mode>:<registers>:<iterations> python tools/diag_forward_steps_under_stem.py ...
        the aggressor is a synthetic one-ingredient kernel instead (tools/hazard_aggressors.hip; mode 1 = a bare
        v_mfma_f32_16x16x32_f16 loop, the strongest aggressor found: with 128 registers per wave it leaves room for a
        co-resident wave of up to 256 registers, i.e. for every kernel of the forward pass)
"""
import os
import sys
import torch
sys.path.insert(0, "/root/repo")
from asr_amd import _lib, weights as W
from asr_amd.model import DeeplabModel

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 3
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda")
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (512, 512, 3), 21, final_upsample=False, last_activation=None)
torch.manual_seed(1)
xin = torch.rand((copies, 512, 512, 3), device=dev)
model.engine.forward(xin, lane=0)
model.engine.forward(xin, lane=1)
torch.cuda.synchronize()
victim, aggr = model.engine.plan(copies, 512, 512, 0), model.engine.plan(copies, 512, 512, 1)
lib = _lib.load()
AGGR = ("asr_entry_stem_f16x3", "asr_sepconv_fused_f16x3")
aggr_steps = [s for s in aggr["steps"] if s[0] in AGGR]
print("aggressor launches:", [s[5] or s[0] for s in aggr_steps], flush=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


_syn = None
if os.environ.get("AGGR", "").startswith("synthetic:"):
    import ctypes
    import subprocess
    so = "/tmp/libhazard_aggressors.so"
    subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC",
                           os.path.join(os.path.dirname(os.path.abspath(__file__)), "hazard_aggressors.hip"), "-o", so],
                          stderr=subprocess.DEVNULL)
    _fn = ctypes.CDLL(so).hazard_aggressor
    _fn.restype, _fn.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    _m, _mode, _regs, _iters = os.environ["AGGR"].split(":")
    _syn = (int(_mode), int(_regs), int(_iters))
    print("synthetic aggressor (mode, registers, iterations):", _syn, flush=True)


def launch_aggressors(stream):
    if _syn is None:
        for a in aggr_steps:
            launch(a, stream)
        return
    with torch.cuda.stream(stream):
        rc = _fn(_syn[0], _syn[1], _syn[2], _lib.stream_ptr())
    assert rc == 0, rc


def launch(step, stream):
    with torch.cuda.stream(stream):
        rc = getattr(lib, step[0])(*step[1], _lib.stream_ptr())
    assert rc == 0, step[0]


victim["x_in"].t.copy_(xin.reshape(-1))
torch.cuda.synchronize()
moved_total, skipped = 0, 0
for idx, (step, out) in enumerate(zip(victim["steps"], victim["outs"])):
    name, label = step[0], step[5] or step[0]
    launch(step, sb)                       # inputs are valid: every earlier launch of the plan has run, in order
    torch.cuda.synchronize()
    if out is None:
        continue
    out = out[0] if isinstance(out, (list, tuple)) else out
    quiet = out.t.clone()
    launch(step, sb)
    torch.cuda.synchronize()
    if not torch.equal(out.t, quiet):      # writes into one of its own inputs (in-place residual): not re-runnable
        skipped += 1
        print(f"[{idx:3d}] {name} {label}: not idempotent (in place), skipped", flush=True)
        # restore what the plan expects downstream: nothing to do -- the second run's result stands in for the first; the
        # remaining launches are compared with THEIR quiet runs on whatever inputs they now have
        continue
    moved = 0
    worst = 0.0
    for t in range(trials):
        launch_aggressors(sa)
        launch(step, sb)
        torch.cuda.synchronize()
        if not torch.equal(out.t, quiet):
            moved += 1
            d = (out.t.float() - quiet.float()).abs()
            worst = max(worst, float(d[torch.isfinite(d)].max()) if torch.isfinite(d).any() else float("inf"))
    moved_total += moved
    flag = "  <-- MOVED" if moved else ""
    print(f"[{idx:3d}] {name} {label}: {moved} of {trials} moved" + (f", max |d| {worst:.3g}" if moved else "") + flag, flush=True)
print(f"forward launches disturbed: {moved_total} (of {len(victim['steps'])} launches x {trials} trials; {skipped} skipped)", flush=True)
