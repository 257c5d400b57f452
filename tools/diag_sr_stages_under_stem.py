#!/usr/bin/env python3
"""Which stage of an SR iteration moves when entry_stem_fused_kernel runs on another stream?
A ONE-iteration solve writes every intermediate into the caller's workspace -- resid (K_fwd), the G_R planes (K_gt), the
bordered copy of the new x and x itself (K_bwd) -- so a quiet solve and solves overlapped with replays of the stem launch
can be compared stage by stage, bit for bit.

    python tools/diag_sr_stages_under_stem.py [trials] [all]     # "all": + the sentinel and prefilled-workspace phases
    ASR_LIB=<variant library> python tools/diag_sr_stages_under_stem.py 12     # a variant of sr.hip linked against the product objects

Round 3 (profiles/r03_sr_next_to_stem_diagnosis.txt): with packed-f32 instructions in sr.hip the residuals of K_fwd were
garbage in lanes 48-63 of ~130 of its 25 600 waves, 44 of 44 trials, whatever the form of the loads; without them 0 of 32.
csrc/build.py builds sr.hip / warp.hip / reduce.hip without them; this tool now reports "0 of N moved".
"""
import os
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from asr_amd import _lib, ops, transforms as T, weights as W
from asr_amd.model import DeeplabModel

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
model = DeeplabModel(W.make_synthetic_weights(1234, 21), (512, 512, 3), 21, final_upsample=False, last_activation=None)
torch.manual_seed(1)
xin = torch.rand((100, 512, 512, 3), device=dev)
model.engine.forward(xin, lane=0)
torch.cuda.synchronize()
plan = model.engine.plan(100, 512, 512, 0)
lib = _lib.load()
kinds = sorted({s[2] for s in plan["steps"]})
print("plan kinds:", kinds, flush=True)

b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
rng = np.random.RandomState(3)
n, H, h = 100, 512, 128
Wd = H
y = ops.to_device((rng.rand(1, n, h, h) > 0.6).astype(np.float32))
angles = rng.uniform(-0.15, 0.15, (1, n)).astype(np.float32); shifts = rng.uniform(-80, 80, (1, n, 2)).astype(np.float32)
tf = lambda a: ops.to_device(a.reshape(1, n, 8))
rot, irot = tf(T.rotation_transforms(angles.reshape(-1), H, H)), tf(T.rotation_transforms(-angles.reshape(-1), H, H))
tr, itr = tf(T.translation_transforms(shifts.reshape(-1, 2))), tf(T.translation_transforms(-shifts.reshape(-1, 2)))
alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, 1)]], np.float32))
cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, eps)
torch.cuda.synchronize()

PH, PW = H + 4, Wd + 64
pe = PH * PW


def stages(x, ws):
    """Views of the workspace of asr_sr_solve_cfg_f32 (sr.hip: resid | x_alt | acc | planes | bordered x | flags)."""
    o = 0
    resid = ws[o:o + n * h * h]; o += n * h * h
    o += 2 * H * Wd
    chunk = (ws.numel() - 1 - o - pe) // pe
    gr = ws[o:o + chunk * pe].view(chunk, PH, PW); o += chunk * pe
    xb = ws[o:o + pe].view(PH, PW); o += pe
    flags = ws[o:o + 1].view(torch.int32)
    return {"resid": resid.view(n, h, h), "planes": gr, "x_bordered": xb, "x": x.view(H, Wd), "flags": flags}, chunk


def one_solve():
    st = {}
    x0 = ops.sr_init_target(y, (H, H))
    ops.sr_solve(x0, y, rot, tr, irot, itr, alphas, (1.0, 0.3, 0.7, 0.0), want_loss=False, cfg=cfg, state=st)
    return x0, st["ws"]


_synthetic = None


def replay(kind, times):
    """kind: a plan kind ("conv" = the stem launch), "name:<entry point>" (every launch of that entry point), or
    "synthetic:<mode>:<registers>:<iterations>" (tools/hazard_aggressors.hip, built into /tmp on first use)."""
    global _synthetic
    s = _lib.stream_ptr()
    if kind.startswith("synthetic:"):
        if _synthetic is None:
            import ctypes, subprocess
            so = "/tmp/libhazard_aggressors.so"
            subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC",
                                   os.path.join(os.path.dirname(os.path.abspath(__file__)), "hazard_aggressors.hip"), "-o", so],
                                  stderr=subprocess.DEVNULL)
            _synthetic = ctypes.CDLL(so).hazard_aggressor
            _synthetic.restype, _synthetic.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _m, mode, regs, iters = kind.split(":")
        for _ in range(times):
            rc = _synthetic(int(mode), int(regs), int(iters), s)
            assert rc == 0, rc
        return
    for _ in range(times):
        for name, args, k, *_ in plan["steps"]:
            if k == kind or kind == "name:" + name:
                rc = getattr(lib, name)(*args, s)
                assert rc == 0, name


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sb):
    xq, wsq = one_solve()
torch.cuda.synchronize()
quiet, chunk = stages(xq, wsq)
quiet = {k: v.clone() for k, v in quiet.items()}
print(f"chunk {chunk} planes of {PH} x {PW}; flags {int(quiet['flags'][0])}", flush=True)
with torch.cuda.stream(sb):
    x2, ws2 = one_solve()
torch.cuda.synchronize()
again, _ = stages(x2, ws2)
print("second quiet solve identical:", {k: bool(torch.equal(v, quiet[k])) for k, v in again.items()}, flush=True)


def describe(name, a, q):
    d = a != q
    cnt = int(d.sum())
    if cnt == 0:
        return f"{name}: same"
    idx = d.nonzero()
    lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
    mx = float((a.float() - q.float()).abs().max())
    first = tuple(idx[0].tolist())
    extra = ""
    if a.dim() == 3:
        extra = f", {int(d.flatten(1).any(1).sum())} of {a.shape[0]} slices"
    return (f"{name}: {cnt} differ{extra}, index box {lo}..{hi}, max |d| {mx:.3g}, first {first}: "
            f"{float(a[first]):.9g} vs quiet {float(q[first]):.9g}")


# DIAG_REPLAY="conv:6" (default) | "name:asr_sepconv_fused_f16x3:3" | "name:asr_pwconv_mfma_f16x3:1" ...: what runs on the other stream
_spec = os.environ.get("DIAG_REPLAY", "conv:6").rsplit(":", 1)      # "<kind>:<replays>"
print("library:", os.environ.get("ASR_LIB", "(product)"), "| replay:", _spec, "| launches per replay:",
      sum(1 for st in plan["steps"] if st[2] == _spec[0] or _spec[0] == "name:" + st[0]), flush=True)
for kind, reps in ((_spec[0], int(_spec[1])),):
    moved = 0
    for t in range(trials):
        with torch.cuda.stream(sa):
            replay(kind, reps)
        with torch.cuda.stream(sb):
            xt, wst = one_solve()
        torch.cuda.synchronize()
        got, _ = stages(xt, wst)
        diff = [k for k, v in got.items() if not torch.equal(v, quiet[k])]
        if diff:
            moved += 1
            if moved <= 4:
                print(f"[{kind}] trial {t}: " + " | ".join(describe(k, got[k], quiet[k]) for k in got), flush=True)
            if moved == 1:      # the lane pattern: K_fwd's workgroup is 32 x 8 residual elements, one wave = 2 rows x 32 columns
                idx = (got["resid"] != quiet["resid"]).nonzero()
                lane = (idx[:, 1] % 2) * 32 + idx[:, 2] % 32
                print("  lanes hit (lane: count):", {int(l): int((lane == l).sum()) for l in lane.unique()}, flush=True)
                wave = idx[:, 0] * 1000000 + (idx[:, 1] // 2) * 1000 + idx[:, 2] // 32
                uw, cnt = wave.unique(return_counts=True)
                print(f"  {uw.numel()} waves hit; elements per wave: min {int(cnt.min())} max {int(cnt.max())}; "
                      f"histogram {torch.bincount(cnt).tolist()}", flush=True)
                for r in idx[:48].tolist():
                    print(f"    resid[{r[0]},{r[1]},{r[2]}] = {float(got['resid'][tuple(r)]):.9g}  (quiet {float(quiet['resid'][tuple(r)]):.9g})", flush=True)
    print(f"[{kind}] {moved} of {trials} one-iteration solves moved", flush=True)

# ---- does the stem launch STORE into memory it does not own?  The workspaces of the solves above, refilled with a bit
# pattern, must survive replays of the stem launch on the other stream (and on their own stream).
if len(sys.argv) <= 2:
    sys.exit(0)
keep = [wsq, ws2, wst]
print("workspace blocks:", [hex(t.data_ptr()) for t in keep], "numel", wsq.numel(), flush=True)
bufs = {s[5] or s[0]: None for s in plan["steps"] if s[2] == "conv"}
print("stem steps:", list(bufs), "x_in", hex(plan["x_in"].ptr), "pool bytes", plan["pool_bytes"], flush=True)
owned = sorted((t.data_ptr(), t.numel() * 4) for t in plan["pool"].owned)
print("plan buffers:", [(hex(a), n) for a, n in owned], flush=True)
PAT = 0x5A5A1234
for t in keep:
    t.view(torch.int32).fill_(PAT)
torch.cuda.synchronize()
for where, stream in (("other stream", sa), ("same stream", sb)):
    with torch.cuda.stream(stream):
        replay("conv", 6)
    torch.cuda.synchronize()
    for i, t in enumerate(keep):
        bad = (t.view(torch.int32) != PAT).nonzero().flatten()
        if bad.numel():
            print(f"sentinel [{where}] block {i}: {bad.numel()} words changed, offsets {int(bad[0])}..{int(bad[-1])}, "
                  f"first values {t[bad[:4]].tolist()}", flush=True)
        else:
            print(f"sentinel [{where}] block {i}: intact", flush=True)

# ---- is the garbage stale workspace content?  Prefill the workspace with 7777 and look at the magnitudes that show up.
moved = 0
for t in range(10):
    with torch.cuda.stream(sb):
        x0 = ops.sr_init_target(y, (H, H))
        st = {"ws": torch.full((wsq.numel(),), 7777.0, device=dev), "m": torch.zeros_like(x0), "v": torch.zeros_like(x0),
              "vhat": torch.zeros_like(x0)}
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        replay("conv", 6)
    with torch.cuda.stream(sb):
        ops.sr_solve(x0, y, rot, tr, irot, itr, alphas, (1.0, 0.3, 0.7, 0.0), want_loss=False, cfg=cfg, state=st)
    torch.cuda.synchronize()
    got, _ = stages(x0, st["ws"])
    if not torch.equal(got["resid"], quiet["resid"]):
        moved += 1
        if moved <= 3:
            print("[prefilled 7777] " + describe("resid", got["resid"], quiet["resid"]), flush=True)
print(f"[prefilled 7777] {moved} of 10 moved", flush=True)
