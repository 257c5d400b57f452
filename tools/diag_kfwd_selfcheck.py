#!/usr/bin/env python3
"""Which of K_fwd's packed-f32 sequences goes wrong beside co-resident MFMA waves?  

Library libasr_hz_pk_check.so (tools/build_hazard_variants.py): the SR solver's forward kernel compiled WITH packed-f32 and with a
self-check that re-derives every intermediate of its fast path with unpacked arithmetic written out in asm -- the coordinate map
(ix, iy), the bilinear sample, the translate blend, the final D -- and counts where the compiler's code disagrees.  Solves run
quietly first (all counters must be zero), then beside the synthetic MFMA aggressor of tools/hazard_aggressors.hip.

    python tools/diag_kfwd_selfcheck.py [trials]       (builds the variant libraries on first use)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deeplabv3plus-augmented-superresolution_amd")
LIBP = os.path.join(PKG, "libasr_hz_pk_check.so")
if not os.path.exists(LIBP):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "build_hazard_variants.py")], stdout=subprocess.DEVNULL)
os.environ["ASR_LIB"] = LIBP
import numpy as np, torch  # noqa: E402
sys.path.insert(0, ROOT)
from asr_amd import _lib, ops, transforms as T  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lib = _lib.load()
read = lib.asr_diag_kfwd_read
read.restype, read.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
so = "/tmp/libhazard_aggressors.so"
subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "hazard_aggressors.hip"), "-o", so],
                      stderr=subprocess.DEVNULL)
aggr = C.CDLL(so).hazard_aggressor
aggr.restype, aggr.argtypes = C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]

b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
rng = np.random.RandomState(3)
n, H, h = 100, 512, 128
y = ops.to_device((rng.rand(1, n, h, h) > 0.6).astype(np.float32))
angles = rng.uniform(-0.15, 0.15, (1, n)).astype(np.float32); shifts = rng.uniform(-80, 80, (1, n, 2)).astype(np.float32)
tf = lambda a: ops.to_device(a.reshape(1, n, 8))
rot, irot = tf(T.rotation_transforms(angles.reshape(-1), H, H)), tf(T.rotation_transforms(-angles.reshape(-1), H, H))
tr, itr = tf(T.translation_transforms(shifts.reshape(-1, 2))), tf(T.translation_transforms(-shifts.reshape(-1, 2)))
alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, 1)]], np.float32))
cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, eps)
torch.cuda.synchronize()
STAGES = ["map ix", "map iy", "bilinear sample", "translate blend", "final D"]


def counters():
    cnt = np.zeros(128, np.uint32)
    rec = np.zeros((16, 12), np.float32)
    assert read(cnt.ctypes.data, rec.ctypes.data) == 0
    return cnt, rec


def solve():
    """One iteration; returns the residuals K_fwd wrote (first n * h * h floats of the solver's workspace)."""
    st = {}
    x0 = ops.sr_init_target(y, (H, H))
    ops.sr_solve(x0, y, rot, tr, irot, itr, alphas, (1.0, 0.3, 0.7, 0.0), want_loss=False, cfg=cfg, state=st)
    return st["ws"][:n * h * h]


counters()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sb):
    solve()
    quiet = solve().clone()
torch.cuda.synchronize()
cnt, _ = counters()
print("quiet: mismatches per stage", {s: int(cnt[i]) for i, s in enumerate(STAGES)}, flush=True)
total = np.zeros(128, np.int64)
recs = []
for t in range(trials):
    with torch.cuda.stream(sa):
        for _ in range(2):
            assert aggr(1, 200, 200000, _lib.stream_ptr()) == 0            # bare v_mfma_f32_16x16x32_f16 loop, 200 registers per wave
    with torch.cuda.stream(sb):
        resid = solve()
    torch.cuda.synchronize()
    moved = int((resid != quiet).sum())
    cnt, rec = counters()
    total += cnt
    k = int(min(cnt[7], 16))
    recs += [rec[i].copy() for i in range(k)]
    print(f"trial {t}: residuals that differ from the quiet solve {moved} | self-check: " + ", ".join(f"{s} {int(cnt[i])}" for i, s in enumerate(STAGES)), flush=True)
print("beside the MFMA aggressor, all trials:", {s: int(total[i]) for i, s in enumerate(STAGES)})
print("lanes:", {l: int(total[8 + l]) for l in range(64) if total[8 + l]})
print("rotation sample (3a + c) of the map mismatches:", {k: int(total[80 + k]) for k in range(9) if total[80 + k]})
print("first records (stage, lane | inputs | got, expected):")
for r in recs[:24]:
    st = int(r[0])
    print(f"  {STAGES[st]:16s} lane {int(r[1]):2d} copy {int(r[10]):3d} | " + " ".join(f"{v:.9g}" for v in r[2:8]) + f" | got {r[8]:.9g} expected {r[9]:.9g}")
