import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from asr_amd import _lib, ops
lib = _lib.load()
old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasr_hip_old.so"))
name = "asr_pwconv_mfma_f16x3_presplit"
f_old = getattr(old, name); f_old.restype, f_old.argtypes = _lib.SIGNATURES[name]
f_new = getattr(lib, name)
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
tot = np.zeros(2)
shapes = [(102400, 728, 728, False, 0, 32), (102400, 728, 728, True, 0, 16), (409600, 728, 728, False, 1, 1), (102400, 1536, 2048, False, 1, 1),
          (102400, 1536, 1536, False, 1, 1), (1638400, 256, 256, False, 1, 2), (102400, 2048, 256, False, 1, 3), (1638400, 128, 256, False, 0, 1),
          (100000, 728, 728, False, 2, 0), (70000, 1024, 1024, False, 0, 0), (102400, 1024, 1536, False, 0, 1), (409600, 256, 728, False, 0, 1)]
if len(sys.argv) > 1: shapes = [(102400, 728, 728, True, 0, 16), (102400, 728, 728, True, 0, 16), (102400, 728, 728, True, 1, 16), (102400, 1024, 1024, True, 0, 1), (102400, 728, 728, False, 0, 32)]
for (M, K, N, res, relu, count) in shapes:
    chunks = (K + 31) // 32
    x = (torch.randn((M, chunks * 64), device=dev) * 0.5).to(torch.float16).view(torch.float32)
    w = ops.pack_pw_weights_f16x3(torch.randn((K, N), device=dev) * 0.05)
    bias = torch.randn(N, device=dev)
    ld = -(-N // 32) * 32
    r = torch.randn((M, ld), device=dev) if res else None
    outs = [torch.zeros((M, ld), device=dev) for _ in range(2)]
    times = [[], []]
    for rr in range(14):
        for v, fn in enumerate((f_old, f_new)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = fn(ops.ptr(x), ops.ptr(w), ops.ptr(bias), ops.ptr(r) if res else None, ops.ptr(outs[v]), M, K, N, chunks, ld, ld if res else 0, relu, st())
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rr: times[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    t = [np.median(times[0]), np.median(times[1])]
    tot += np.array(t) * count
    print(f"M={M} K={K} N={N} res={int(res)} relu={relu}: old {t[0]:8.1f} us  new {t[1]:8.1f} us  x{t[1] / t[0]:.4f}  bit-identical {same}", flush=True)
print(f"launch-weighted forward-pass mix: old {tot[0] / 1e3:.3f} ms  new {tot[1] / 1e3:.3f} ms  x{tot[1] / tot[0]:.4f}")
