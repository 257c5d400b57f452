import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from asr_amd import _lib, ops
lib = _lib.load()
old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasr_hip_old.so"))
name = "asr_pwconv_mfma_f16x3"
f_old = getattr(old, name); f_old.restype, f_old.argtypes = _lib.SIGNATURES[name]
f_new = getattr(lib, name)
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
tot = np.zeros(2)
for (M, K, N, count) in [(102400, 728, 1024, 1), (102400, 2048, 256, 1), (102400, 1280, 256, 1), (1638400, 128, 128, 1), (1638400, 64, 128, 1),
                          (409600, 128, 256, 1), (102400, 256, 728, 1)]:
    ldx = -(-K // 32) * 32
    x = torch.randn((M, ldx), device=dev) * 3
    x[0, :8] = torch.tensor([7e4, -9e4, 65504., 65520., -65519., 1e-9, 3e5, -0.0], device=dev)
    w = ops.pack_pw_weights_f16x3(torch.randn((K, N), device=dev) * 0.05)
    bias = torch.randn(N, device=dev)
    ld = -(-N // 32) * 32
    outs = [torch.zeros((M, ld), device=dev) for _ in range(2)]
    times = [[], []]
    for rr in range(8):
        for v, fn in enumerate((f_old, f_new)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = fn(ops.ptr(x), ops.ptr(w), ops.ptr(bias), None, ops.ptr(outs[v]), M, K, N, ldx, ld, 0, 1, 1, 0, 0, st())
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rr: times[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    same = torch.equal(outs[0], outs[1])
    t = [np.median(times[0]), np.median(times[1])]
    tot += np.array(t) * count
    print(f"M={M} K={K} N={N}: old {t[0]:8.1f} us  new {t[1]:8.1f} us  x{t[1] / t[0]:.4f}  bit-identical {same}  finite {bool(torch.isfinite(outs[1]).all())}")
print(f"sum: old {tot[0] / 1e3:.3f} ms  new {tot[1] / 1e3:.3f} ms  x{tot[1] / tot[0]:.4f}")
