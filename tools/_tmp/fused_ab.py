import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from asr_amd import _lib, ops
lib = _lib.load()
old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasr_hip_old.so"))
def pair(name):
    f = getattr(old, name); f.restype, f.argtypes = _lib.SIGNATURES[name]; return f, getattr(lib, name)
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
def ab(label, call, outs):
    times = [[], []]
    for rr in range(7):
        for v in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): assert call(v) == 0
            e1.record(); torch.cuda.synchronize()
            if rr: times[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    t = [np.median(times[0]), np.median(times[1])]
    print(f"{label}: old {t[0]:8.1f} us  new {t[1]:8.1f} us  x{t[1] / t[0]:.4f}  bit-identical {torch.equal(outs[0], outs[1])}")
B = 100
# fused separable convs of entry-flow block 1
fs = pair("asr_sepconv_fused_f16x3")
for cin in (64, 128):
    x = torch.randn((B, 256, 256, cin), device=dev) * 2
    x[0, 0, 0, :4] = torch.tensor([7e4, -9e4, 65520., 3e5], device=dev)
    wd = torch.randn((3, 3, cin), device=dev) * 0.3; bd = torch.randn(cin, device=dev)
    wp = ops.pack_pw_weights_f16x3(torch.randn((cin, 128), device=dev) * 0.05); bp = torch.randn(128, device=dev)
    outs = [torch.zeros((B, 256, 256, 128), device=dev) for _ in range(2)]
    ab(f"sepconv fused {cin}->128", lambda v: fs[v](ops.ptr(x), ops.ptr(wd), ops.ptr(bd), ops.ptr(wp), ops.ptr(bp), ops.ptr(outs[v]), B, 256, 256, cin, 128, cin, 128, 1, 0, 0, st()), outs)
# fused stem
fe = pair("asr_entry_stem_f16x3")
x = torch.rand((B, 512, 512, 3), device=dev) * 2 - 1
w1 = torch.randn((3, 3, 3, 32), device=dev) * 0.3; b1 = torch.randn(32, device=dev)
w2 = ops.pack_pw_weights_f16x3(torch.randn((288, 64), device=dev) * 0.05); b2 = torch.randn(64, device=dev)
outs = [torch.zeros((B, 256, 256, 64), device=dev) for _ in range(2)]
ab("entry stem fused", lambda v: fe[v](ops.ptr(x), ops.ptr(w1), ops.ptr(b1), ops.ptr(w2), ops.ptr(b2), ops.ptr(outs[v]), B, 512, 512, 3, 64, st()), outs)
