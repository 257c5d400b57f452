import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import weights as W, ops
from asr_amd.model import DeeplabModel
from asr_amd.utils import load_image
from asr_amd.superresolution_scripts import augmentation_utils as au
w = W.make_synthetic_weights(1234)
size = (256, 256)
np.random.seed(1234)
image = load_image("tests/golden/test_cat.jpg", image_size=size)
m = DeeplabModel(w, size + (3,), 21, False, None)
angles, shifts = au.draw_augmentation_parameters(8, 0.15, 40)
copies = au.augment_on_device(ops.to_device(image), angles, shifts)
c4 = m.engine.forward_capture(copies[0:4].contiguous())
c4b = m.engine.forward_capture(copies[0:4].contiguous())
c8 = m.engine.forward_capture(copies)
n = 0
for (l4, t4), (_, t4b), (l8, t8) in zip(c4, c4b, c8):
    d = (t4 - t8[:4]).abs().max().item()
    dd = (t4 - t4b).abs().max().item()
    if d > 1e-4 or dd > 0:
        print("MISMATCH", l4, "| vs B=8:", d, "| run-to-run:", dd, flush=True)
        n += 1
        if n > 6: break
print("done", len(c4))
plan = m.engine.plan(4, 256, 256)
lib = __import__("asr_amd")._lib.load()
from asr_amd import _lib
idx = [i for i, st in enumerate(plan["steps"]) if st[5] == "gap"][0]
for j in (idx - 1, idx, idx + 1, idx + 2):
    st = plan["steps"][j]
    print(j, st[0], st[5], [hex(a) if isinstance(a, int) and a > 1 << 20 else a for a in st[1]])
import ctypes
def run(sync_after=()):
    plan["x_in"].t.copy_(copies[0:4].contiguous().reshape(-1))
    s = _lib.stream_ptr()
    for j, st in enumerate(plan["steps"]):
        _lib.check(getattr(lib, st[0])(*st[1], s), st[0])
        if j in sync_after:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return plan["outs"][idx + 1].t.clone(), plan["outs"][idx].t.clone(), plan["outs"][idx - 1].t.clone()
def poison():
    plan["outs"][idx].t.fill_(1e3); plan["outs"][idx + 1].t.fill_(-5.0); torch.cuda.synchronize()
poison(); pp_ref, pooled_ref, x_ref = run(sync_after=range(len(plan["steps"])))
print("ref pooled vs torch mean", (pooled_ref.view(4, 2048) - x_ref.view(4, 256, 2048).mean(1)).abs().max().item())
for name, sa in (("none", ()), ("after_gap", (idx,)), ("before_gap", (idx - 1,)), ("none2", ())):
    poison(); pp, pooled, x = run(sync_after=sa)
    print(name, "pp diff", (pp - pp_ref).abs().max().item(), "pooled diff", (pooled - pooled_ref).abs().max().item(), "x diff", (x - x_ref).abs().max().item(), flush=True)
l8 = m.engine.forward(copies).clone()
for rep in range(4):
    l4 = m.engine.forward(copies[0:4].contiguous()).clone()
    l4b = m.engine.forward(copies[4:8].contiguous()).clone()
    torch.cuda.synchronize()
    print("rep", rep, "first half diff", (l4 - l8[:4]).abs().max().item(), "second half diff", (l4b - l8[4:]).abs().max().item(), flush=True)
