import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import weights as W, ops
from asr_amd.model import DeeplabModel
from asr_amd.utils import load_image
from asr_amd.superresolution_scripts import augmentation_utils as au
w = W.make_synthetic_weights(1234)
size = (256, 256)
order = sys.argv[1]
if order == "model_first":
    m = DeeplabModel(w, size + (3,), 21, False, None)
np.random.seed(1234)
image = load_image("tests/golden/test_cat.jpg", image_size=size)
if order != "model_first":
    m = DeeplabModel(w, size + (3,), 21, False, None)
angles, shifts = au.draw_augmentation_parameters(8, 0.15, 40)
copies = au.augment_on_device(ops.to_device(image), angles, shifts)
for bs in (4, 8, 4):
    p = m.predict_device(copies, batch_size=bs)
    print(order, "bs", bs, "pred mean", float(p.mean()), flush=True)
print("stream ptr", torch.cuda.current_stream().cuda_stream, torch.cuda.default_stream().cuda_stream)
outs = []
for i in (0, 4):
    torch.cuda.synchronize()
    o = m.engine.forward(copies[i:i + 4].contiguous()).clone()
    torch.cuda.synchronize()
    outs.append(o)
print("synced bs4 mean", float(torch.cat(outs).mean()))
o1 = m.engine.forward(copies[0:4].contiguous()).clone(); torch.cuda.synchronize()
o2 = m.engine.forward(copies[4:8].contiguous()).clone(); torch.cuda.synchronize()
print("halves", float(o1.mean()), float(o2.mean()))
p8 = m.predict_device(copies, batch_size=8)
print("halves from bs8", float(p8[:4].mean()), float(p8[4:].mean()))
