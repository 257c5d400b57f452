import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import weights as W, ops
from asr_amd.model import DeeplabModel
from asr_amd.utils import load_image
from asr_amd.superresolution_scripts import augmentation_utils as au
from oracle.model import OracleDeeplabV3Plus
from oracle import augment as o_aug
w = W.make_synthetic_weights(1234)
size = (256, 256)
img_path = "tests/golden/test_cat.jpg"
np.random.seed(1234)
image = load_image(img_path, image_size=size)
o_img = o_aug.load_image(img_path, image_size=size)
print("image diff", np.abs(image - o_img).max())
angles, shifts = au.draw_augmentation_parameters(8, 0.15, 40)
copies = au.augment_on_device(ops.to_device(image), angles, shifts)
np.random.seed(1234)
o_copies, oa, os_ = o_aug.create_augmented_copies(o_img, 8, 0.15, 40)
print("copies diff", np.abs(copies.cpu().numpy() - o_copies).max())
m = DeeplabModel(w, size + (3,), 21, False, None)
o_pred = OracleDeeplabV3Plus(w).predict(o_copies, batch_size=4)
for bs in (8, 4, 2):
    p = m.predict_device(copies, batch_size=bs).cpu().numpy()
    print("bs", bs, "pred diff per copy", [float(np.abs(p[i] - o_pred[i]).max()) for i in range(8)])
p = m.predict_device(copies, batch_size=4)
cls, _ = au.output_processing(p, 8, "argmax")
o_masks, _ = o_aug.opm(o_pred, 8, "argmax")
om = np.stack(o_masks)[..., 0]
print("opm agree", (cls.cpu().numpy() == om).mean(), "frac class8 oracle", (om == 8).mean(), "prod", (cls.cpu().numpy() == 8).mean())
am = ops.argmax(p).cpu().numpy()
print("argmax kernel vs numpy on same preds", (am == p.cpu().numpy().argmax(-1)).mean(), "vs oracle", (am == o_pred.argmax(-1)).mean())
