"""Generates the committed golden fixtures.  They are produced by the build's CPU restatement
(oracle/), NOT by TensorFlow: TF/TFA cannot be installed offline, so these vectors pin the oracle
against regressions and give the GPU tests fixed targets -- they do not pin parity with the
reference ("parity unpinned", see oracle/__init__.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import augment as o_aug, sr as o_sr, tf_ops  # noqa: E402
from oracle.model import OracleDeeplabV3Plus  # noqa: E402


def blobs(rng, n, h):
    yy, xx = np.mgrid[0:h, 0:h].astype(np.float32)
    out = np.zeros((n, h, h), np.float32)
    for i in range(n):
        cy, cx = h * (0.4 + 0.2 * rng.random()), h * (0.4 + 0.2 * rng.random())
        r = h * (0.2 + 0.1 * rng.random())
        out[i] = ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r)
    return out


def main():
    # 1. seed-1234 augmentation parameters (numpy legacy stream, reference draw order)
    np.random.seed(1234)
    a8, s8 = o_aug.draw_angles_shifts(8, 0.15, 80)
    np.random.seed(1234)
    a100, s100 = o_aug.draw_angles_shifts(100, 0.15, 80)
    np.savez(os.path.join(HERE, "rng_1234.npz"), angles8=a8, shifts8=s8, angles100=a100, shifts100=s100)

    # 2. 64x64 warp in/out (rotate then translate)
    rng = np.random.default_rng(1)
    img = rng.random((64, 64, 3), dtype=np.float32)
    ang = np.array([0.0, 0.12, -0.3, 0.7], np.float32)
    sh = np.array([[0, 0], [5.5, -3.25], [-10, 4], [2, 2]], np.float32)
    tiled = torch.from_numpy(np.broadcast_to(img[None], (4, 64, 64, 3)).copy())
    out = tf_ops.translate(tf_ops.rotate(tiled, ang), sh).numpy()
    np.savez_compressed(os.path.join(HERE, "warp_64.npz"), image=img, angles=ang, shifts=sh,
                        out=out)

    # 3. 10-iteration SR trajectory on a 32 -> 128 toy (shipped hyper-parameters)
    rng = np.random.default_rng(2)
    y = blobs(rng, 5, 32)
    ang = rng.uniform(-0.15, 0.15, 5).astype(np.float32)
    sh = rng.uniform(-20, 20, (5, 2)).astype(np.float32)
    ang[0] = 0
    sh[0] = 0
    opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=10, num_aug=5, optimizer=opt, feature_size=(32, 32),
                              output_size=(128, 128))
    x, loss = sr.augmented_superresolution(y[..., None], ang, sh)
    mx, _ = sr.max_superresolution(y[..., None], ang, sh)
    mn, _ = sr.mean_superresolution(y[..., None], ang, sh)
    np.savez_compressed(os.path.join(HERE, "sr_toy_32_128.npz"), y=y.astype(np.uint8), angles=ang, shifts=sh,
                        x10=x[..., 0], loss=np.float64(loss), max_sr=mx[..., 0], mean_sr=mn[..., 0])

    # 4. model logits on a 32x32 input, synthetic weights seed 1234 (regenerated from the seed by the tests)
    from asr_amd import weights as W
    w = W.make_synthetic_weights(1234, 21)
    xin = np.random.default_rng(3).random((1, 32, 32, 3), dtype=np.float32)
    logits = OracleDeeplabV3Plus(w).forward(xin)
    np.savez_compressed(os.path.join(HERE, "model_logits_32.npz"), x=xin, logits=logits)

    # 5. end-to-end test_cat (resized to 128x128), N = 8, argmax OPM, class 8, 10 SR iterations
    np.random.seed(1234)
    cat = o_aug.load_image(os.path.join(HERE, "test_cat.jpg"), image_size=(128, 128))
    copies, ang, sh = o_aug.create_augmented_copies(cat, 8, 0.15, 20)
    pred = OracleDeeplabV3Plus(w).predict(copies, batch_size=8)
    masks, _ = o_aug.opm(pred, 8, "argmax")
    opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=10, num_aug=8, optimizer=opt, feature_size=(32, 32),
                              output_size=(128, 128))
    finals = {t: o_sr.compute_SR(sr, masks, ang, sh, SR_type=t, class_id=8, th_factor=0.2)[..., 0] for t in ("aug", "max", "mean")}
    np.savez_compressed(os.path.join(HERE, "e2e_test_cat_128.npz"), angles=ang, shifts=sh,
                        lr_masks=np.stack(masks)[..., 0].astype(np.uint8),
                        **{f"mask_{t}": v.astype(np.uint8) for t, v in finals.items()})
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
