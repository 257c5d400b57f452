"""The other BASELINE.json configurations as parity cases (configs[1] is the bench / test_gpu_model):
  configs[2]  softmax + slice OPM (dense per-class float maps)          -> parity with the oracle
  configs[2'] slice_max OPM (two SR solves per image, threshold class >= max)
  configs[4]  1024x1024 inputs, num_aug = 200 (chunked), 2x SR (256 -> 512) -> properties at full size, the forward pass
              (incl. the fused ASPP depthwise on the 64 x 64 map) and a 200-copy 2x SR solve against the oracle
"""
import os

import numpy as np
import pytest
import torch

from oracle import augment as o_aug
from oracle import sr as o_sr
from oracle.model import OracleDeeplabV3Plus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synthetic():
    from asr_amd import weights as W
    return W.make_synthetic_weights(seed=1234, classes=21)


@pytest.mark.parametrize("mode,activation", [("slice", "softmax"), ("slice_max", None), ("slice", None)])
def test_float_opm_paths_match_oracle(dev, synthetic, golden_dir, tmp_path, mode, activation):
    from asr_amd.model import DeeplabModel
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.augmentation_utils import compute_augmented_feature_maps
    from asr_amd.superresolution_scripts.superres_utils import compute_SR
    size, fsize, n_aug, cls, iters = (128, 128), (32, 32), 6, 8, 10
    img_path = os.path.join(golden_dir, "test_cat.jpg")
    np.random.seed(1234)
    model = DeeplabModel(synthetic, size + (3,), 21, False, activation)
    masks, max_masks, angles, shifts, name = compute_augmented_feature_maps(
        img_path, model, filter_class_id=cls, mode=mode, num_aug=n_aug, angle_max=0.15, shift_max=20, image_size=size,
        batch_size=6)
    assert len(max_masks) == (n_aug if mode == "slice_max" else 0)

    np.random.seed(1234)
    o_img = o_aug.load_image(img_path, image_size=size)
    o_copies, o_angles, o_shifts = o_aug.create_augmented_copies(o_img, n_aug, 0.15, 20)
    o_pred = OracleDeeplabV3Plus(synthetic, last_activation=activation).predict(o_copies, batch_size=6)
    o_masks, o_max = o_aug.opm(o_pred, cls, mode)
    # dense float maps: f32 rounding only (softmax outputs are in [0,1]; logits O(1))
    np.testing.assert_allclose(np.stack(masks), np.stack(o_masks), rtol=0, atol=2e-5)
    if mode == "slice_max":
        np.testing.assert_allclose(np.stack(max_masks), np.stack(o_max), rtol=0, atol=2e-5)

    def solvers():
        opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        o_opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        return (Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=opt, feature_size=fsize,
                                output_size=size),
                o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=o_opt,
                                     feature_size=fsize, output_size=size))

    sr, o_srobj = solvers()
    for t in ("aug", "max", "mean"):
        got = compute_SR(sr, masks, angles, shifts, name, str(tmp_path), SR_type=t, max_masks=max_masks, class_id=cls,
                         th_factor=0.5)
        ref = o_sr.compute_SR(o_srobj, o_masks, o_angles, o_shifts, SR_type=t, max_masks=o_max, class_id=cls,
                              th_factor=0.5)
        agree = float(np.mean(got == ref))
        assert agree >= 0.999, (mode, t, agree)
    if mode == "slice_max":
        assert sr.optimizer.optimizer.iterations == 2 * iters         # two ASR solves (class map + max map)


def test_config4_1024_inputs_200_copies_2x_sr(dev, synthetic):
    """configs[4]: 1024x1024 inputs, num_aug=200 drawn in chunks of 100, model output 256x256, SR 2x (256 -> 512).
    Shifts / angles are applied in the SR output frame (512x512), as superresolution.py:61-64 does."""
    from asr_amd import ops
    from asr_amd.model import DeeplabModel
    from asr_amd.superresolution_scripts.augmentation_utils import create_augmented_copies_chunked, output_processing
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    g = torch.Generator(device="cpu").manual_seed(4)
    img = torch.rand((1024, 1024, 3), generator=g)
    np.random.seed(1234)
    copies, angles, shifts = create_augmented_copies_chunked(img.numpy(), 200, 0.15, 80, chunk_size=100)
    assert copies.shape == (200, 1024, 1024, 3) and angles.shape == (200,) and shifts.shape == (200, 2)
    assert np.array_equal(copies[0], img.numpy())                      # copy 0 un-augmented, also in the chunked path
    np.random.seed(1234)
    ref_a = np.random.uniform(-0.15, 0.15, 200).astype("float32")
    assert np.array_equal(angles[1:], ref_a[1:])                       # one draw for all chunks, reference order
    model = DeeplabModel(synthetic, (1024, 1024, 3), 21, False, None)
    preds = model.predict_device(copies[:8], batch_size=8)             # a slice of the copies keeps the test short
    assert preds.shape == (8, 256, 256, 21) and torch.isfinite(preds).all()
    y8, _ = output_processing(preds, 8, "argmax")
    # 2x SR on 200 synthetic LR masks at 256x256 -> 512x512 (f = 2: D is the 2x2 box mean)
    yy, xx = torch.meshgrid(torch.arange(256.), torch.arange(256.), indexing="ij")
    blob = (((yy - 128) / 70) ** 2 + ((xx - 120) / 90) ** 2 < 1).float().to(dev)
    y = blob[None, None].expand(1, 200, 256, 256).contiguous()
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=5, num_aug=200, optimizer=opt, feature_size=(256, 256),
                         output_size=(512, 512))
    zeros_a, zeros_s = np.zeros((1, 200), np.float32), np.zeros((1, 200, 2), np.float32)
    x0 = ops.sr_init_target(y, (512, 512))
    resid = ops.sr_forward_residual(x0, y, *sr._transforms(zeros_a, zeros_s, dev))
    # D(upsample(y)) - y: zero away from the blob edge, bounded at the edge (bilinear 2x then 2x2 box mean)
    assert float(resid.abs().max()) <= 0.4375 + 1e-6 and float((resid == 0).float().mean()) > 0.95
    x, terms = sr.augmented_superresolution_batch(y, angles[None], 0.5 * shifts[None])
    assert x.shape == (1, 512, 512) and torch.isfinite(x).all() and float(terms[0, 0]) > 0
    mx = sr.realign_batch(y, zeros_a, zeros_s, "max")
    assert torch.equal(mx, x0)


def test_config4_forward_and_sr_match_the_oracle_at_full_size(dev, synthetic):
    """configs[4] against the oracle at its real sizes.
    (a) The 1024 x 1024 forward pass of two copies (the un-augmented image and copy 1 of the seed-1234 stream) after the
        bench's class-bias calibration: logits within 2e-4 * max |logit|, argmax agreement >= 0.999 on a real class-8
        region.  This pins every layer at the 4x larger maps -- in particular the three ASPP depthwise convs on the 64 x 64
        map, which the fused phase kernel serves (csrc/dwconv.hip; round 2 fell back to three direct launches there).
    (b) A 200-copy solve at f = 2 (256 x 256 -> 512 x 512, D = 2x2 box mean), shifts in the SR frame (x 0.5), two AMSGrad
        iterations, against oracle.sr: x to 1e-6 (the same operation order; expected bit-identical), loss to 1e-4."""
    import sys
    from conftest import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from asr_amd import distributed as D, ops
    from asr_amd.model import DeeplabModel
    from asr_amd.superresolution_scripts import augmentation_utils as au
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from oracle import tf_ops
    size, cls, n = 1024, 8, 200
    img = bench.synth_image(np.random.default_rng(1234), size)
    img_dev = ops.to_device(img, device=dev)
    model = DeeplabModel(synthetic, (size, size, 3), 21, final_upsample=False, last_activation=None)
    delta = bench.calibrate_class_bias(model, img_dev, cls)
    weights = bench.shifted_weights(synthetic, cls, delta)
    angles, shifts = D.replay_augmentation_stream(1, n, 0.15, 80, seed=1234)[0]
    idx = [0, 1]
    copies = au.augment_on_device(img_dev, angles[idx], shifts[idx])
    logits = model.predict_device(copies, batch_size=2).cpu().numpy()
    tiled = torch.from_numpy(np.broadcast_to(img[None], (2, size, size, 3)).copy())
    o_copies = tf_ops.translate(tf_ops.rotate(tiled, angles[idx]), shifts[idx]).numpy()
    np.testing.assert_allclose(copies.cpu().numpy(), o_copies, rtol=0, atol=2e-6)
    o_logits = OracleDeeplabV3Plus(weights).predict(o_copies, batch_size=1)
    assert o_logits.shape == logits.shape == (2, 256, 256, 21)
    np.testing.assert_allclose(logits, o_logits, rtol=0, atol=2e-4 * np.abs(o_logits).max())
    frac = float((o_logits.argmax(-1) == cls).mean())
    assert 0.05 < frac < 0.8, frac
    assert float((logits.argmax(-1) == o_logits.argmax(-1)).mean()) >= 0.999

    # (b) 200 LR masks at 256 x 256: the two real ones plus shifted / rotated blobs (cheap, structured), N = 200, f = 2
    masks, _ = o_aug.opm(o_logits, cls, "argmax")
    base = (np.stack(masks)[..., 0] / np.float32(cls)).astype(np.float32)              # {0, 1}, [2, 256, 256]
    yy, xx = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
    y = np.empty((n, 256, 256), np.float32)
    rng = np.random.default_rng(5)
    for i in range(n):
        cy, cx, ry, rx = rng.uniform(90, 166), rng.uniform(90, 166), rng.uniform(40, 80), rng.uniform(40, 80)
        blob = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1).astype(np.float32)
        y[i] = np.maximum(blob, base[i % 2] * 0.5)
    sr_shifts = (shifts * np.float32(0.5)).astype(np.float32)
    iters = 2
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    o_opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n, optimizer=opt, feature_size=(256, 256),
                         output_size=(512, 512))
    o_srobj = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n, optimizer=o_opt, feature_size=(256, 256),
                                   output_size=(512, 512))
    got, loss = sr.augmented_superresolution(y[..., None], angles, sr_shifts)
    ref, ref_loss = o_srobj.augmented_superresolution(y[..., None], angles, sr_shifts)
    d = np.abs(got - ref)
    assert d.max() <= 1e-6, (d.max(), d.mean())
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    for mode in ("max", "mean"):
        g, _ = getattr(sr, f"{mode}_superresolution")(y[..., None], angles, sr_shifts)
        r, _ = getattr(o_srobj, f"{mode}_superresolution")(y[..., None], angles, sr_shifts)
        np.testing.assert_allclose(g, r, rtol=0, atol=2e-6)
