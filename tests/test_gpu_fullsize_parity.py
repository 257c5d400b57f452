"""Full-size parity against the oracle at the shapes bench.py times (BASELINE configs[1]: 512x512 images,
num_aug=100, 128x128 features, argmax OPM, class 8; configs[2]: the float OPM maps), through ``HotPath``'s own stages:

  * augment: all 100 copies are produced by the HIP kernel; 4 of them (first, second, a middle one, last) are compared
    with the oracle's tile -> rotate -> translate (augmentation_utils.py:11-27);
  * model + OPM: the oracle runs the unfused DeepLabV3+ on those 4 copies (model.py:64-147, ~10 s of CPU time); logits
    and argmax masks are compared with the HIP forward pass of the whole 100-copy batch;
  * SR: the oracle solves on ALL 100 low-resolution masks produced by the HIP path (superresolution.py:102-161) for 10
    AMSGrad iterations, plus max- and mean-SR; target, thresholded masks and IoUs are compared.

The reduced-size cases live in test_gpu_model.py / test_gpu_warp_sr.py; the size-independent identities at this size
in test_gpu_fullsize_properties.py."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import augment as o_aug
from oracle import sr as o_sr
from oracle import tf_ops
from oracle.model import OracleDeeplabV3Plus

pytestmark = pytest.mark.gpu

H = W = 512
h = w = 128
N = 100
CLS = 8
ITERS = 10
SAMPLE = (0, 1, 37, 99)


@pytest.fixture(scope="module")
def problem(dev):
    """One bench image through stage 1 of HotPath on the GPU (kept on the host for the oracle comparisons)."""
    from conftest import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from asr_amd import distributed as D, ops, weights as Wt
    from asr_amd.model import DeeplabModel
    from asr_amd.superresolution_scripts import augmentation_utils as au
    weights = Wt.make_synthetic_weights(1234, 21)
    model = DeeplabModel(weights, (H, W, 3), 21, final_upsample=False, last_activation=None)
    img = bench.synth_image(np.random.default_rng(1234))
    img_dev = ops.to_device(img, device=dev)
    # class 8 must occupy a real region, otherwise every comparison below is vacuous: the same bias shift goes into the
    # oracle's weights (float32 add on both sides)
    delta = bench.calibrate_class_bias(model, img_dev, CLS)
    weights = dict(weights)
    b = weights["logits_semantic/bias"].copy()
    b[CLS] = np.float32(b[CLS] + np.float32(delta))
    weights["logits_semantic/bias"] = b
    angles, shifts = D.replay_augmentation_stream(1, N, 0.15, 80, seed=1234)[0]
    copies = au.augment_on_device(img_dev, angles, shifts)
    logits = model.predict_device(copies, batch_size=N)
    masks, _ = au.output_processing(logits, CLS, "argmax")
    # configs[2]: the per-class float map path -- last_activation="softmax" (model.py:124-125) + "slice" OPM, and the
    # "slice_max" OPM on the raw logits -- on the same 100 copies
    model.last_activation = "softmax"
    probs = model.predict_device(copies, batch_size=N)
    model.last_activation = None
    slice_sm, _ = au.output_processing(probs, CLS, "slice")
    smax_cls, smax_max = au.output_processing(logits, CLS, "slice_max")
    torch.cuda.synchronize()
    idx = list(SAMPLE)
    out = dict(weights=weights, img=img, angles=angles, shifts=shifts, model=model,
               copies=copies[idx].cpu().numpy(), logits=logits[idx].cpu().numpy(), masks=masks.cpu().numpy(),
               probs=probs[idx].cpu().numpy(), slice_sm=slice_sm[idx].cpu().numpy(), smax_cls=smax_cls[idx].cpu().numpy(),
               smax_max=smax_max[idx].cpu().numpy())
    del copies, logits, probs
    return out


@pytest.fixture(scope="module")
def oracle_side(problem):
    """The oracle's copies and logits of the 4 sampled copies (~10 s of CPU time, shared by the tests below)."""
    p = problem
    idx = list(SAMPLE)
    tiled = torch.from_numpy(np.broadcast_to(p["img"][None], (len(idx), H, W, 3)).copy())
    o_copies = tf_ops.translate(tf_ops.rotate(tiled, p["angles"][idx]), p["shifts"][idx]).numpy()
    o_logits = OracleDeeplabV3Plus(p["weights"]).predict(o_copies, batch_size=len(idx))
    return dict(copies=o_copies, logits=o_logits)


def test_augment_forward_opm_at_full_size(problem, oracle_side):
    p = problem
    idx = list(SAMPLE)
    o_copies, o_logits = oracle_side["copies"], oracle_side["logits"]
    assert np.array_equal(p["copies"][0], p["img"])                       # copy 0 is the image itself
    np.testing.assert_allclose(p["copies"], o_copies, rtol=0, atol=2e-6)
    assert o_logits.shape == p["logits"].shape == (len(idx), h, w, 21)
    np.testing.assert_allclose(p["logits"], o_logits, rtol=0, atol=2e-4 * np.abs(o_logits).max())
    o_masks, _ = o_aug.opm(o_logits, CLS, "argmax")
    o_masks = np.stack(o_masks)[..., 0]
    frac = float((o_masks == CLS).mean())
    assert 0.05 < frac < 0.8, frac                                        # a real class-8 region
    agree = float((p["masks"][idx] == o_masks).mean())
    assert agree >= 0.999, agree


def test_float_opm_maps_at_full_size(problem, oracle_side):
    """BASELINE configs[2] at full size: softmax + "slice" (each copy's class-8 probability min-max-normalised by that
    prediction's own global min / max, augmentation_utils.py:95-104) and "slice_max" (class logit + max of the other 20,
    :82-93) against the oracle on the sampled copies."""
    p = problem
    o_logits = oracle_side["logits"]
    o_probs = torch.softmax(torch.from_numpy(o_logits), dim=-1).numpy()
    np.testing.assert_allclose(p["probs"], o_probs, rtol=0, atol=2e-5)
    o_slice, _ = o_aug.opm(o_probs, CLS, "slice")
    o_slice = np.stack(o_slice)[..., 0]
    assert o_slice.max() - o_slice.min() > 0.5                            # a dense float map with structure, not a constant
    np.testing.assert_allclose(p["slice_sm"], o_slice, rtol=0, atol=5e-5)
    o_cls, o_max = o_aug.opm(o_logits, CLS, "slice_max")
    scale = np.abs(o_logits).max()
    np.testing.assert_allclose(p["smax_cls"], np.stack(o_cls)[..., 0], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(p["smax_max"], np.stack(o_max)[..., 0], rtol=0, atol=2e-4 * scale)
    # the decision compute_SR takes from the two maps (class >= max of the others, superres_utils.py:253-256), per pixel
    agree = float(((p["smax_cls"] >= p["smax_max"]) == (np.stack(o_cls)[..., 0] >= np.stack(o_max)[..., 0])).mean())
    assert agree >= 0.999, agree


def test_sr_on_all_100_masks_at_full_size(dev, problem, tmp_path):
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.superres_utils import compute_SR
    p = problem
    y = (p["masks"] / np.float32(CLS)).astype(np.float32)                # load_SR_data's min-max normalisation of argmax masks
    assert set(np.unique(y)) == {0.0, 1.0}
    masks = [m[..., None] for m in y]
    gt = np.zeros((H, W), np.int32)
    gt[128:384, 96:400] = CLS
    gt[120:128, 96:400] = 255
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=ITERS, num_aug=N, optimizer=opt, feature_size=(h, w), output_size=(H, W))
    o_opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    o_obj = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=ITERS, num_aug=N, optimizer=o_opt, feature_size=(h, w),
                                 output_size=(H, W))
    # raw ASR target after 10 iterations: the sign() of the TV gradient makes single pixels chaotic w.r.t. 1-ulp
    # differences, so the trajectory is held in the mean and on the thresholded mask (as at the reduced size)
    got_x, _ = sr.augmented_superresolution(masks, p["angles"], p["shifts"])
    ref_x, _ = o_obj.augmented_superresolution(masks, p["angles"], p["shifts"])
    d = np.abs(np.asarray(got_x)[..., 0] - ref_x[..., 0])
    assert d.mean() < 1e-5 and d.max() < 5e-3, (d.mean(), d.max())
    opt.optimizer.iterations = 0
    # thresholded masks and IoUs: the oracle's ASR target from above (one oracle solve is ~100 s of CPU time), its max- and
    # mean-SR through compute_SR; the product side always through its own compute_SR (superres_utils.py:213-273)
    for t in ("aug", "max", "mean"):
        got = compute_SR(sr, masks, p["angles"], p["shifts"], "img", str(tmp_path), SR_type=t, class_id=CLS, th_factor=0.2)
        if t == "aug":
            ref = o_sr.threshold_image(ref_x, CLS, th_factor=0.2)
        else:
            ref = o_sr.compute_SR(o_obj, masks, p["angles"], p["shifts"], SR_type=t, class_id=CLS, th_factor=0.2)
        assert (ref == CLS).any(), t
        assert o_aug.single_class_IOU(ref, got, CLS, False) >= 0.999, t
        di = abs(o_aug.compute_IoU(gt, got, img_size=(H, W), class_id=CLS) - o_aug.compute_IoU(gt, ref, img_size=(H, W), class_id=CLS))
        assert di <= 1e-3, (t, di)
