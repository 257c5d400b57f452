"""GPU parity of the fused DeepLabV3+ engine against the unfused layer-by-layer torch-CPU oracle,
with the same seeded synthetic weights, and of the whole hot path through the reference-shaped
Python surface (config 1 of BASELINE.json: test_cat.jpg, num_aug=8, argmax OPM, class 8)."""
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


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_engine_logits_match_oracle(dev, synthetic, precision):
    """Both GEMM arithmetics: exact-f32 MFMA and the split-f16 MFMA (f32-grade by construction)."""
    from asr_amd.model import DeeplabModel
    rng = np.random.default_rng(21)
    x = rng.random((3, 64, 96, 3), dtype=np.float32)
    ref = OracleDeeplabV3Plus(synthetic).forward(x)
    model = DeeplabModel(synthetic, (64, 96, 3), 21, final_upsample=False, last_activation=None, precision=precision)
    assert model.precision == precision
    got = model.predict(x, batch_size=2)                      # 2 + 1: exercises two plans
    assert got.shape == ref.shape == (3, 16, 24, 21)
    scale = np.abs(ref).max()
    # float32 end to end, BN folded and ReLU/Add fused: differences are rounding only
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * scale)
    agree = (got.argmax(-1) == ref.argmax(-1)).mean()
    assert agree >= 0.999, agree


def test_xception_os8_matches_oracle(dev, synthetic):
    """model.py:42-47: OS = 8 -- entry block 3 at stride 1, middle flow dilated by 2, exit flow by (2, 4), ASPP rates
    (12, 24, 36); same weights, same decoder."""
    from asr_amd.model import DeeplabModel
    rng = np.random.default_rng(23)
    x = rng.random((2, 64, 64, 3), dtype=np.float32)
    ref, st = OracleDeeplabV3Plus(synthetic, OS=8).forward(x, return_stages=True)
    assert st["exit"].shape == (2, 8, 8, 2048)                       # 64 / 8
    model = DeeplabModel(synthetic, (64, 64, 3), 21, final_upsample=False, last_activation=None, OS=8)
    assert model.name == "DLV3Plus-xception-OS8"
    got = model.predict(x, batch_size=2)
    assert got.shape == ref.shape == (2, 16, 16, 21)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * np.abs(ref).max())


@pytest.mark.parametrize("kw,name,shape", [
    (dict(only_DCNN_output=True, first_upsample_size=(24, 40)), "DLV3Plus-xception-OS16-Only_DCNN_Output", (2, 24, 40, 21)),
    (dict(only_ASPP_output=True, first_upsample_size=(32, 32)), "DLV3Plus-xception-OS16-Only_ASPP_Output", (2, 32, 32, 21)),
    (dict(final_class_prediction=False), "DLV3Plus-xception-OS16-no_class_prediction", (2, 16, 16, 256)),
    (dict(only_ASPP_output=True, first_upsample_size=(16, 16), final_class_prediction=False, final_upsample=True),
     "DLV3Plus-xception-OS16-no_class_prediction", (2, 64, 64, 256)),
])
def test_modified_decoders_match_oracle(dev, kw, name, shape):
    """build_model(only_DCNN_output / only_ASPP_output / first_upsample_size / final_class_prediction), model.py:64-118
    and the decoders of :261-294, through the reference-shaped constructor; synthetic weights of the variant's inventory."""
    from asr_amd import weights as W
    from asr_amd.model import DeeplabV3Plus
    rng = np.random.default_rng(29)
    x = rng.random((2, 64, 64, 3), dtype=np.float32)
    decoder = "dcnn" if kw.get("only_DCNN_output") else ("aspp" if kw.get("only_ASPP_output") else "full")
    cp = kw.get("final_class_prediction", True)
    weights = W.make_synthetic_weights(1234, 21, decoder=decoder, class_prediction=cp)
    ref = OracleDeeplabV3Plus(weights, decoder=decoder, first_upsample_size=kw.get("first_upsample_size", (128, 128)),
                              class_prediction=cp).forward(x, final_upsample=kw.get("final_upsample", False))
    model = DeeplabV3Plus(input_shape=(64, 64, 3), classes=21, OS=16, synthetic_seed=1234).build_model(
        **{"final_upsample": False, **kw})
    assert model.name == name
    got = model.predict(x, batch_size=2)
    assert got.shape == ref.shape == shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * np.abs(ref).max())


def test_final_upsample_matches_oracle(dev, synthetic):
    from asr_amd.model import DeeplabModel
    rng = np.random.default_rng(22)
    x = rng.random((1, 64, 64, 3), dtype=np.float32)
    ref = OracleDeeplabV3Plus(synthetic).forward(x, final_upsample=True)
    got = DeeplabModel(synthetic, (64, 64, 3), 21, True, None).predict(x)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
    # reshape_outputs (model.py:120-122) + softmax over the class axis of the flattened output
    flat = DeeplabModel(synthetic, (64, 64, 3), 21, True, "softmax", reshape_outputs=True).predict(x)
    ref_sm = OracleDeeplabV3Plus(synthetic, last_activation="softmax").forward(x, final_upsample=True)
    assert flat.shape == (1, 64 * 64, 21)
    np.testing.assert_allclose(flat, ref_sm.reshape(1, 64 * 64, 21), rtol=0, atol=2e-5)


@pytest.mark.parametrize("side,fside,iters,shift_max", [(256, 64, 20, 40), (512, 128, 50, 80)],
                         ids=["256-reduced", "512-configs0-full-size"])
def test_hot_path_config1_test_cat(dev, synthetic, golden_dir, tmp_path, side, fside, iters, shift_max):
    """test_SR.py:57-97 on the bundled sample through the reference-shaped API: same seed -> same angles/shifts; masks
    from the HIP path vs the oracle path; IoU within 1e-3 (north_star bar).  The 512 case is BASELINE configs[0] exactly
    as stated (test_cat at 512x512, 128x128 features, num_aug=8, argmax OPM, class 8, angle 0.15 / shift 80 of
    test_SR.py:20-47) with 50 AMSGrad iterations; the 256 case is the quick reduced form."""
    from asr_amd.model import DeeplabModel
    from asr_amd.utils import load_image, compute_IoU
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.augmentation_utils import compute_augmented_feature_maps
    from asr_amd.superresolution_scripts.superres_utils import compute_SR

    img_path = os.path.join(golden_dir, "test_cat.jpg")
    gt_path = os.path.join(golden_dir, "test_cat_gt.png")
    size, fsize, n_aug, cls = (side, side), (fside, fside), 8, 8

    # ---- HIP path through the reference-shaped API ----
    np.random.seed(1234)
    model = DeeplabModel(synthetic, size + (3,), 21, False, None)
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=opt, feature_size=fsize,
                         output_size=size)
    masks, max_masks, angles, shifts, name = compute_augmented_feature_maps(
        img_path, model, filter_class_id=cls, mode="argmax", num_aug=n_aug, angle_max=0.15, shift_max=shift_max,
        image_size=size, batch_size=4)
    assert name == "test_cat" and len(masks) == n_aug and masks[0].shape == fsize + (1,)
    out = {t: compute_SR(sr, masks, angles, shifts, name, str(tmp_path), SR_type=t, max_masks=max_masks,
                         class_id=cls, th_factor=0.2) for t in ("aug", "max", "mean")}
    gt = load_image(gt_path, image_size=size, normalize=False, is_png=True, resize_method="nearest")

    # ---- oracle path, same seed ----
    np.random.seed(1234)
    o_img = o_aug.load_image(img_path, image_size=size)
    o_copies, o_angles, o_shifts = o_aug.create_augmented_copies(o_img, n_aug, 0.15, shift_max)
    assert np.array_equal(o_angles, angles) and np.array_equal(o_shifts, shifts)
    o_pred = OracleDeeplabV3Plus(synthetic).predict(o_copies, batch_size=4)
    o_masks, _ = o_aug.opm(o_pred, cls, "argmax")
    lr_agree = np.mean([np.mean(a == b) for a, b in zip(masks, o_masks)])
    assert lr_agree >= 0.999, lr_agree
    o_opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    o_srobj = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=o_opt,
                                   feature_size=fsize, output_size=size)
    o_gt = o_aug.load_image(gt_path, image_size=size, normalize=False, is_png=True, resize_method="nearest")
    assert np.array_equal(o_gt, gt)
    for t in ("aug", "max", "mean"):
        ref = o_sr.compute_SR(o_srobj, o_masks, o_angles, o_shifts, SR_type=t, class_id=cls, th_factor=0.2)
        # mask-vs-mask IoU (identical augmentation seeds) and IoU-vs-GT delta, both within 1e-3
        assert (ref == cls).any(), t                   # class 8 must be present: an empty mask would compare vacuously
        assert o_aug.single_class_IOU(ref, out[t], cls, False) >= 0.999, t
        d = abs(np.nan_to_num(compute_IoU(gt, out[t], img_size=size, class_id=cls)) -
                np.nan_to_num(o_aug.compute_IoU(o_gt, ref, img_size=size, class_id=cls)))
        assert d <= 1e-3, (t, d)


def test_two_lane_pipeline_equals_sequential(dev, synthetic):
    """Two images in flight on alternating HIP streams with separate activation pools: same masks and IoUs as run_image."""
    from asr_amd.model import DeeplabModel
    from asr_amd.pipeline import HotPath
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd import distributed as D, ops
    rng = np.random.default_rng(33)
    size, n_aug, iters = 128, 6, 8
    model = DeeplabModel(synthetic, (size, size, 3), 21, False, None)
    imgs = [ops.to_device(rng.random((size, size, 3), dtype=np.float32)) for _ in range(4)]
    gts = [ops.to_device(rng.choice(np.array([0, 8], np.int32), size=(size, size)), torch.int32) for _ in range(4)]
    params = D.replay_augmentation_stream(4, n_aug, 0.15, 20)

    def make():
        opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=opt, feature_size=(size // 4,) * 2,
                             output_size=(size, size))
        return HotPath(model, sr, class_id=8, mode="argmax", th_factor=0.15, batch_size=n_aug)

    seq_path, lane_path = make(), make()
    seq = [seq_path.run_image(imgs[i], *params[i], gt_dev=gts[i], adam_start=D.adam_start_step(i, iters)) for i in range(4)]
    handles = [lane_path.submit_lane(i % 2, imgs[i], *params[i], gt_dev=gts[i], adam_start=D.adam_start_step(i, iters))
               for i in range(4)]
    for a, h in zip(seq, handles):
        b = h.result()
        for k in ("standard", "aug", "max", "mean"):
            assert torch.equal(a[k], b[k]), k
        np.testing.assert_array_equal(a["ious"], b["ious"])


def test_pipelined_submit_equals_sequential(dev, synthetic):
    """The side-stream pipeline (SR of image i under the forward pass of image i+1) must give exactly the
    masks and IoUs of the sequential path."""
    from asr_amd import ops
    from asr_amd.model import DeeplabModel
    from asr_amd.pipeline import HotPath
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.augmentation_utils import draw_augmentation_parameters
    size, feat, n_aug, iters = (128, 128), (32, 32), 6, 8
    model = DeeplabModel(synthetic, size + (3,), 21, False, None)
    rng = np.random.default_rng(5)
    imgs = [ops.to_device(rng.random(size + (3,), dtype=np.float32)) for _ in range(3)]
    gt = np.zeros(size, np.int32)
    gt[30:90, 40:100] = 8
    gtd = ops.to_device(gt, torch.int32)
    np.random.seed(7)
    params = [draw_augmentation_parameters(n_aug, 0.15, 20) for _ in imgs]

    def make():
        opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=opt, feature_size=feat,
                             output_size=size)
        return HotPath(model, sr, class_id=8, mode="argmax", th_factor=0.2, batch_size=6)

    seq = make()
    ref = [seq.run_image(im, a, s, gt_dev=gtd, adam_start=i * iters) for i, (im, (a, s)) in enumerate(zip(imgs, params))]
    pipe = make()
    handles = [pipe.submit_image(im, a, s, gt_dev=gtd, adam_start=i * iters) for i, (im, (a, s)) in enumerate(zip(imgs, params))]
    got = [h.result() for h in handles]
    for r, g in zip(ref, got):
        for k in ("standard", "aug", "max", "mean"):
            assert torch.equal(r[k], g[k]), k
        np.testing.assert_array_equal(r["ious"], g["ious"])


@pytest.mark.parametrize("case", ["large-activations", "tiny-activations", "large-weights"])
def test_range_guard_routes_layers_to_the_exact_f32_kernels(dev, synthetic, case):
    """The split-f16 GEMMs cover operands in [2^-10, 2^15); outside it the engine moves the layer to asr_pwconv_mfma_f32:
    weights at upload (folded kernel out of range), activations by calibrate_range on a probe batch.  The logits then
    match the oracle as closely as everywhere else; without calibration an overflowing activation still gives finite
    logits (the kernels saturate)."""
    from asr_amd.model import DeeplabModel
    w = dict(synthetic)
    if case == "large-activations":       # everything after the stem is 3e4 times larger: up to ~1e5 inside the net
        w["entry_flow_conv1_1_BN/gamma"] = w["entry_flow_conv1_1_BN/gamma"] * np.float32(3e4)
        w["entry_flow_conv1_1_BN/beta"] = w["entry_flow_conv1_1_BN/beta"] * np.float32(3e4)
    elif case == "tiny-activations":
        w["entry_flow_conv1_1_BN/gamma"] = w["entry_flow_conv1_1_BN/gamma"] * np.float32(1e-6)
        w["entry_flow_conv1_1_BN/beta"] = w["entry_flow_conv1_1_BN/beta"] * np.float32(1e-6)
    else:                                 # one folded kernel beyond f16: routed at upload, before any data is seen
        w["aspp0/kernel"] = w["aspp0/kernel"] * np.float32(1e6)
    rng = np.random.default_rng(41)
    x = rng.random((2, 64, 64, 3), dtype=np.float32)
    ref = OracleDeeplabV3Plus(w).predict(x, batch_size=2)
    model = DeeplabModel(w, (64, 64, 3), 21, False, None, precision="f16x3")
    if case == "large-weights":
        assert "aspp0" in model.engine.routed_f32
    raw = model.predict(x, batch_size=2)
    assert np.isfinite(raw).all()
    # the on-demand check names the layers this input drives out of range without changing anything ...
    report = model.check_range(x)
    routed_before = dict(model.engine.routed_f32)
    assert len(report) > 0 and model.engine.routed_f32 == routed_before
    moved = model.calibrate_range(x)
    assert set(report) <= set(moved)                          # ... and calibration then moves (at least) those
    if case == "large-activations":
        assert len(moved) >= 10 and "middle_flow_unit_8_separable_conv2_pointwise" in moved
    elif case == "tiny-activations":     # the next BatchNorm's beta brings the scale back: only the stem's second conv sees it
        assert "entry_flow_conv1_2" in moved
    got = model.predict(x, batch_size=2)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
    assert model.calibrate_range(x) == {}                       # idempotent: nothing left to move
    assert model.check_range(x) == {}                           # and the input is now inside every layer's range
    if case == "large-weights":          # a LATER input far beyond the probe's range is reported on demand, nothing is re-routed
        plain = DeeplabModel(synthetic, (64, 64, 3), 21, False, None, precision="f16x3")
        assert plain.check_range(x) == {}
        late = plain.check_range(x * np.float32(1e5))
        assert len(late) > 0 and plain.engine.routed_f32 == {}


def test_hot_path_forward_batches_do_not_change_the_result(dev, synthetic):
    """HotPath pushes the copies through the model one forward batch at a time (what augmentation_utils.py:30-59 chunks
    for): a batch size that does not divide the copies (3, 3, 2), one copy per batch and all 8 at once give bit-identical
    masks and IoU records.  With an SR output smaller than the image the shifts are applied in the SR frame."""
    from asr_amd import ops
    from asr_amd.model import DeeplabModel
    from asr_amd.pipeline import HotPath
    from asr_amd.superresolution_scripts.augmentation_utils import draw_augmentation_parameters
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    rng = np.random.default_rng(3)
    img = ops.to_device(rng.random((128, 128, 3), dtype=np.float32), device=dev)
    gt = ops.to_device((rng.random((64, 64)) > 0.5).astype(np.int32) * 8, torch.int32, device=dev)
    np.random.seed(99)
    angles, shifts = draw_augmentation_parameters(8, 0.15, 20)
    model = DeeplabModel(synthetic, (128, 128, 3), 21, False, None)

    def run(bs, mode):
        opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=3, num_aug=8, optimizer=opt, feature_size=(32, 32), output_size=(64, 64))
        path = HotPath(model, sr, class_id=8, mode=mode, th_factor=0.2, batch_size=bs)
        assert np.array_equal(path._sr_frame(img, shifts), (shifts * np.float32(0.5)).astype(np.float32))   # 128 -> 64: x 0.5
        res = path.run_image(img, angles, shifts, gt_dev=gt, adam_start=0)
        return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}

    for mode in ("argmax", "slice_max"):
        ref = run(8, mode)
        for bs in (3, 1):
            got = run(bs, mode)
            for k in ("standard", "aug", "max", "mean"):
                assert np.array_equal(got[k], ref[k]), (mode, bs, k)
            np.testing.assert_array_equal(got["ious"], ref["ious"])
