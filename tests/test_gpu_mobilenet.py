"""the MobileNetV2 backbone (model.py:308-379, 426-461; OS = 8, ASPP = image pooling + aspp0, no
decoder) on the same kernels -- engine vs the unfused oracle, and the hot path with it (model output 64x64 at 512x512,
i.e. 8x SR; here 16x16 -> 128x128)."""
import os

import numpy as np
import pytest

from oracle import augment as o_aug
from oracle import sr as o_sr
from oracle.model import OracleDeeplabV3Plus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mobile_weights():
    from asr_amd import weights as W
    return W.make_synthetic_weights(seed=77, classes=21, backbone="mobilenet")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_mobilenet_logits_match_oracle(dev, mobile_weights, precision):
    from asr_amd.model import DeeplabModel
    rng = np.random.default_rng(31)
    x = rng.random((3, 64, 96, 3), dtype=np.float32)
    ref, ref_stages = OracleDeeplabV3Plus(mobile_weights, backbone="mobilenet").forward(x, return_stages=True)
    model = DeeplabModel(mobile_weights, (64, 96, 3), 21, final_upsample=False, last_activation=None, precision=precision,
                         backbone="mobilenet")
    assert model.name == "DLV3Plus-mobilenet-OS8"
    got = model.predict(x, batch_size=2)
    assert got.shape == ref.shape == (3, 8, 12, 21)                   # OS 8
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
    # ReLU6 must actually clip somewhere, or the test would not notice a plain ReLU
    assert ref_stages["entry"].shape == (3, 32, 48, 16)
    up = DeeplabModel(mobile_weights, (64, 96, 3), 21, True, "softmax", precision=precision, backbone="mobilenet").predict(x)
    ref_up = OracleDeeplabV3Plus(mobile_weights, last_activation="softmax", backbone="mobilenet").forward(x, final_upsample=True)
    np.testing.assert_allclose(up, ref_up, rtol=0, atol=2e-5)


def test_relu6_epilogues_clip(dev):
    """relu = 2 in the GEMM epilogue, post_relu = 2 in the depthwise kernel, relu = 2 in the stem: min(max(v, 0), 6)."""
    import torch
    from asr_amd import _lib, ops
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((64, 32)) * 4).astype(np.float32)
    w = np.eye(32, dtype=np.float32) * 3
    out = torch.empty((64, 32), dtype=torch.float32, device=dev)
    wp = ops.pack_pw_weights(ops.to_device(w))
    _lib.call("asr_pwconv_mfma_f32", ops.to_device(x).data_ptr(), wp.data_ptr(), None, None, out.data_ptr(), 64, 32, 32, 32,
              32, 0, 2, 1, 0, 0, _lib.stream_ptr())
    np.testing.assert_allclose(out.cpu().numpy(), np.clip(3 * x, 0, 6), rtol=1e-6, atol=1e-6)
    assert (out == 6).any() and (out == 0).any()
    img = (rng.standard_normal((1, 8, 8, 32)) * 6).astype(np.float32)
    k = np.zeros((3, 3, 32), np.float32)
    k[1, 1] = 1.0
    y = torch.empty((1, 8, 8, 32), dtype=torch.float32, device=dev)
    _lib.call("asr_dwconv3x3_nhwc_f32", ops.to_device(img).data_ptr(), ops.to_device(k).data_ptr(),
              ops.to_device(np.zeros(32, np.float32)).data_ptr(), y.data_ptr(), 1, 8, 8, 32, 1, 1, 1, 1, 8, 8, 32, 32, 0, 2, 0,
              _lib.stream_ptr())
    np.testing.assert_array_equal(y.cpu().numpy(), np.clip(img, 0, 6))


def test_hot_path_with_mobilenet_backbone(dev, mobile_weights, golden_dir, tmp_path):
    """compute_augmented_feature_maps -> compute_SR with the MobileNet model: the SR factor is 8 (feature 16 -> 128)."""
    from asr_amd.model import DeeplabModel
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.augmentation_utils import compute_augmented_feature_maps
    from asr_amd.superresolution_scripts.superres_utils import compute_SR
    size, fsize, n_aug, cls, iters = (128, 128), (16, 16), 6, 8, 10
    img_path = os.path.join(golden_dir, "test_cat.jpg")
    np.random.seed(1234)
    model = DeeplabModel(mobile_weights, size + (3,), 21, False, None, backbone="mobilenet")
    masks, max_masks, angles, shifts, name = compute_augmented_feature_maps(
        img_path, model, filter_class_id=cls, mode="slice", num_aug=n_aug, angle_max=0.15, shift_max=20, image_size=size,
        batch_size=6)
    assert np.stack(masks).shape == (n_aug, 16, 16, 1)
    np.random.seed(1234)
    o_img = o_aug.load_image(img_path, image_size=size)
    o_copies, o_angles, o_shifts = o_aug.create_augmented_copies(o_img, n_aug, 0.15, 20)
    o_pred = OracleDeeplabV3Plus(mobile_weights, backbone="mobilenet").predict(o_copies, batch_size=6)
    o_masks, _ = o_aug.opm(o_pred, cls, "slice")
    np.testing.assert_allclose(np.stack(masks), np.stack(o_masks), rtol=0, atol=2e-5)
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    o_opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=opt, feature_size=fsize, output_size=size)
    o_srobj = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n_aug, optimizer=o_opt, feature_size=fsize,
                                   output_size=size)
    for t in ("aug", "max", "mean"):
        got = compute_SR(sr, masks, angles, shifts, name, str(tmp_path), SR_type=t, class_id=cls, th_factor=0.5)
        ref = o_sr.compute_SR(o_srobj, o_masks, o_angles, o_shifts, SR_type=t, class_id=cls, th_factor=0.5)
        assert float(np.mean(got == ref)) >= 0.999, t


def test_nearest_warp_and_mean_iou_match_oracle(dev):
    """check_robustness.py:45-51,128-133: label maps are rotated / translated with NEAREST interpolation and scored with
    the multi-class Mean_IOU (utils.py:151-177) -- bit-exact vs the oracle (index arithmetic only)."""
    import torch
    from asr_amd import ops, transforms as T
    from asr_amd.utils import compute_IoU
    from oracle import tf_ops
    rng = np.random.default_rng(41)
    lab = rng.choice(np.array([0, 3, 8, 15, 255], np.float32), size=(3, 40, 56, 1))
    angles = np.array([0.3, -0.45, 0.0], np.float32)
    shifts = np.array([[7, -3], [-12, 9], [0, 0]], np.float32)
    rot = ops.to_device(T.rotation_transforms(angles, 40, 56))
    tr = ops.to_device(T.translation_transforms(shifts))
    got = ops.warp_affine(ops.warp_affine(ops.to_device(lab), rot, interpolation="nearest"), tr, interpolation="nearest")
    ref = tf_ops.translate(tf_ops.rotate(torch.from_numpy(lab), angles, interpolation="nearest"), shifts, interpolation="nearest")
    assert np.array_equal(got.cpu().numpy(), ref.numpy())
    assert np.array_equal(got[2].cpu().numpy(), lab[2])                              # identity copy untouched
    pred = rng.choice(np.array([0, 3, 8, 15], np.int32), size=(40, 56))
    truth = ref.numpy()[0, :, :, 0].astype(np.int32)
    assert compute_IoU(truth, pred, img_size=(40, 56)) == pytest.approx(o_aug.Mean_IOU(truth, pred), rel=1e-12)
    counts = ops.class_counts(ops.to_device(truth, torch.int32), ops.to_device(pred, torch.int32)).cpu().numpy()[0]
    for l in (0, 3, 8, 15, 255):
        assert counts[0, l] == (truth == l).sum() and counts[1, l] == (pred == l).sum()
        assert counts[2, l] == ((truth == l) & (pred == l)).sum()
    only_void = np.full((40, 56), 255, np.int32)
    assert np.isnan(compute_IoU(only_void, pred, img_size=(40, 56)))                 # no scorable label -> NaN, dropped later
