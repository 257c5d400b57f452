"""The only artefacts the reference holds for this path: the three masks its own test_SR.py:57-97 wrote for the bundled
sample with the pretrained weights (test_images/SR_output/{aug,max,mean}_SR/test_cat_*_SR.png, committed here as data
under tests/golden/reference_SR_output/), and the sample's ground truth.

  * CPU: the oracle's load_image (nearest resize of the label map) + compute_IoU (utils.py:94-112, 180-230) reproduce,
    on the reference's own output masks, the IoUs recorded for them in SURVEY.md 4 / BASELINE.md -- the part of the path
    the reference pins without the weights.
  * GPU, skipped unless ASR_WEIGHTS points at the bonlime checkpoint the reference downloads
    (deeplabv3_xception_tf_dim_ordering_tf_kernels.h5, model.py:9,134-145; not available offline): the whole
    scripts/test_SR.py flow with the real weights -- weights.load_weights on the real file, real activations through the
    split-f16 GEMMs, seed-1234 augmentation, 100 copies, 300 AMSGrad iterations -- against those three masks and IoUs.
    This is the test that turns "parity unpinned" green the day the checkpoint is supplied.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

REF_IOU = {"aug": 0.75695, "max": 0.65947, "mean": 0.76557}        # the committed masks vs test_cat_gt.png, class 8
SIZE = (512, 512)


def _reference_mask(t):
    from PIL import Image
    a = np.array(Image.open(os.path.join(GOLDEN, "reference_SR_output", f"test_cat_{t}_SR.png")))
    assert a.shape == SIZE and set(np.unique(a)) <= {0, 255}
    return ((a > 0).astype(np.int32) * 8)[..., None]


def test_reference_masks_reproduce_the_recorded_ious():
    from oracle import augment as o_aug
    gt = o_aug.load_image(os.path.join(GOLDEN, "test_cat_gt.png"), image_size=SIZE, normalize=False, is_png=True,
                          resize_method="nearest")
    assert set(np.unique(gt)) == {0, 8, 255}
    for t, want in REF_IOU.items():
        got = o_aug.compute_IoU(gt, _reference_mask(t), img_size=SIZE, class_id=8)
        assert abs(got - want) < 5e-6, (t, got)
    # void pixels are NOT excluded (utils.py:193-199): excluding them would give different numbers
    m = _reference_mask("aug")[..., 0]
    g = gt[..., 0]
    keep = g != 255
    excl = ((m == 8) & (g == 8) & keep).sum() / (((m == 8) | (g == 8)) & keep).sum()
    assert abs(excl - REF_IOU["aug"]) > 1e-3


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("ASR_WEIGHTS"), reason="needs the pretrained bonlime .h5 (ASR_WEIGHTS=/path/to/file); "
                    "a network download in the reference (model.py:134-143), unavailable offline")
def test_real_weights_reproduce_the_reference_outputs(dev, tmp_path):
    from asr_amd.model import DeeplabV3Plus
    from asr_amd.utils import compute_IoU, load_image
    from asr_amd.superresolution_scripts.augmentation_utils import compute_augmented_feature_maps
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superres_utils import compute_SR
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from oracle import augment as o_aug
    np.random.seed(1234)                                                     # test_SR.py:16-17
    model = DeeplabV3Plus(input_shape=SIZE + (3,), classes=21, OS=16, last_activation=None, load_weights=True,
                          backbone="xception", weights_path=os.environ["ASR_WEIGHTS"]).build_model(final_upsample=False)
    opt = Optimizer(optimizer="adam", learning_rate=1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(lambda_df=1.0, lambda_tv=0.3, lambda_L2=0.7, lambda_L1=0.0, num_iter=300, num_aug=100,
                         optimizer=opt, feature_size=(128, 128))
    masks, max_masks, angles, shifts, name = compute_augmented_feature_maps(
        os.path.join(GOLDEN, "test_cat.jpg"), model, filter_class_id=8, mode="argmax", num_aug=100, angle_max=0.15,
        shift_max=80, image_size=SIZE, batch_size=16)
    gt = load_image(os.path.join(GOLDEN, "test_cat_gt.png"), image_size=SIZE, normalize=False, is_png=True,
                    resize_method="nearest")
    for t in ("aug", "max", "mean"):
        got = compute_SR(sr, masks, angles, shifts, name, str(tmp_path), SR_type=t, max_masks=max_masks, class_id=8,
                         th_factor=0.2)
        ref = _reference_mask(t)
        # tolerance: 1e-3 on the IoU against the ground truth (north_star), and the two masks themselves within 1e-2 of
        # each other (the JPEG decoder and TF's float summation order differ by single border pixels)
        assert abs(compute_IoU(gt, got, img_size=SIZE, class_id=8) - REF_IOU[t]) <= 1e-3, t
        assert o_aug.single_class_IOU(ref, got, 8, False) >= 0.99, t
