#!/usr/bin/env python3
"""Single-image end-to-end run of the Augmented Super-Resolution path on MI355X -- counterpart of the
reference's test_SR.py (same constants, same call sequence, asr_amd modules instead of TF).

    python scripts/test_SR.py [--weights weights.npz] [--image tests/golden/test_cat.jpg] [--gt ...png]

Without --weights, seeded synthetic DeepLabV3+ weights are used (the pretrained .h5 of the reference is a
network download), so the printed IoUs are only meaningful with real weights.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from asr_amd.model import DeeplabV3Plus  # noqa: E402
from asr_amd.utils import load_image, compute_IoU  # noqa: E402
from asr_amd.superresolution_scripts.optimizer import Optimizer  # noqa: E402
from asr_amd.superresolution_scripts.superresolution import Superresolution  # noqa: E402
from asr_amd.superresolution_scripts.augmentation_utils import compute_augmented_feature_maps  # noqa: E402
from asr_amd.superresolution_scripts.superres_utils import compute_SR  # noqa: E402

SEED = 1234
np.random.seed(SEED)

# General parameters (test_SR.py:20-27 of the reference)
IMG_SIZE = (512, 512)
FEATURE_SIZE = (128, 128)
BATCH_SIZE = 16
CLASS_ID = 8
MODE = "argmax"
MODEL_BACKBONE = "xception"
# Augmentation parameters
NUM_AUG = 100
ANGLE_MAX = 0.15
SHIFT_MAX = 80
# Optimizer parameters
OPTIMIZER = "adam"
LEARNING_RATE = 1e-3
AMSGRAD = True
LR_SCHEDULER = True
DECAY_STEPS = 60
DECAY_RATE = 0.3
# Super-resolution parameters
LAMBDA_DF = 1.0
LAMBDA_TV = 0.3
LAMBDA_L2 = 0.7
LAMBDA_L1 = 0.0
NUM_ITER = 300
TH_FACTOR = 0.2


def main():
    ap = argparse.ArgumentParser()
    golden = os.path.join(ROOT, "tests", "golden")
    ap.add_argument("--image", default=os.path.join(golden, "test_cat.jpg"))
    ap.add_argument("--gt", default=os.path.join(golden, "test_cat_gt.png"))
    ap.add_argument("--weights", default=None, help="local .npz of Keras weights (layer/variable keys)")
    ap.add_argument("--out", default=os.path.join(ROOT, "SR_output"))
    ap.add_argument("--num-aug", type=int, default=NUM_AUG)
    ap.add_argument("--num-iter", type=int, default=NUM_ITER)
    ap.add_argument("--mode", default=MODE, choices=["argmax", "slice", "slice_max"])
    ap.add_argument("--backbone", default=MODEL_BACKBONE, choices=["xception", "mobilenet"],
                    help="mobilenet: model output 64x64 (OS 8), i.e. 8x super-resolution")
    args = ap.parse_args()

    model = DeeplabV3Plus(input_shape=IMG_SIZE + (3,), classes=21, OS=16, last_activation=None, load_weights=True,
                          backbone=args.backbone, weights_path=args.weights).build_model(final_upsample=False)
    optimizer_obj = Optimizer(optimizer=OPTIMIZER, learning_rate=LEARNING_RATE, amsgrad=AMSGRAD,
                              lr_scheduler=LR_SCHEDULER, decay_steps=DECAY_STEPS, decay_rate=DECAY_RATE)
    superresolution_obj = Superresolution(lambda_df=LAMBDA_DF, lambda_tv=LAMBDA_TV, lambda_L2=LAMBDA_L2,
                                          lambda_L1=LAMBDA_L1, num_iter=args.num_iter, num_aug=args.num_aug,
                                          optimizer=optimizer_obj,
                                          feature_size=FEATURE_SIZE if args.backbone == "xception" else (IMG_SIZE[0] // 8, IMG_SIZE[1] // 8))
    class_masks, max_masks, angles, shifts, filename = compute_augmented_feature_maps(
        args.image, model, filter_class_id=CLASS_ID, mode=args.mode, num_aug=args.num_aug, angle_max=ANGLE_MAX,
        shift_max=SHIFT_MAX, image_size=IMG_SIZE, batch_size=BATCH_SIZE)
    results = {}
    for sr_type in ("aug", "max", "mean"):
        results[sr_type] = compute_SR(superresolution_obj, class_masks, angles, shifts, filename, max_masks=max_masks,
                                      SR_type=sr_type, save_final_output=True, class_id=CLASS_ID, dest_folder=args.out,
                                      th_factor=TH_FACTOR)
    gt_mask = load_image(args.gt, image_size=IMG_SIZE, normalize=False, is_png=True, resize_method="nearest")
    ious = {k: compute_IoU(gt_mask, v, img_size=IMG_SIZE, class_id=CLASS_ID) for k, v in results.items()}
    print(f"Aug. SR ({args.mode} OPM) IoU: {ious['aug']}, Max SR IoU: {ious['max']}, Mean SR IoU: {ious['mean']}")


if __name__ == "__main__":
    main()
