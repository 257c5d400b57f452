#!/usr/bin/env python3
"""Robustness of the standard DeepLabV3+ prediction to rotation / translation of its input -- counterpart of the
reference's check_robustness.py: for every (angle, shift_x, shift_y) of the grid, rotate + translate the images
(bilinear) and their ground truth (nearest), predict with the final-upsample model, argmax, and average the per-image
IoU (multi-class Mean_IOU, or single class with --single_class); one CSV row per grid point.  Images and labels stay in
HBM for the whole sweep; grid points are sharded over the ranks of the node and merged on rank 0."""
import argparse
import csv
import itertools
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED = 1234
random.seed(SEED)
np.random.seed(SEED)

IMG_SIZE = (512, 512)
BATCH_SIZE = 16


def default_grid():
    """check_robustness.py:103-106."""
    angle_values = [round(float(a), 2) for a in np.arange(-0.7, 0.75, step=0.05)]
    shift_values = [int(v) for v in np.linspace(-80, 80, num=9, dtype=int)]
    return angle_values, shift_values, shift_values


def augment_images(images, angle, shift_x, shift_y, interpolation="bilinear"):
    """check_robustness.py:45-51 on a device batch [N,H,W,C]: rotate, then translate, zero fill."""
    from asr_amd import ops, transforms as T
    _, h, w, _ = images.shape
    rot = ops.to_device(T.rotation_transforms(np.array([angle], np.float32), h, w)[0], device=images.device)
    tr = ops.to_device(T.translation_transforms(np.array([[shift_x, shift_y]], np.float32))[0], device=images.device)
    rotated = ops.warp_affine(images, rot, interpolation=interpolation)
    return ops.warp_affine(rotated, tr, interpolation=interpolation)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", required=True, help="folder of .jpg images")
    ap.add_argument("--gt", required=True, help="folder of label PNGs with the same file names")
    ap.add_argument("--num_samples", type=int, default=350)
    ap.add_argument("--backbone", choices=["xception", "mobilenet"], default="xception")
    ap.add_argument("--single_class", action="store_true")
    ap.add_argument("--class_id", type=int, default=8)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--angles", type=float, nargs="*", default=None, help="override the angle grid (radians)")
    ap.add_argument("--shifts", type=int, nargs="*", default=None, help="override both shift grids (pixels)")
    ap.add_argument("--image_size", type=int, default=IMG_SIZE[0])
    ap.add_argument("--out", default=os.path.join(ROOT, "data", "robustness_check"))
    args = ap.parse_args()

    import torch
    from asr_amd import distributed as D, ops
    from asr_amd.model import DeeplabV3Plus
    from asr_amd.utils import compute_IoU, create_mask, load_image

    rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    size = (args.image_size, args.image_size)
    names = sorted(f for f in os.listdir(args.images) if f.lower().endswith((".jpg", ".jpeg")))
    if len(names) > args.num_samples:
        names = random.sample(names, args.num_samples)                       # check_robustness.py:70
    images = ops.to_device(np.stack([load_image(os.path.join(args.images, n), image_size=size, normalize=True) for n in names]),
                           device=dev)
    gts = np.stack([load_image(os.path.join(args.gt, os.path.splitext(n)[0] + ".png"), image_size=size, normalize=False,
                               is_png=True, resize_method="nearest") for n in names]).astype(np.float32)
    gts = ops.to_device(gts.reshape(len(names), size[0], size[1], 1), device=dev)
    model = DeeplabV3Plus(input_shape=size + (3,), classes=21, OS=16, last_activation=None, load_weights=True,
                          backbone=args.backbone, reshape_outputs=False, alpha=1., weights_path=args.weights).build_model()

    angle_values, shift_x_values, shift_y_values = default_grid()
    if args.angles:
        angle_values = list(args.angles)
    if args.shifts:
        shift_x_values = shift_y_values = list(args.shifts)
    all_combinations = list(itertools.product(angle_values, shift_x_values, shift_y_values))
    mine = D.shard_indices(len(all_combinations), rank, world)
    rows = []
    for g in mine:
        angle, shift_x, shift_y = all_combinations[g]
        aug_images = augment_images(images, angle, shift_x, shift_y)
        aug_gt = augment_images(gts, angle, shift_x, shift_y, interpolation="nearest")
        predictions = model.predict_device(aug_images, batch_size=BATCH_SIZE)
        ious = []
        for k in range(predictions.shape[0]):
            pred_mask = create_mask(predictions[k])
            iou = round(compute_IoU(aug_gt[k], pred_mask, img_size=size, class_id=args.class_id if args.single_class else None), 3)
            ious.append(iou)
        ious = np.array(ious)
        ious = ious[~np.isnan(ious)]              # the object may leave the image when the ground truth is augmented
        avg = round(float(np.mean(ious)), 3) if len(ious) else float("nan")
        print(f"Angle: {angle}, Shift X: {shift_x}, Shift Y: {shift_y}, mIoU: {avg}, final ious: {len(ious)}", flush=True)
        rows.append([g, avg])
    # merge on rank 0: one all_gather of (grid index, mIoU) pairs, like the IoU records of the SR scripts
    table = np.full(len(all_combinations), np.nan)
    if world > 1:
        rec = torch.full((-(-len(all_combinations) // world), 2), -1.0, dtype=torch.float64)
        for i, (g, v) in enumerate(rows):
            rec[i, 0], rec[i, 1] = g, v
        rec = rec.to(D.collective_device(dev))
        gathered = [torch.empty_like(rec) for _ in range(world)]
        torch.distributed.all_gather(gathered, rec)
        for t in gathered:
            t = t.cpu().numpy()
            ok = t[:, 0] >= 0
            table[t[ok, 0].astype(np.int64)] = t[ok, 1]
    else:
        for g, v in rows:
            table[g] = v
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        csv_path = os.path.join(args.out, f"robustness_{len(names)}_class_{'all' if not args.single_class else str(args.class_id)}_small.csv")
        with open(csv_path, "w") as f:
            writer = csv.writer(f, delimiter=',', quotechar='"', quoting=csv.QUOTE_ALL)
            writer.writerow(["Angle", "Shift_X", "Shift_Y", "mIoU"])
            for (angle, sx, sy), v in zip(all_combinations, table):
                writer.writerow([angle, sx, sy, v])
        print(f"Done: {csv_path}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
