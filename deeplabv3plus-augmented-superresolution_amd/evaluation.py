"""The per-image evaluation loop shared by SR_single_class.py:83-134 and sweep_script.py:96-171 of the reference:
interchange file -> augmented / max / mean SR -> six IoUs per image -> means.  Images are sharded over the ranks of
the node (asr_amd.distributed), the per-image records are all-gathered once at the end."""
from __future__ import annotations

import os

import numpy as np

from . import distributed as D
from .superresolution_scripts.superres_utils import compute_SR, load_SR_data
from .utils import compute_IoU, load_image

# wandb.log keys of sweep_script.py:164-171 <- columns of distributed.IOU_FIELDS
SWEEP_METRICS = {"aug_iou_single": "aug_single", "aug_iou_multiple": "aug_bg", "standard_iou_single": "standard_single",
                 "standard_iou_multiple": "standard_bg", "mean_iou": "mean", "max_iou": "max"}


def evaluate_precomputed(sr, paths, gt_dir, standard_dir=None, num_aug=100, class_id=8, th_factor=0.65,
                         img_size=(512, 512), out_dir=None, rank=0, world=1, save_final_output=False):
    """Returns the [len(paths), 6] IoU table (distributed.IOU_FIELDS order; NaN rows for invalid files) on every rank.
    ``sr.optimizer``'s global step counter is set per image to what the reference's sequential loop would have reached
    (image_index * num_iter * solves_per_image), so sharding does not change any update."""
    mine = D.shard_indices(len(paths), rank, world)
    records = []
    for g in mine:
        try:
            class_masks, max_masks, angles, shifts, filename = load_SR_data(paths[g], num_aug=num_aug)
        except Exception:
            print(f"File: {paths[g]} is invalid, skipping...")
            records.append([np.nan] * len(D.IOU_FIELDS))
            continue
        sr.optimizer.optimizer.iterations = D.adam_start_step(g, sr.num_iter, "slice_max" if max_masks is not None else "argmax")
        true_mask = load_image(os.path.join(gt_dir, f"{filename}.png"), image_size=img_size, normalize=False, is_png=True,
                               resize_method="nearest")
        mm = max_masks if max_masks is not None else []
        out = {t: compute_SR(sr, class_masks, angles, shifts, filename, max_masks=mm, SR_type=t, class_id=class_id,
                             dest_folder=out_dir, th_factor=th_factor, save_final_output=save_final_output)
               for t in ("aug", "max", "mean")}
        std = [np.nan, np.nan]
        if standard_dir:
            sm = load_image(os.path.join(standard_dir, f"{filename}.png"), image_size=img_size, normalize=False, is_png=True,
                            resize_method="nearest")
            std = [compute_IoU(true_mask, sm, img_size=img_size, class_id=class_id),
                   compute_IoU(true_mask, sm, img_size=img_size, class_id=class_id, include_bg=True)]
        records.append(std + [compute_IoU(true_mask, out["aug"], img_size=img_size, class_id=class_id),
                              compute_IoU(true_mask, out["aug"], img_size=img_size, class_id=class_id, include_bg=True),
                              compute_IoU(true_mask, out["max"], img_size=img_size, class_id=class_id),
                              compute_IoU(true_mask, out["mean"], img_size=img_size, class_id=class_id)])
    return D.all_gather_iou(mine, records, len(paths))


def sweep_metrics(table):
    """The dict sweep_script.py:164-171 hands to wandb.log."""
    m = D.mean_ious(table)
    return {k: m[v] for k, v in SWEEP_METRICS.items()}
