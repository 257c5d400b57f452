"""CPU restatement of augmentation, OPM and IoU glue.  TEST INFRASTRUCTURE ONLY.

Follows superresolution_scripts/augmentation_utils.py:11-27 (create_augmented_copies),
:62-138 (compute_augmented_feature_maps: OPM modes), utils.py:94-119 (load_image,
create_mask) and utils.py:180-230 (single_class_IOU / compute_IoU) of the reference.
PARITY UNPINNED (see package docstring).
"""
from __future__ import annotations

import numpy as np

from . import tf_ops
from .sr import min_max_normalization


def draw_angles_shifts(num_aug, angle_max, shift_max):
    """augmentation_utils.py:14-20: numpy GLOBAL legacy RNG, angles first then shifts,
    entry 0 forced to identity, cast to float32."""
    angles = np.random.uniform(-angle_max, angle_max, num_aug)
    shifts = np.random.uniform(-shift_max, shift_max, (num_aug, 2))
    angles[0] = 0
    shifts[0] = np.array([0, 0])
    return angles.astype("float32"), shifts.astype("float32")


def create_augmented_copies(image, num_aug, angle_max, shift_max):
    """augmentation_utils.py:11-27: tile -> rotate -> translate (two bilinear resamplings)."""
    image = np.asarray(image, dtype=np.float32)
    angles, shifts = draw_angles_shifts(num_aug, angle_max, shift_max)
    batched = np.broadcast_to(image[None], (num_aug,) + image.shape)
    import torch
    rot = tf_ops.rotate(torch.as_tensor(np.ascontiguousarray(batched)), angles)
    out = tf_ops.translate(rot, shifts)
    return out.numpy(), angles, shifts


def load_image(img_path, image_size=None, normalize=True, is_png=False, resize_method="bilinear"):
    """utils.py:94-112.  JPEG/PNG decode via PIL (TF's decoder is libjpeg-turbo too; its
    default dct_method may differ by +-1 level -- unpinnable offline), then tf.image.resize
    semantics (half-pixel, no antialias) restated in tf_ops."""
    from PIL import Image
    img = Image.open(img_path)
    if not is_png:
        arr = np.asarray(img.convert("RGB"))
    else:
        arr = np.asarray(img)                      # palette PNG -> raw indices, like decode_png(channels=1)
        if arr.ndim == 3:
            arr = arr[..., :1]
        else:
            arr = arr[..., None]
    if image_size is not None:
        if resize_method == "nearest":
            arr = tf_ops.resize_nearest(arr[None], image_size)[0].numpy()
        else:
            arr = tf_ops.resize_bilinear(arr[None].astype(np.float32), image_size)[0].numpy()
    arr = arr.astype(np.float32)
    if normalize:
        arr = arr / np.float32(255.0)
    return arr


def create_mask(pred):
    """utils.py:115-119: argmax over the class axis (first max wins), keep a trailing dim."""
    return np.argmax(np.asarray(pred), axis=-1)[..., None].astype(np.int64)


def opm(predictions, filter_class_id, mode):
    """augmentation_utils.py:80-115, per copy.  Returns (class_masks, max_masks) lists."""
    class_masks, max_masks = [], []
    for p in np.asarray(predictions, dtype=np.float32):
        if mode == "slice_max":
            cm = p[..., filter_class_id][..., None]
            others = np.delete(np.arange(p.shape[-1]), filter_class_id)
            max_masks.append(p[..., others].max(axis=-1)[..., None])
        elif mode == "slice":
            cm = p[..., filter_class_id][..., None]
            cm = min_max_normalization(cm, new_min=0.0, new_max=1.0, global_min=p.min(), global_max=p.max())
        else:
            m = create_mask(p)
            cm = np.where(m == filter_class_id, m, 0).astype(np.float32)
        class_masks.append(np.asarray(cm, dtype=np.float32))
    return class_masks, max_masks


def single_class_IOU(y_true, y_pred, class_id, include_bg):
    """utils.py:180-204.  int32 counts, float64 ratio, NaN classes dropped, mean.  Void (255)
    pixels are NOT excluded."""
    t = np.asarray(y_true).reshape(-1)
    p = np.asarray(y_pred).reshape(-1)
    classes = [class_id]
    if include_bg:
        classes.append(0)
        t = np.where(t != class_id, 0, t)
    ious = []
    for c in classes:
        tl = t == c
        pl = p == c
        inter = np.int32((tl & pl).sum())
        union = np.int32((tl | pl).sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            ious.append(np.float64(inter) / np.float64(union))
    ious = np.array(ious)
    ious = ious[~np.isnan(ious)]
    return float(np.mean(ious)) if len(ious) else float("nan")


def compute_IoU(true_image, image, img_size=(512, 512), class_id=None, include_bg=False):
    """utils.py:207-230 (single-class mode only)."""
    t = np.asarray(true_image).reshape(img_size[0] * img_size[1], 1)
    p = np.asarray(image).reshape(img_size[0] * img_size[1], 1)
    if class_id is None:
        raise NotImplementedError("oracle restates the single-class mode only")
    return single_class_IOU(t, p, class_id, include_bg)


def Mean_IOU(y_true, y_pred):
    """utils.py:151-177: mean over the labels PRESENT IN THE GROUND TRUTH (void 255 removed) of
    |true == i & pred == i| / |true == i | pred == i|."""
    t = np.asarray(y_true).astype(np.int64).reshape(-1)
    p = np.asarray(y_pred).astype(np.int64).reshape(-1)
    ious = []
    for i in np.unique(t):
        if i == 255:
            continue
        tl, pl = t == i, p == i
        ious.append(np.sum(tl & pl) / np.sum(tl | pl))
    return float(np.mean(ious)) if ious else float("nan")
