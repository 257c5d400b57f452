"""Augmentation + feature-map generation with the reference's surface
(superresolution_scripts/augmentation_utils.py:11-138): ``create_augmented_copies``,
``create_augmented_copies_chunked``, ``compute_augmented_feature_maps``.

Random draws use numpy's GLOBAL legacy RNG in the reference's order (angles, then shifts; copy 0
forced to identity) so that ``np.random.seed(1234)`` reproduces the reference's augmentation
parameters.  The tile -> rotate -> translate chain is one fused kernel; the per-copy Python OPM
loop of the reference is one kernel launch over all copies.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib, ops, transforms as T
from ..utils import load_image
from . import superres_utils as su


def draw_augmentation_parameters(num_aug, angle_max, shift_max):
    """augmentation_utils.py:14-20 (float64 draws cast to float32)."""
    angles = np.random.uniform(-angle_max, angle_max, num_aug)
    shifts = np.random.uniform(-shift_max, shift_max, (num_aug, 2))
    # First sample is not augmented
    angles[0] = 0
    shifts[0] = np.array([0, 0])
    return angles.astype("float32"), shifts.astype("float32")


def augment_on_device(image_dev, angles, shifts, out=None):
    """image_dev [H,W,C] device tensor -> [N,H,W,C] device tensor (written into ``out`` when given)."""
    h, w, _ = image_dev.shape
    rot = ops.to_device(T.rotation_transforms(angles, h, w), device=image_dev.device)
    tr = ops.to_device(T.translation_transforms(shifts), device=image_dev.device)
    return ops.augment_copies(image_dev.contiguous(), rot, tr, out=out)


def _image_to_device(image):
    dev = _lib.require_gpu()
    if isinstance(image, torch.Tensor):
        return image.to(device=dev, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(image, dtype=np.float32)).to(dev)


def create_augmented_copies(image, num_aug, angle_max, shift_max, as_numpy=False):
    """Returns (copies [N,H,W,C], angles [N] f32, shifts [N,2] f32).  ``copies`` stays on the
    device (pass as_numpy=True for a host array, like the eager tensor's .numpy())."""
    angles, shifts = draw_augmentation_parameters(num_aug, angle_max, shift_max)
    copies = augment_on_device(_image_to_device(image), angles, shifts)
    return (copies.cpu().numpy() if as_numpy else copies), angles, shifts


def create_augmented_copies_chunked(image, num_aug, angle_max, shift_max, chunk_size=100):
    """augmentation_utils.py:30-59: same draws, copies produced chunk by chunk and returned on the host."""
    if (num_aug % chunk_size) != 0:
        raise Exception("Num aug must be a multiple of 50")
    num_chunks = num_aug // chunk_size
    angles, shifts = draw_augmentation_parameters(num_aug, angle_max, shift_max)
    img = _image_to_device(image)
    chunks = [augment_on_device(img, a, s).cpu().numpy()
              for a, s in zip(np.split(angles, num_chunks), np.split(shifts, num_chunks))]
    return np.concatenate(chunks, axis=0), angles, shifts


def output_processing(predictions, filter_class_id, mode, out=None, out_max=None):
    """predictions [N,h,w,C] device tensor -> (class_masks [N,h,w], max_masks [N,h,w] | None), device.
    argmax / slice / slice_max of augmentation_utils.py:80-115.  out / out_max: optional [N,h,w] destinations (the
    rows of a per-image stack that is filled forward batch by forward batch)."""
    predictions = predictions.contiguous()
    if mode == "slice_max":
        return ops.opm_slice_max(predictions, filter_class_id, out=out, out_max=out_max)
    if mode == "slice":
        return ops.opm_slice(predictions, filter_class_id, 0.0, 1.0, out=out), None
    return ops.opm_argmax(predictions, filter_class_id, out=out), None        # any other string = argmax, like the reference


def feature_maps_on_device(image_dev, model, filter_class_id, mode, angles, shifts, batch_size=16, profile=None):
    """augment -> model -> OPM entirely on the device.  Returns (class_masks, max_masks) [N,h,w]."""
    copies = augment_on_device(image_dev, angles, shifts)
    preds = model.predict_device(copies, batch_size=batch_size, profile=profile)
    return output_processing(preds, filter_class_id, mode)


def compute_augmented_feature_maps(image_path, model, filter_class_id, mode="slice", num_aug=100,
                                   angle_max=0.5, shift_max=30, image_size=(512, 512), batch_size=16, dest_folder=None):
    image_name = os.path.splitext(os.path.basename(image_path))[0]
    image = load_image(image_path, image_size=image_size, normalize=True)
    angles, shifts = draw_augmentation_parameters(num_aug, angle_max, shift_max)
    cls, mx = feature_maps_on_device(_image_to_device(image), model, filter_class_id, mode, angles, shifts,
                                     batch_size=batch_size)
    cls_h = cls.cpu().numpy()[..., None]
    class_masks = [cls_h[i] for i in range(num_aug)]
    max_masks = []
    if mx is not None:
        mx_h = mx.cpu().numpy()[..., None]
        max_masks = [mx_h[i] for i in range(num_aug)]
    if dest_folder is not None:
        su.save_SR_data(os.path.join(dest_folder, image_name), class_masks, max_masks if mode == "slice_max" else None,
                        angles, shifts, image_name, mode, angle_max, shift_max)
    return class_masks, max_masks, angles, shifts, image_name
