"""Counterparts of the reference's utils.py hot-path helpers: ``load_image`` (utils.py:94-112),
``create_mask`` (:115-119), ``single_class_IOU`` / ``compute_IoU`` (:180-230).  Plotting and
training losses are out of scope."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops


def _resize_nearest_host(arr, size):
    """tf.image.resize(method='nearest'), half-pixel centres: pure index selection on the host."""
    h, w = arr.shape[:2]
    ho, wo = int(size[0]), int(size[1])

    def idx(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        o = np.arange(out_size, dtype=np.float32)
        return np.minimum(np.floor((o + np.float32(0.5)) * scale).astype(np.int64), in_size - 1)

    return arr[idx(ho, h)][:, idx(wo, w)]


def load_image(img_path, image_size=None, normalize=True, is_png=False, resize_method="bilinear"):
    """Decode (PIL) -> optional resize -> float32 [H,W,C] host array (C = 3 for jpg, 1 for png).
    Bilinear resizing runs on the GPU (asr_resize_bilinear_f32, half-pixel, no antialias)."""
    from PIL import Image
    img = Image.open(img_path)
    if not is_png:
        arr = np.asarray(img.convert("RGB"))
    else:
        arr = np.asarray(img)
        arr = arr[..., :1] if arr.ndim == 3 else arr[..., None]
    if image_size is not None:
        if resize_method == "nearest":
            arr = _resize_nearest_host(arr, image_size)
        elif resize_method == "bilinear":
            dev = _lib.require_gpu()
            c = arr.shape[-1]
            cp = (c + 3) // 4 * 4
            x = torch.zeros((1,) + arr.shape[:2] + (cp,), dtype=torch.float32, device=dev)
            x[0, :, :, :c] = torch.as_tensor(arr.astype(np.float32)).to(dev)
            arr = ops.resize_bilinear(x, image_size)[0, :, :, :c].cpu().numpy()
        else:
            raise ValueError(f"unsupported resize_method {resize_method!r}")
    arr = arr.astype(np.float32)
    if normalize:
        arr = arr / np.float32(255.0)
    return arr


def create_mask(pred_mask):
    """argmax over the class axis with a trailing singleton axis (int64, like tf.argmax)."""
    if isinstance(pred_mask, torch.Tensor) and pred_mask.is_cuda:
        return ops.argmax(pred_mask.contiguous()).to(torch.int64).unsqueeze(-1)
    t = ops.to_device(np.asarray(pred_mask, dtype=np.float32))
    return ops.argmax(t).cpu().numpy().astype(np.int64)[..., None]


def get_prediction(model, input_image):
    """utils.py:122-127: predict one image and argmax it."""
    x = input_image if isinstance(input_image, torch.Tensor) else np.asarray(input_image, dtype=np.float32)
    prediction = model.predict(x[None, ...] if not isinstance(x, torch.Tensor) else x[None].cpu().numpy())
    return create_mask(prediction[0])


def print_labels(masks):
    """utils.py:144-148: label histograms of the (standard, super-resolved) mask pair."""
    title = ["Standard Labels: ", "Superres Labels: "]
    for i in range(2):
        m = masks[i].cpu().numpy() if isinstance(masks[i], torch.Tensor) else np.asarray(masks[i])
        values, count = np.unique(m, return_counts=True)
        print(title[i] + str(dict(zip(values, count))))


def _as_label_tensor(a, dev):
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=torch.int32).contiguous().reshape(-1)
    return torch.as_tensor(np.asarray(a).astype(np.int32).reshape(-1)).to(dev)


def iou_from_counts(counts, include_bg):
    """counts: [inter_c, union_c, inter_bg, union_bg] -> float64 mean of the non-NaN class IoUs."""
    with np.errstate(divide="ignore", invalid="ignore"):
        ious = [np.float64(counts[0]) / np.float64(counts[1])]
        if include_bg:
            ious.append(np.float64(counts[2]) / np.float64(counts[3]))
    ious = np.array(ious)
    ious = ious[~np.isnan(ious)]
    return float(np.mean(ious)) if len(ious) else float("nan")


def single_class_IOU(y_true, y_pred, class_id, include_bg):
    dev = _lib.require_gpu()
    t = _as_label_tensor(y_true, dev)
    p = _as_label_tensor(y_pred, dev)
    counts = ops.iou_counts(t, p, class_id, include_bg=include_bg).cpu().numpy()[0]
    return iou_from_counts(counts, include_bg)


def mean_iou_from_counts(counts):
    """counts [3, 256] (ops.class_counts) -> Mean_IOU (utils.py:151-177): mean over the labels present in the ground
    truth, void (255) removed, of inter / union; like tf.reduce_mean of an empty list, NaN when no label qualifies."""
    c = np.asarray(counts, dtype=np.int64)
    labels = [l for l in range(255) if c[0, l] > 0]
    if not labels:
        return float("nan")
    ious = [np.float64(c[2, l]) / np.float64(c[0, l] + c[1, l] - c[2, l]) for l in labels]
    return float(np.mean(ious))


def Mean_IOU(y_true, y_pred):
    dev = _lib.require_gpu()
    counts = ops.class_counts(_as_label_tensor(y_true, dev), _as_label_tensor(y_pred, dev)).cpu().numpy()[0]
    return mean_iou_from_counts(counts)


def compute_IoU(true_image, image, img_size=(512, 512), class_id=None, include_bg=False):
    """IoU of two label maps (utils.py:207-230): single class (optionally with background) when class_id is given,
    otherwise the multi-class Mean_IOU.  Void (255) pixels are NOT excluded from the single-class form, exactly
    like the reference."""
    n = img_size[0] * img_size[1]
    size = lambda a: a.numel() if isinstance(a, torch.Tensor) else np.asarray(a).size
    if size(true_image) != n or size(image) != n:
        raise ValueError(f"expected {n} pixels, got {size(true_image)} and {size(image)}")
    if class_id is None:
        return Mean_IOU(true_image, image)
    return single_class_IOU(true_image, image, class_id, include_bg)
