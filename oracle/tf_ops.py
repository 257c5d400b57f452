"""TF 2.7 / TFA 0.15 image-op semantics restated on torch-CPU float32.  TEST INFRASTRUCTURE ONLY.

The reference never implements these ops itself: it calls ``tfa.image.rotate`` /
``tfa.image.translate`` (superresolution_scripts/augmentation_utils.py:22-25,
superresolution_scripts/superresolution.py:61-64,142-147), ``tf.image.resize``
(superresolution.py:67-68,112-113,140-141; utils.py:105-106) and
``tf.image.image_gradients`` (superresolution.py:81).  The arithmetic lives in
tensorflow==2.7.0 / tensorflow-addons==0.15.0 (configs/requirements.txt:114-115), absent
from /root/reference; what follows restates their published kernels:

* ``ImageProjectiveTransformV3`` (interpolation BILINEAR, fill_mode CONSTANT, fill 0) and its
  registered gradient (same op applied to the upstream gradient with the matrix-inverse
  transform) -- TF ``image_ops.h`` ProjectiveGenerator / ``image_grad.py``.
* ``ResizeBilinear`` with half-pixel centres (TF2 default, no antialias) and its exact-adjoint
  gradient ``ResizeBilinearGrad``.
* ``tfa.image.angles_to_projective_transforms`` / ``translations_to_projective_transforms``.

Every elementwise step is a separate float32 torch op (no FMA contraction), in the order the
TF kernels write them.  PARITY UNPINNED (see package docstring).
"""
from __future__ import annotations

import numpy as np
import torch

F32 = torch.float32


def _t(x, dtype=F32):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


# --------------------------------------------------------------------------------------
# transform vectors
# --------------------------------------------------------------------------------------
def angles_to_projective_transforms(angles, image_height, image_width):
    """tfa.image.angles_to_projective_transforms (called by tfa.image.rotate;
    reference call sites augmentation_utils.py:22, superresolution.py:61,145,157).
    Returns float32 [N,8] = [cos,-sin,x_off, sin,cos,y_off, 0,0]."""
    a = np.asarray(angles, dtype=np.float32).reshape(-1)
    h = np.float32(image_height)
    w = np.float32(image_width)
    one = np.float32(1.0)
    two = np.float32(2.0)
    cos = np.cos(a).astype(np.float32)
    sin = np.sin(a).astype(np.float32)
    x_off = ((w - one) - (cos * (w - one) - sin * (h - one))) / two
    y_off = ((h - one) - (sin * (w - one) + cos * (h - one))) / two
    z = np.zeros_like(a)
    return np.stack([cos, -sin, x_off, sin, cos, y_off, z, z], axis=1).astype(np.float32)


def translations_to_projective_transforms(translations):
    """tfa.image.translations_to_projective_transforms (called by tfa.image.translate;
    reference call sites augmentation_utils.py:24, superresolution.py:63,142,154).
    translations [N,2] = (dx, dy) -> [1,0,-dx, 0,1,-dy, 0,0]."""
    t = np.asarray(translations, dtype=np.float32).reshape(-1, 2)
    n = t.shape[0]
    o = np.ones(n, np.float32)
    z = np.zeros(n, np.float32)
    return np.stack([o, z, -t[:, 0], z, o, -t[:, 1], z, z], axis=1).astype(np.float32)


def invert_transforms(transforms):
    """TF image_grad.py: flat_transforms_to_matrices -> matrix_inverse ->
    matrices_to_flat_transforms (divide by the [2,2] entry), all float32."""
    t = np.asarray(transforms, dtype=np.float32).reshape(-1, 8)
    n = t.shape[0]
    m = np.concatenate([t, np.ones((n, 1), np.float32)], axis=1).reshape(n, 3, 3)
    inv = np.linalg.inv(m).astype(np.float32)
    flat = inv.reshape(n, 9)
    flat = flat / flat[:, 8:9]
    return flat[:, :8].astype(np.float32)


# --------------------------------------------------------------------------------------
# ImageProjectiveTransformV3, BILINEAR, CONSTANT fill 0
# --------------------------------------------------------------------------------------
def projective_transform(images, transforms, output_shape=None, interpolation="bilinear"):
    """images [N,H,W,C] f32, transforms [N,8] (or [1,8]) f32 -> [N,Ho,Wo,C] f32.
    interpolation="nearest" (check_robustness.py:45-50 on the label maps): TF nearest_interpolation reads
    I(round(in_y), round(in_x)) with std::round (half away from zero), 0 outside.

    TF image_ops.h ProjectiveGenerator::operator() + bilinear_interpolation +
    read_with_fill_value: for output (x=col, y=row)
        projection = c0*x + c1*y + 1
        in_x = (a0*x + a1*y + a2) / projection ; in_y = (b0*x + b1*y + b2) / projection
        value  = (y_c - in_y) * [(x_c - in_x) I(y_f,x_f) + (in_x - x_f) I(y_f,x_c)]
               + (in_y - y_f) * [(x_c - in_x) I(y_c,x_f) + (in_x - x_f) I(y_c,x_c)]
    with x_f = floor(in_x), x_c = x_f + 1, and every out-of-bounds tap reading 0."""
    img = _t(images)
    tr = _t(transforms).reshape(-1, 8)
    n, h, w, c = img.shape
    if tr.shape[0] == 1 and n > 1:
        tr = tr.expand(n, 8)
    ho, wo = (h, w) if output_shape is None else (int(output_shape[0]), int(output_shape[1]))
    t = tr.reshape(n, 8, 1, 1)
    xs = torch.arange(wo, dtype=F32).reshape(1, 1, wo)
    ys = torch.arange(ho, dtype=F32).reshape(1, ho, 1)
    proj = t[:, 6] * xs + t[:, 7] * ys + 1.0
    in_x = (t[:, 0] * xs + t[:, 1] * ys + t[:, 2]) / proj
    in_y = (t[:, 3] * xs + t[:, 4] * ys + t[:, 5]) / proj
    flat = img.reshape(n, h * w, c)
    if interpolation == "nearest":
        rnd = lambda v: torch.sign(v) * torch.floor(torch.abs(v) + 0.5)          # std::round
        yy, xx = rnd(in_y), rnd(in_x)
        valid = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w) & (proj != 0)
        idx = (yy.clamp(0, h - 1).to(torch.int64) * w + xx.clamp(0, w - 1).to(torch.int64))
        v = torch.gather(flat, 1, idx.reshape(n, ho * wo, 1).expand(n, ho * wo, c)).reshape(n, ho, wo, c)
        return v * valid.reshape(n, ho, wo, 1).to(F32)
    x_f = torch.floor(in_x)
    y_f = torch.floor(in_y)
    x_c = x_f + 1.0
    y_c = y_f + 1.0

    def read(yy, xx):
        valid = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        yi = yy.clamp(0, h - 1).to(torch.int64)
        xi = xx.clamp(0, w - 1).to(torch.int64)
        idx = (yi * w + xi).reshape(n, ho * wo, 1).expand(n, ho * wo, c)
        v = torch.gather(flat, 1, idx).reshape(n, ho, wo, c)
        return v * valid.reshape(n, ho, wo, 1).to(F32)

    wx_lo = (x_c - in_x).unsqueeze(-1)
    wx_hi = (in_x - x_f).unsqueeze(-1)
    wy_lo = (y_c - in_y).unsqueeze(-1)
    wy_hi = (in_y - y_f).unsqueeze(-1)
    v_yf = wx_lo * read(y_f, x_f) + wx_hi * read(y_f, x_c)
    v_yc = wx_lo * read(y_c, x_f) + wx_hi * read(y_c, x_c)
    out = wy_lo * v_yf + wy_hi * v_yc
    out = torch.where((proj == 0).unsqueeze(-1), torch.zeros((), dtype=F32), out)
    return out


def projective_transform_grad(grad, transforms, input_hw):
    """Registered gradient of ImageProjectiveTransformV3 w.r.t. images (TF image_grad.py):
    the SAME op applied to the upstream gradient with the inverted transform, output shape =
    input shape.  Not the scatter adjoint."""
    return projective_transform(grad, invert_transforms(np.asarray(transforms)), output_shape=input_hw)


def rotate(images, angles, interpolation="bilinear"):
    """tfa.image.rotate(images, angles, interpolation=...) (augmentation_utils.py:22, check_robustness.py:46-47)."""
    n, h, w, _ = images.shape
    return projective_transform(images, angles_to_projective_transforms(angles, h, w), interpolation=interpolation)


def translate(images, translations, interpolation="bilinear"):
    """tfa.image.translate(images, shifts, interpolation=...) (augmentation_utils.py:24, check_robustness.py:48-49)."""
    return projective_transform(images, translations_to_projective_transforms(translations), interpolation=interpolation)


# --------------------------------------------------------------------------------------
# tf.image.resize (bilinear / nearest), TF2 half-pixel centres
# --------------------------------------------------------------------------------------
def _interp_weights(out_size, in_size):
    """TF compute_interpolation_weights with HalfPixelScaler:
    in = (o + 0.5) * scale - 0.5 ; lower = max(floor(in),0) ; upper = min(ceil(in), in-1);
    lerp = in - floor(in).  scale = in_size / out_size in float32."""
    scale = np.float32(in_size) / np.float32(out_size)
    o = np.arange(out_size, dtype=np.float32)
    pos = (o + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(pos)
    lower = np.maximum(fl.astype(np.int64), 0)
    upper = np.minimum(np.ceil(pos).astype(np.int64), in_size - 1)
    lerp = (pos - fl).astype(np.float32)
    return lower, upper, lerp


def resize_bilinear(images, size):
    """tf.image.resize(images, size) default method (bilinear, antialias=False).
    images [N,H,W,C] -> [N,size[0],size[1],C].  TF resize_bilinear_op compute_lerp:
    top = tl + (tr - tl) * xl ; bottom = bl + (br - bl) * xl ; out = top + (bottom - top) * yl."""
    img = _t(images)
    n, h, w, c = img.shape
    ho, wo = int(size[0]), int(size[1])
    ylo, yhi, yl = _interp_weights(ho, h)
    xlo, xhi, xl = _interp_weights(wo, w)
    ylo_t, yhi_t = torch.from_numpy(ylo), torch.from_numpy(yhi)
    xlo_t, xhi_t = torch.from_numpy(xlo), torch.from_numpy(xhi)
    yl_t = torch.from_numpy(yl).reshape(1, ho, 1, 1)
    xl_t = torch.from_numpy(xl).reshape(1, 1, wo, 1)
    top_rows = img[:, ylo_t]
    bot_rows = img[:, yhi_t]
    tl = top_rows[:, :, xlo_t]
    tr = top_rows[:, :, xhi_t]
    bl = bot_rows[:, :, xlo_t]
    br = bot_rows[:, :, xhi_t]
    top = tl + (tr - tl) * xl_t
    bottom = bl + (br - bl) * xl_t
    return top + (bottom - top) * yl_t


def resize_bilinear_grad(grad, input_hw):
    """ResizeBilinearGrad: exact adjoint (scatter-add with the forward weights)."""
    g = _t(grad)
    n, ho, wo, c = g.shape
    h, w = int(input_hw[0]), int(input_hw[1])
    ylo, yhi, yl = _interp_weights(ho, h)
    xlo, xhi, xl = _interp_weights(wo, w)
    yl_t = torch.from_numpy(yl).reshape(1, ho, 1, 1)
    xl_t = torch.from_numpy(xl).reshape(1, 1, wo, 1)
    out = torch.zeros(n, h * w, c, dtype=F32)
    for (ys, wy) in ((ylo, 1.0 - yl_t), (yhi, yl_t)):
        for (xs_, wx) in ((xlo, 1.0 - xl_t), (xhi, xl_t)):
            idx = (torch.from_numpy(ys).reshape(ho, 1) * w + torch.from_numpy(xs_).reshape(1, wo)).reshape(-1)
            out.index_add_(1, idx, (g * wy * wx).reshape(n, ho * wo, c))
    return out.reshape(n, h, w, c)


def resize_nearest(images, size):
    """tf.image.resize(method='nearest') (utils.py:105-106 with resize_method='nearest';
    test_SR.py:86-87): half-pixel centres, in = min(floor((o + 0.5) * scale), in_size - 1)."""
    img = torch.as_tensor(np.asarray(images))
    n, h, w, c = img.shape
    ho, wo = int(size[0]), int(size[1])

    def idx(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        o = np.arange(out_size, dtype=np.float32)
        return np.minimum(np.floor((o + np.float32(0.5)) * scale).astype(np.int64), in_size - 1)

    yi = torch.from_numpy(idx(ho, h))
    xi = torch.from_numpy(idx(wo, w))
    return img[:, yi][:, :, xi]


def image_gradients(image):
    """tf.image.image_gradients (superresolution.py:81): forward differences, last row/col 0.
    Returns (dy, dx), each [N,H,W,C]."""
    x = _t(image)
    dy = torch.zeros_like(x)
    dx = torch.zeros_like(x)
    dy[:, :-1] = x[:, 1:] - x[:, :-1]
    dx[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1]
    return dy, dx
