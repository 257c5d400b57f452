"""Host-side float32 parameter math for the warp / SR kernels.

The kernels take ImageProjectiveTransformV3-style 8-vectors; this module builds them the way
tensorflow-addons does for ``tfa.image.rotate`` / ``tfa.image.translate`` (reference call sites
superresolution_scripts/augmentation_utils.py:22-25, superresolution.py:61-64,142-147) and
inverts them the way TensorFlow's registered gradient does (3x3 matrix inverse, renormalised).
"""
from __future__ import annotations

import numpy as np

_F = np.float32


def rotation_transforms(angles, height, width):
    """[N] angles (radians) -> [N,8] float32, rotation about the image centre."""
    a = np.asarray(angles, dtype=_F).reshape(-1)
    c, s = np.cos(a).astype(_F), np.sin(a).astype(_F)
    wm1, hm1 = _F(width) - _F(1), _F(height) - _F(1)
    x_off = (wm1 - (c * wm1 - s * hm1)) / _F(2)
    y_off = (hm1 - (s * wm1 + c * hm1)) / _F(2)
    out = np.zeros((a.shape[0], 8), dtype=_F)
    out[:, 0], out[:, 1], out[:, 2] = c, -s, x_off
    out[:, 3], out[:, 4], out[:, 5] = s, c, y_off
    return out


def translation_transforms(shifts):
    """[N,2] (dx, dy) -> [N,8] float32: output (x,y) reads input (x - dx, y - dy)."""
    t = np.asarray(shifts, dtype=_F).reshape(-1, 2)
    out = np.zeros((t.shape[0], 8), dtype=_F)
    out[:, 0] = 1
    out[:, 4] = 1
    out[:, 2] = -t[:, 0]
    out[:, 5] = -t[:, 1]
    return out


def inverse_transforms(tf8):
    """Flat transforms -> 3x3 -> float32 inverse -> flat (divided by the last entry)."""
    t = np.asarray(tf8, dtype=_F).reshape(-1, 8)
    mats = np.concatenate([t, np.ones((t.shape[0], 1), _F)], axis=1).reshape(-1, 3, 3)
    inv = np.linalg.inv(mats).astype(_F).reshape(-1, 9)
    inv = inv / inv[:, 8:9]
    return np.ascontiguousarray(inv[:, :8], dtype=_F)


def exponential_decay_lr(initial_lr, decay_steps, decay_rate, step):
    """ExponentialDecay (non-staircase) in float32: lr0 * rate ** (step / decay_steps)."""
    return _F(_F(initial_lr) * np.power(_F(decay_rate), _F(step) / _F(decay_steps), dtype=_F))


def adam_alpha(lr, beta_1, beta_2, step):
    """Keras Adam step size for 1-based global step t: lr * sqrt(1 - b2^t) / (1 - b1^t), float32."""
    t = _F(step)
    b1p = np.power(_F(beta_1), t, dtype=_F)
    b2p = np.power(_F(beta_2), t, dtype=_F)
    return _F(_F(lr) * np.sqrt(_F(1) - b2p, dtype=_F) / (_F(1) - b1p))
