"""``Superresolution`` with the reference's surface (superresolution_scripts/superresolution.py:26-161):
``loss_function``, ``augmented_superresolution``, ``max_superresolution``, ``mean_superresolution``
-- plus batched variants that solve many images in one sequence of launches.  All arithmetic
runs in the fused HIP kernels of csrc/sr.hip; this class only prepares float32 parameters.

``use_BTV`` swaps the TV prior for the bilateral TV of superresolution.py:8-23 (inside the same
kernels); ``copy_dropout`` drops a fixed random subset of the copies from the data term: the
reference draws the mask with np.random.shuffle inside a @tf.function (superresolution.py:44-53),
i.e. ONCE, when the function is traced on the first solve of a Superresolution object, and every
later iteration and image reuses it -- ``_drop_mask`` restates exactly that.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib, ops, transforms as T
from .optimizer import Optimizer


def _stack_copies(copies, device):
    """list of [h,w,1] / [h,w] arrays, stacked ndarray or tensor -> device tensor [N,h,w]."""
    if isinstance(copies, torch.Tensor):
        t = copies.to(device=device, dtype=torch.float32)
    else:
        if isinstance(copies, (list, tuple)):
            if len(copies) and isinstance(copies[0], torch.Tensor):
                t = torch.stack([c.to(device=device, dtype=torch.float32) for c in copies])
            else:
                t = torch.as_tensor(np.stack([np.asarray(c, dtype=np.float32) for c in copies])).to(device)
        else:
            t = torch.as_tensor(np.asarray(copies, dtype=np.float32)).to(device)
    if t.dim() == 4 and t.shape[-1] == 1:
        t = t[..., 0]
    if t.dim() != 3:
        raise ValueError(f"augmented copies must be [N,h,w(,1)], got {tuple(t.shape)}")
    return t.contiguous()


def bilateral_tv(target_image, alpha=0.6, shift_factor=2):
    """The reference's module-level prior (superresolution.py:8-23): sum over the 15 integer shifts p = (h, v), h in
    [-s, s], v in [0, s], of alpha^(|h|+|v|) * || x - translate(x, p) ||_1 (zero fill).  target_image [1,H,W,1] / [H,W]
    (host array or device tensor) -> Python float; evaluated by the solver's own prior kernel (sr_loss_prior_kernel)."""
    import ctypes as C
    dev = _lib.require_gpu()
    t = target_image if isinstance(target_image, torch.Tensor) else torch.as_tensor(np.asarray(target_image, dtype=np.float32))
    t = t.to(device=dev, dtype=torch.float32)
    if t.dim() == 4:
        if t.shape[0] != 1 or t.shape[-1] != 1:
            raise ValueError(f"bilateral_tv expects [1,H,W,1] or [H,W], got {tuple(t.shape)}")
        t = t[0, :, :, 0]
    if t.dim() != 2:
        raise ValueError(f"bilateral_tv expects [1,H,W,1] or [H,W], got {tuple(t.shape)}")
    x = t.contiguous()[None]
    H, W = x.shape[1:]
    resid = torch.zeros(1, dtype=torch.float32, device=dev)                 # no data term: one zero residual
    terms = torch.empty((1, 4), dtype=torch.float64, device=dev)
    cfg = ops.sr_config(use_btv=True, btv_alpha=alpha, btv_shift=shift_factor)
    ops.call("asr_sr_loss_terms_cfg_f64", ops.ptr(x), ops.ptr(resid), ops.ptr(terms, torch.float64), 1, 1, H, W, 1, 1,
             C.byref(cfg), ops.stream_ptr())
    return float(terms[0, 1].item())


class Superresolution:
    def __init__(self, lambda_df, lambda_tv, lambda_L2, lambda_L1, num_iter=200, num_aug=100,
                 optimizer: Optimizer = None, feature_size=(64, 64), output_size=(512, 512), use_BTV=False,
                 verbose=False, copy_dropout=0.0):
        self.lambda_df = lambda_df
        self.lambda_tv = lambda_tv
        self.lambda_L2 = lambda_L2
        self.lambda_L1 = lambda_L1
        self.num_iter = num_iter
        self.num_aug = num_aug
        self.optimizer = optimizer
        self.feature_size = tuple(feature_size)
        self.output_size = tuple(output_size)
        self.use_BTV = use_BTV
        self.verbose = verbose
        self.copy_dropout = copy_dropout
        self._drop_masks = {}        # n_drop -> bool [num_aug], frozen at first use (tf.function trace time)

    # -- parameter preparation ----------------------------------------------------------------
    @property
    def _lambdas(self):
        return (self.lambda_df, self.lambda_tv, self.lambda_L2, self.lambda_L1)

    def _transforms(self, angles, shifts, device, inverse=False, negate=False):
        """angles [B,N], shifts [B,N,2] -> device [B,N,8] rotation and translation transforms."""
        H, Wd = self.output_size
        angles = np.asarray(angles, dtype=np.float32)
        shifts = np.asarray(shifts, dtype=np.float32)
        if negate:
            angles, shifts = -angles, -shifts
        b, n = angles.shape
        rot = T.rotation_transforms(angles.reshape(-1), H, Wd)
        tr = T.translation_transforms(shifts.reshape(-1, 2))
        if inverse:
            rot, tr = T.inverse_transforms(rot), T.inverse_transforms(tr)
        return ops.to_device(rot.reshape(b, n, 8), device=device), ops.to_device(tr.reshape(b, n, 8), device=device)

    def _drop_mask(self, n_drop):
        """superresolution.py:47-50; np.random.shuffle consumes the global numpy stream once per (object, n_drop)."""
        if n_drop not in self._drop_masks:
            mask = np.full(self.num_aug, fill_value=True)
            mask[:n_drop] = False
            np.random.shuffle(mask)
            self._drop_masks[n_drop] = mask
        return self._drop_masks[n_drop]

    def _config(self):
        """asr_sr_config of this solver: the optimizer's update rule + the TV / bilateral-TV choice."""
        if self.optimizer is None:
            return ops.sr_config(use_btv=self.use_BTV)
        return self.optimizer.optimizer.config(use_btv=self.use_BTV)

    @staticmethod
    def _batchify(angles, shifts):
        a = np.asarray(angles, dtype=np.float32)
        s = np.asarray(shifts, dtype=np.float32)
        return (a[None], s[None]) if a.ndim == 1 else (a, s)

    # -- superresolution.py:44-100 ------------------------------------------------------------------
    def loss_function(self, target_image, augmented_samples, angles, shifts, n_drop=0):
        dev = _lib.require_gpu()
        y = _stack_copies(augmented_samples, dev)[None]
        a, s = self._batchify(angles, shifts)
        if n_drop != 0:
            keep = torch.as_tensor(self._drop_mask(n_drop), device=dev)
            y, a, s = y[:, keep].contiguous(), a[:, self._drop_mask(n_drop)], s[:, self._drop_mask(n_drop)]
        x = torch.as_tensor(np.asarray(target_image.cpu() if isinstance(target_image, torch.Tensor) else target_image,
                                       dtype=np.float32)).to(dev).reshape(1, *self.output_size).contiguous()
        rot, tr = self._transforms(a, s, dev)
        resid = ops.sr_forward_residual(x, y, rot, tr)
        return self._loss_from_terms(ops.sr_loss_terms(x, resid, self._config()).cpu().numpy()[0])

    def _loss_from_terms(self, t):
        f = np.float32
        loss = f(self.lambda_df) * f(t[0]) + f(self.lambda_tv) * f(t[1])
        loss = loss + f(self.lambda_L2) * f(t[2])
        if self.lambda_L1 > 0.0:
            loss = loss + f(self.lambda_L1) * f(t[3])
        return float(loss)

    # -- superresolution.py:102-137 -------------------------------------------------------------------
    def augmented_superresolution_batch(self, copies, angles, shifts):
        """copies [B,N,h,w] device tensor, angles [B,N], shifts [B,N,2] -> (device [B,H,W], [B] losses).
        The global Adam step counter advances image by image (reference order), the device then
        iterates all images together."""
        if self.optimizer is None:
            raise Exception("You must provide an instance of the Optimizer class to compute the augmented SR")
        dev = copies.device
        b, n, h, w = copies.shape
        if (h, w) != self.feature_size:
            raise ValueError(f"copies are {h}x{w} but feature_size is {self.feature_size}")
        x = ops.sr_init_target(copies, self.output_size)          # from copy 0 of the FULL stack (superresolution.py:112-114)
        n_drop = int(self.num_aug * self.copy_dropout)
        if n_drop != 0:
            if n != self.num_aug:
                raise ValueError(f"copy_dropout needs num_aug={self.num_aug} copies per image, got {n}")
            mask = self._drop_mask(n_drop)
            copies = copies[:, torch.as_tensor(mask, device=dev)].contiguous()
            angles, shifts = np.asarray(angles)[:, mask], np.asarray(shifts)[:, mask]
        rot, tr = self._transforms(angles, shifts, dev)
        irot, itr = self._transforms(angles, shifts, dev, inverse=True)
        alphas = np.stack([self.optimizer.schedule_alphas(self.num_iter) for _ in range(b)], axis=1)  # [iter, B]
        if self.num_iter == 0:
            return x, [None] * b
        state = self.optimizer.optimizer
        alphas_dev = ops.to_device(alphas, device=dev)
        kw = dict(want_loss=True, cfg=state.config(use_btv=self.use_BTV), slot_init=state.slot_init)
        if not self.verbose:
            return ops.sr_solve(x, copies, rot, tr, irot, itr, alphas_dev, self._lambdas, **kw)
        # superresolution.py:130-131 prints the loss of iteration i (evaluated before that iteration's update) for
        # i % 10 == 0 and for the last one.  The solver reports the loss of the LAST iteration of a call, so the solve is
        # cut after each printing iteration; the slots and the workspace carry over (same updates as one call).
        carry, first, terms = {}, 0, None
        for i in range(self.num_iter):
            if i % 10 == 0 or i == self.num_iter - 1:
                x, terms = ops.sr_solve(x, copies, rot, tr, irot, itr, alphas_dev[first:i + 1].contiguous(), self._lambdas,
                                        state=carry, **kw)
                first = i + 1
                for bi, t in enumerate(terms.cpu().numpy()):
                    tag = f"[image {bi}] " if b > 1 else ""
                    print(f"{tag}{i + 1}/{self.num_iter} -- loss = {self._loss_from_terms(t)}")
        return x, terms

    def augmented_superresolution(self, augmented_copies, angles, shifts):
        if self.optimizer is None:
            raise Exception(
                "You must provide an instance of the Optimizer class to compute the augmented SR")
        dev = _lib.require_gpu()
        y = _stack_copies(augmented_copies, dev)[None]
        a, s = self._batchify(angles, shifts)
        x, terms = self.augmented_superresolution_batch(y, a, s)
        loss = self._loss_from_terms(terms.cpu().numpy()[0]) if isinstance(terms, torch.Tensor) else None
        return x[0].cpu().numpy()[..., None], loss

    # -- superresolution.py:139-161 ---------------------------------------------------------------------
    def realign_batch(self, copies, angles, shifts, mode):
        """copies [B,N,h,w] device -> device [B,H,W]; translate(-shift) then rotate(-angle), max / mean
        (mode "both": the pair (max, mean) from one pass)."""
        rot, tr = self._transforms(angles, shifts, copies.device, negate=True)
        return ops.realign(copies, tr, rot, self.output_size, mode)

    def _realign_single(self, augmented_copies, angles, shifts, mode):
        dev = _lib.require_gpu()
        y = _stack_copies(augmented_copies, dev)[None]
        a, s = self._batchify(angles, shifts)
        return self.realign_batch(y, a, s, mode)[0].cpu().numpy()[..., None], None

    def max_superresolution(self, augmented_copies, angles, shifts):
        return self._realign_single(augmented_copies, angles, shifts, "max")

    def mean_superresolution(self, augmented_copies, angles, shifts):
        return self._realign_single(augmented_copies, angles, shifts, "mean")
