"""``DeeplabV3Plus`` with the reference's constructor / ``build_model`` / ``predict`` surface
(model.py:16-147), executing on hand-written gfx950 kernels through ``engine.DeeplabEngine``.

Scope: Xception backbone at OS=16 or OS=8 (model.py:42-52) and the MobileNetV2 backbone
(always OS=8, model.py:53-55), ``classes`` logits, ``final_upsample`` on or off, ``last_activation`` None /
softmax / sigmoid, ``reshape_outputs``, the modified decoders ``only_DCNN_output`` / ``only_ASPP_output`` with
``first_upsample_size`` (model.py:261-294, Xception only) and ``final_class_prediction=False`` (the decoder's 256-channel
features as the output, model.py:100-106).
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

from . import _lib, ops, weights as W
from .engine import DeeplabEngine


class DeeplabV3Plus:
    def __init__(self, weights='pascal_voc', input_tensor=None, input_shape=(512, 512, 3), classes=21, OS=16,
                 last_activation=None, load_weights=True, reshape_outputs=False, backbone="xception", alpha=1.,
                 weights_path=None, synthetic_seed=1234, precision=None):
        # same argument checks / messages as model.py:20-30
        if not (weights in {'pascal_voc', None}):
            raise ValueError('The `weights` argument should be either '
                             '`None` (random initialization) or `pascal_voc`'
                             '(pre-trained on PASCAL VOC)')
        if not (last_activation in {"softmax", "sigmoid", None}):
            raise ValueError("The last_activation parameter must be either None, softmax or sigmoid")
        if not (backbone in {"xception", "mobilenet"}):
            raise ValueError("Backbone must be either xception or mobilenet")
        if backbone == "xception" and OS not in (8, 16):
            raise ValueError("OS must be 8 or 16 (model.py:42-52 only distinguishes OS == 8)")
        if backbone == "mobilenet":
            OS = 8                                     # model.py:53-55: OS is set to 8 for the mobilenet backbone
        if input_tensor is not None:
            # model.py:57-62,114-116: the reference wires an existing Keras tensor in as the model's input.  There is no
            # graph to splice into here; what carries over is the tensor's static shape, which replaces input_shape
            # (anything with a .shape of rank 3 or 4 -- a Keras tensor, an ndarray, a torch tensor).
            shape = tuple(int(d) for d in tuple(input_tensor.shape)[-3:])
            if len(shape) != 3 or any(d <= 0 for d in shape):
                raise ValueError(f"input_tensor must have a static [.., H, W, C] shape, got {tuple(input_tensor.shape)}")
            input_shape = shape
        self.input_tensor = input_tensor
        self.weights = weights
        self.input_shape = tuple(input_shape)
        self.classes = classes
        self.last_activation = last_activation
        self.load_weights = load_weights
        self.reshape_outputs = reshape_outputs
        self.backbone = backbone
        self.alpha = alpha
        self.OS = OS
        self.weights_path = weights_path
        self.synthetic_seed = synthetic_seed
        self.precision = precision          # None -> $ASR_PRECISION or 'f32'; 'f16x3' = split-f16 MFMA GEMMs

    def build_model(self, only_DCNN_output=False, only_ASPP_output=False, first_upsample_size=(128, 128),
                    final_upsample=True, final_class_prediction=True):
        if self.backbone == "xception" and only_DCNN_output is True and only_ASPP_output is True:
            raise ValueError("Both only_DCNN_output and only_ASPP_output cannot be True at the same time")
        # model.py:80-93: the modified decoders exist for the xception backbone only (the flags are ignored for mobilenet)
        decoder = "full"
        if self.backbone == "xception":
            decoder = "dcnn" if only_DCNN_output else ("aspp" if only_ASPP_output else "full")
        # Seeded initialisation of the whole inventory: what Keras holds before load_weights (random init there).
        params = W.make_synthetic_weights(self.synthetic_seed, self.classes, backbone=self.backbone, alpha=self.alpha,
                                          decoder=decoder, class_prediction=final_class_prediction)
        path = self.weights_path          # an explicit argument only: no environment variable changes which weights a model has
        loaded_ok = False
        if self.load_weights and path:
            # local file only (never the URL of model.py:9); by_name=True, skip_mismatch=True like model.py:145
            template = params
            params, skipped = W.merge_by_name(template, W.load_weights(path))
            for name, why in skipped:
                print(f"asr_amd: layer variable {name} not loaded from {path} ({why}); keeping its seeded initialisation",
                      file=sys.stderr)
            frac = W.loaded_fraction(template, skipped)
            loaded_ok = frac > 0.0
            if frac < 0.5:      # e.g. an Xception checkpoint given to a MobileNet model: (nearly) nothing matched by name
                warnings.warn(f"DeeplabV3Plus: only {100.0 * frac:.0f} % of the model's kernels were found in {path} "
                              f"(backbone={self.backbone!r}, classes={self.classes}); the rest keeps its SEEDED SYNTHETIC "
                              "initialisation -- masks and IoUs are meaningless as segmentation results.", RuntimeWarning,
                              stacklevel=2)
        elif self.load_weights:
            # The reference would download the pretrained .h5 here (model.py:134-143): unavailable offline.
            warnings.warn("DeeplabV3Plus(load_weights=True) without weights_path: the pretrained checkpoint is a "
                          "network download in the reference and is NOT available here -- running on SEEDED SYNTHETIC "
                          "weights; masks and IoUs are meaningless as segmentation results (pass load_weights=False to "
                          "say so explicitly).", RuntimeWarning, stacklevel=2)
        return DeeplabModel(params, self.input_shape, self.classes, final_upsample, self.last_activation,
                            precision=self.precision, backbone=self.backbone, alpha=self.alpha, OS=self.OS,
                            reshape_outputs=self.reshape_outputs, decoder=decoder, first_upsample_size=first_upsample_size,
                            class_prediction=final_class_prediction,
                            calibrate=loaded_ok)           # weights from a file: their activation ranges are unknown


class DeeplabModel:
    """What ``build_model`` returns: only ``predict`` (and ``predict_device``) are used by callers
    (augmentation_utils.py:76)."""

    def __init__(self, params, input_shape, classes, final_upsample, last_activation, precision=None,
                 backbone="xception", alpha=1.0, OS=16, reshape_outputs=False, decoder="full",
                 first_upsample_size=(128, 128), class_prediction=True, calibrate=False):
        """calibrate: run DeeplabEngine.calibrate_range on the first two images of the first predict call (the range guard
        of the split-f16 GEMMs; the default for weights loaded from a file, whose activation ranges nobody has seen)."""
        self._calibrate_pending = bool(calibrate)
        self.input_shape = tuple(input_shape)
        self.reshape_outputs = reshape_outputs
        self.classes = classes
        self.final_upsample = final_upsample
        self.last_activation = last_activation
        self.backbone = backbone
        prefix = f"DLV3Plus-{backbone}-OS{OS if backbone == 'xception' else 8}"          # model.py:71, 85-106
        self.name = prefix + {"full": "", "dcnn": "-Only_DCNN_Output", "aspp": "-Only_ASPP_Output"}[decoder]
        if not class_prediction:
            self.name = prefix + "-no_class_prediction"
        self.engine = DeeplabEngine(params, classes, precision=precision, backbone=backbone, alpha=alpha, OS=OS,
                                    decoder=decoder, first_upsample_size=first_upsample_size,
                                    class_prediction=class_prediction)
        self.precision = self.engine.precision
        self.device = self.engine.device

    def predict_device(self, x, batch_size=16, profile=None, lane=0, clone=True):
        """x: [N,H,W,3] float32 (host array or device tensor) -> device tensor [N,h,w,classes].
        lane: activation pool to use (forward passes running concurrently on different streams need different lanes).
        clone=False (single-batch calls only): return the plan's own logits buffer instead of a copy -- valid until the next
        forward of the same (shape, lane)."""
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))
        n, h, w, c = x.shape
        if c != 3:
            raise ValueError(f"expected [N,H,W,3], got {tuple(x.shape)}")
        if self._calibrate_pending:
            self._calibrate_pending = False
            self.calibrate_range(x[:min(2, n)])
        outs = []
        for i in range(0, n, batch_size):
            xb = x[i:i + batch_size].to(self.device, non_blocking=True).contiguous()
            logits = self.engine.forward(xb, profile=profile, lane=lane)
            self._raw_logits = logits       # pre-activation, pre-upsample logits of the LAST batch (plan memory)
            if self.final_upsample:
                logits = self._upsample(logits, (h, w))
            elif clone or n > batch_size:
                logits = logits.clone()
            outs.append(logits)
        out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        if self.reshape_outputs:
            # model.py:120-122: Reshape((input_h * input_w, classes)) -- only meaningful with final_upsample, like the reference
            if out.shape[1] * out.shape[2] != self.input_shape[0] * self.input_shape[1]:
                raise ValueError("reshape_outputs needs the model output at the input size (final_upsample=True)")
            out = out.reshape(out.shape[0], self.input_shape[0] * self.input_shape[1], out.shape[-1])
        if self.last_activation in ("softmax", "sigmoid"):
            out = ops.class_activation(out.contiguous(), self.last_activation)
        return out

    def logits_of(self, preds, i):
        """Row i of the raw logits behind ``preds`` = the return value of the LAST single-batch predict_device call: preds[i]
        itself without a last_activation, else the plan's logits buffer (valid until the next forward of that shape and lane).
        The standard-output mask is taken from logits whatever the hot path's activation (generate_standard_output.py:82-86
        builds its model with last_activation=None)."""
        if self.last_activation in ("softmax", "sigmoid") and not self.final_upsample:
            return self._raw_logits[i]
        return preds[i]

    def calibrate_range(self, x):
        """Range guard of the split-f16 GEMMs on a probe batch x [B,H,W,3] (host array or device tensor): layers whose
        operands leave the f16 split's range move to the exact-f32 kernels (DeeplabEngine.calibrate_range).  Returns
        {layer: reason}."""
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))
        return self.engine.calibrate_range(x.to(self.device).contiguous())

    def check_range(self, x):
        """{layer: max |operand|} of the split-f16 layers that THIS input x [B,H,W,3] drives out of the f16 split's range
        (DeeplabEngine.range_report; empty = fine).  On demand: costs one opened-up forward pass of x, nothing per predict."""
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))
        return self.engine.range_report(x.to(self.device).contiguous())

    def _upsample(self, logits, hw):
        """Resizing(H, W, bilinear) of the logits (model.py:108-111); channels padded to a multiple
        of 4 for the 16-byte lanes of the kernel."""
        b, h, w, c = logits.shape
        cp = (c + 3) // 4 * 4
        padded = torch.zeros((b, h, w, cp), dtype=torch.float32, device=logits.device)
        padded[..., :c] = logits
        up = ops.resize_bilinear(padded, hw)
        return up[..., :c].contiguous()

    def predict(self, x, batch_size=16, verbose=0):
        """Keras ``Model.predict`` counterpart: host ndarray in, host ndarray out."""
        return self.predict_device(x, batch_size=batch_size).cpu().numpy()

    __call__ = predict_device
