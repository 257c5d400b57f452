"""The whole hot path for one image, device-resident end to end:

    augment (N copies) -> DeepLabV3+ forward -> OPM -> {ASR solve, max-SR, mean-SR} -> threshold
    -> IoU counts

i.e. the body of test_SR.py:73-94 / generate_augmented_copies.py:88-91 + SR_single_class.py:83-127
without the host round trips (ndarray lists, HDF5) the reference puts between the stages.
Used by bench.py, the scripts and the smoke test.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .superresolution_scripts import augmentation_utils as au
from .utils import iou_from_counts


class HotPath:
    def __init__(self, model, superresolution, class_id=8, mode="argmax", th_factor=0.15, batch_size=16):
        self.model = model
        self.sr = superresolution
        self.class_id = class_id
        self.mode = mode
        self.th_factor = th_factor
        self.batch_size = batch_size
        self._side = None
        self._lanes = {}

    def _threshold(self, target, target_max, out=None):
        if target_max is not None:
            return ops.threshold(target, self.class_id, th_mask=target_max, out=out)
        return ops.threshold(target, self.class_id, th_factor=self.th_factor, out=out)

    MASK_KEYS = ("standard", "aug", "max", "mean")          # rows of the per-image mask buffer

    def standard_mask(self, logits0, out_hw, out=None):
        """generate_standard_output.py:52-65: final bilinear upsample + argmax + class filter, from the
        logits of the un-augmented copy 0 (one kernel)."""
        return ops.standard_mask(logits0.contiguous(), out_hw, self.class_id, out=out)

    # ---- stage 1 (model stream): augment -> forward -> OPM (-> standard mask) --------------------------
    def _stage_model(self, image_dev, angles, shifts, profile=None, want_standard=True, lane=0):
        """The copies go through the model one forward batch at a time (augmentation_utils.py:30-59 draws them in chunks
        for the same reason): each batch is augmented straight into the plan's input buffer, its logits are consumed in
        place by the OPM kernel, which writes its rows of the image's [N,h,w] stack -- nothing of size [N,H,W,3] or
        [N,h,w,classes] outlives a batch."""
        out_hw = self.sr.output_size
        n = len(angles)
        h, w, _ = image_dev.shape
        eng = self.model.engine
        bs = min(self.batch_size, n)
        y = ymax = None                     # [1, N, fh, fw], allocated from the first batch's logits: their size depends on the
        res = {"_masks": torch.empty((len(self.MASK_KEYS),) + tuple(out_hw), dtype=torch.int32, device=image_dev.device)}
        for i in range(0, n, bs):
            k = min(bs, n - i)
            copies = au.augment_on_device(image_dev, angles[i:i + k], shifts[i:i + k], out=eng.input_view(k, h, w, lane))
            preds = self.model.predict_device(copies, batch_size=k, profile=profile, lane=lane, clone=False)  # logits stay there
            if y is None:                   # decoder / upsampling options, not only on the backbone's stride
                y = torch.empty((1, n) + tuple(preds.shape[1:3]), dtype=torch.float32, device=image_dev.device)
                ymax = torch.empty_like(y) if self.mode == "slice_max" else None
            if i == 0 and want_standard:
                res["standard"] = self.standard_mask(self.model.logits_of(preds, 0), out_hw, out=res["_masks"][0])
            au.output_processing(preds, self.class_id, self.mode, out=y[0, i:i + k],
                                 out_max=ymax[0, i:i + k] if ymax is not None else None)
            del copies, preds
        if self.mode != "slice":            # load_SR_data's global min-max normalisation (superres_utils.py:183-192)
            y = self._normalise(y)
            if ymax is not None:
                ymax = self._normalise(ymax)
        return res, y, ymax

    def _sr_frame(self, image_dev, shifts):
        """The SR stage applies the copies' shifts in ITS pixel frame (superresolution.py:61-64 translates the HR estimate,
        of output_size).  The reference always runs with image size == output_size; when they differ (BASELINE configs[4]:
        1024 x 1024 inputs, 512 x 512 SR output) a shift of s input pixels is s * output_size / image_size output pixels
        (per axis; shifts are [dx, dy]).  Angles are frame-independent."""
        h, w, _ = image_dev.shape
        H, Wd = self.sr.output_size
        if (h, w) == (H, Wd):
            return shifts
        return (np.asarray(shifts, dtype=np.float32) * np.array([Wd / w, H / h], dtype=np.float32)).astype(np.float32)

    # ---- stage 2 (any stream): ASR solve, max / mean realign, threshold, IoU counts --------------------
    def _stage_sr(self, res, y, ymax, angles, shifts, gt_dev, adam_start, sr_types):
        sr = self.sr
        a, s = angles[None], shifts[None]
        both = None
        if "max" in sr_types and "mean" in sr_types:            # one pass over the copies serves both (bit-identical)
            both = (sr.realign_batch(y, a, s, "both"), sr.realign_batch(ymax, a, s, "both") if ymax is not None else None)
        for t in sr_types:
            if t == "aug":
                if adam_start is not None:
                    sr.optimizer.optimizer.iterations = adam_start
                tgt, _ = sr.augmented_superresolution_batch(y, a, s)
                tmax = sr.augmented_superresolution_batch(ymax, a, s)[0] if ymax is not None else None
            elif both is not None:
                k = 0 if t == "max" else 1
                tgt, tmax = both[0][k], (both[1][k] if both[1] is not None else None)
            else:
                tgt = sr.realign_batch(y, a, s, t)
                tmax = sr.realign_batch(ymax, a, s, t) if ymax is not None else None
            res[t] = self._threshold(tgt[0], tmax[0] if tmax is not None else None,
                                     out=res["_masks"][self.MASK_KEYS.index(t)])
        if gt_dev is not None:
            res["_iou_keys"], res["_iou_counts"] = self._iou_counts(res, gt_dev)
        return res

    def run_image(self, image_dev, angles, shifts, gt_dev=None, adam_start=None, profile=None,
                  sr_types=("aug", "max", "mean"), want_standard=True):
        """image_dev [H,W,3] float32 device; angles [N], shifts [N,2] float32 host arrays;
        gt_dev [H,W] int32 device labels (optional).  Returns dict of device masks (+ 6 IoUs)."""
        res, y, ymax = self._stage_model(image_dev, angles, shifts, profile, want_standard)
        if profile is not None:         # the SR stage (solve, realign, thresholds, IoU counts) as one HIP-event interval
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        res = self._stage_sr(res, y, ymax, angles, self._sr_frame(image_dev, shifts), gt_dev, adam_start, sr_types)
        if profile is not None:
            e1.record()
            torch.cuda.synchronize()
            profile["_sr_stage_ms"] = profile.get("_sr_stage_ms", 0.0) + e0.elapsed_time(e1)
        return self._finish(res)

    def submit_image(self, image_dev, angles, shifts, gt_dev=None, adam_start=None,
                     sr_types=("aug", "max", "mean"), want_standard=True):
        """Pipelined form: stage 1 on the current stream, stage 2 on a side HIP stream, so the latency-bound
        SR solve of image i runs under the MFMA-bound forward pass of image i+1.  Returns a handle whose
        .result() waits for the side stream and yields the same dict as run_image."""
        main = torch.cuda.current_stream()
        if self._side is None:
            # high priority: the SR stage is a chain of short launches that must slip in between the forward pass's
            # workgroups as they retire, not queue behind a whole GEMM grid
            self._side = torch.cuda.Stream(priority=-1)
        res, y, ymax = self._stage_model(image_dev, angles, shifts, None, want_standard)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            res = self._stage_sr(res, y, ymax, angles, self._sr_frame(image_dev, shifts), gt_dev, adam_start, sr_types)
            done = torch.cuda.Event()
            done.record(self._side)
        for t in [y, ymax] + [v for v in res.values() if isinstance(v, torch.Tensor)]:
            if t is not None:
                t.record_stream(self._side)          # allocated on the main stream, consumed on the side stream
        return _Pending(self, res, done, keep=(y, ymax, gt_dev))

    def submit_lane(self, lane, image_dev, angles, shifts, gt_dev=None, adam_start=None,
                    sr_types=("aug", "max", "mean"), want_standard=True):
        """Two-lane pipelining: the WHOLE image (both stages) runs on the HIP stream of `lane`, with that lane's own
        activation pool, so consecutive images submitted to alternating lanes overlap like two independent processes
        (the forward pass of one under the SR solve, realign and reductions of the other, and vice versa).  Results
        are bit-identical to run_image; .result() of the returned handle waits for the lane."""
        if lane not in self._lanes:
            self._lanes[lane] = torch.cuda.Stream()
        stream = self._lanes[lane]
        stream.wait_stream(torch.cuda.current_stream())        # inputs prepared on the caller's stream
        with torch.cuda.stream(stream):
            res, y, ymax = self._stage_model(image_dev, angles, shifts, None, want_standard, lane=lane)
            res = self._stage_sr(res, y, ymax, angles, self._sr_frame(image_dev, shifts), gt_dev, adam_start, sr_types)
            done = torch.cuda.Event()
            done.record(stream)
        return _Pending(self, res, done, keep=(y, ymax, gt_dev, image_dev))

    def _finish(self, res):
        res.pop("_masks", None)          # the rows stay alive through the per-key views
        if "_iou_counts" in res:
            res["ious"] = self._ious_from_counts(res.pop("_iou_keys"), res.pop("_iou_counts").cpu().numpy())
        return res

    @staticmethod
    def _normalise(stack):
        """load_SR_data's global min-max normalisation of an image's masks to [0, 1] (superres_utils.py:183-206)."""
        return ops.minmax_normalize(stack.contiguous(), segments=1, new_min=0.0, new_max=1.0)

    def _iou_counts(self, res, gt_dev):
        """Integer intersection / union counts of every produced mask against the ground truth (device): one launch over
        the image's mask buffer; rows of masks that were not asked for are ignored."""
        keys = [k for k in self.MASK_KEYS if k in res]
        rows = [self.MASK_KEYS.index(k) for k in keys]
        masks = res["_masks"]
        if rows != list(range(len(self.MASK_KEYS))):                    # a subset of the four masks: compact it
            masks = masks[rows].contiguous()
        gt = gt_dev if gt_dev.dtype == torch.int32 else gt_dev.to(torch.int32)
        return keys, ops.iou_counts_shared_truth(gt.contiguous(), masks, self.class_id, include_bg=True)

    @staticmethod
    def _ious_from_counts(keys, counts):
        """[standard_single, standard_bg, aug_single, aug_bg, max, mean] (SR_single_class.py:109-120)."""
        by = dict(zip(keys, counts))
        nan = float("nan")

        def iou(k, bg):
            return iou_from_counts(by[k], bg) if k in by else nan

        return np.array([iou("standard", False), iou("standard", True), iou("aug", False), iou("aug", True),
                         iou("max", False), iou("mean", False)], dtype=np.float64)


class _Pending:
    """Handle of an image whose SR stage is still running on the side stream."""

    def __init__(self, path, res, done, keep):
        self._path, self._res, self._done, self._keep = path, res, done, keep

    def result(self):
        self._done.synchronize()
        self._keep = None
        return self._path._finish(self._res)
