#!/usr/bin/env python3
"""Benchmark of the Augmented Super-Resolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,4}] [--images M]

Workloads = BASELINE.json configs, all synthetic, inputs resident in HBM before the timed region:
  --config 1 (default, the configuration the metric is quoted on): 512x512 images, num_aug=100, angle +-0.15,
      shift +-80, argmax OPM, class 8, 50 AMSGrad SR iterations (lr 1e-3, decay 60/0.3, lambda = 1/.3/.7/0), plus
      max-SR and mean-SR, threshold and the 6 IoUs of SR_single_class.py.
  --config 2: the same with last_activation="softmax" (model.py:124-125) and the slice OPM
      (augmentation_utils.py:95-104): dense per-class float maps through the SR stage.
  --config 4: 1024x1024 images, num_aug=200 drawn once and pushed through the model in forward batches of 50
      (augmentation_utils.py:30-59 chunks for the same reason), model output 256x256, SR 2x (256 -> 512; D is the 2x2
      box mean); shifts are applied in the SR output frame (x 512/1024, pipeline.HotPath._sr_frame).
One STEP = one image = num_aug augmented copies through
    augment -> DeepLabV3+ Xception-65 forward (f32) -> OPM -> {ASR 50 iters, max-SR, mean-SR} -> threshold -> IoU counts.
Scaling: by default every rank processes K images of its own (weak scaling).  ``--images M`` is the strong-scaling form
of BASELINE configs[3] (SR_single_class.py:83-134 over a fixed image set): M images in all, image g on rank g mod N
(ragged shards), value = M * num_aug / time.  No data-path collective; one RCCL all-gather of the per-image IoU records at
the end (inside the timed region).

Prints ONE JSON line on rank 0: metric / value, `roofline` of the dominant kernel (HIP events on the launch stream in an
extra profiled step), `cpu_baseline` (the CPU oracle on the host cores, bounded sample), `parity` (HIP against that
oracle sample: the "IoU delta vs ref" half of BASELINE's metric) and, on the default single-GPU run, `configs`: the same
measurement of configs[2] and configs[4] (a few steps each) next to the headline.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ANGLE_MAX = 0.15
SHIFT_MAX = 80
CLASS_ID = 8
SR_ITERS = 50
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 matrix peak (spec), dense
HBM_PEAK_GBS = 8000.0
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak (spec)

# BASELINE.json configs -> concrete synthetic workloads.  feat = img / 4 (DeepLabV3+ without final upsample).
CONFIGS = {
    1: dict(img=512, out=512, num_aug=100, mode="argmax", activation=None, batch=100, steps=64, th_factor=0.2,
            what="BASELINE configs[1]: synthetic 512x512 images, num_aug=100, angle+-0.15 shift+-80, argmax OPM class 8"),
    2: dict(img=512, out=512, num_aug=100, mode="slice", activation="softmax", batch=100, steps=64, th_factor=0.65,
            what="BASELINE configs[2]: synthetic 512x512 images, num_aug=100, angle+-0.15 shift+-80, softmax + slice OPM "
                 "(dense float map of class 8, per-copy min-max normalised)"),
    4: dict(img=1024, out=512, num_aug=200, mode="argmax", activation=None, batch=50, steps=8, th_factor=0.2,
            what="BASELINE configs[4]: synthetic 1024x1024 images, num_aug=200 (one draw, forward batches of 50), "
                 "angle+-0.15 shift+-80 input px (x0.5 in the 512x512 SR frame), argmax OPM class 8, model output 256x256, "
                 "2x SR"),
}


def synth_image(rng, size=512, box=8):
    """U[0,1) noise low-pass filtered with a box filter so that masks have structure; the box scales with the
    image so that a 1024x1024 image has the structure of the 512x512 one."""
    box = box * size // 512 if size >= 512 else box
    x = rng.random((size + box, size + box, 3), dtype=np.float32)
    c = np.cumsum(np.cumsum(x, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0), (0, 0)))
    s = c[box:, box:] - c[:-box, box:] - c[box:, :-box] + c[:-box, :-box]
    s = s[:size, :size] / (box * box)
    s = (s - s.min()) / (s.max() - s.min())
    return np.ascontiguousarray(s, dtype=np.float32)


def unaugmented_logits(model, image_dev, batch):
    """Logits of the un-augmented image, computed as row 0 of a forward pass of `batch` identical copies: the setup passes
    then launch exactly the kernels (shapes, grids) of the timed steps, so a rocprofv3 kernel summary of a bench run holds
    one population of launches per layer (a batch-1 pass would mix 100x smaller launches into every average).  The rows of
    a forward pass do not depend on the batch size."""
    copies = image_dev[None].expand(batch, -1, -1, -1).contiguous()
    preds = model.predict_device(copies, batch_size=batch)
    return model.logits_of(preds, 0).clone()


def calibrate_class_bias(model, image_dev, class_id, fraction=0.3, batch=1):
    """Seeded synthetic weights never make class `class_id` the argmax on low-passed noise, which would hand the SR stage
    an empty problem.  Shift that class's logit bias (a synthetic parameter like all the others) so that it wins on
    `fraction` of the pixels of the un-augmented global image 0; every rank computes the same shift from the same image."""
    import torch
    logits = unaugmented_logits(model, image_dev, batch)
    other = logits.clone()
    other[..., class_id] = float("-inf")
    margin = (other.max(dim=-1).values - logits[..., class_id]).flatten()
    delta = float(torch.quantile(margin, fraction))
    model.engine.shift_logit_bias(class_id, delta)
    return delta


def model_gt(path, model, image_dev, class_id, ring=4, batch=1):
    """Ground truth consistent with the (synthetic) model: the standard-output mask of the un-augmented image
    (generate_standard_output.py:52-65), eroded by `ring` pixels, with a void (255) band of 2*ring pixels around it --
    the shape of a VOC label map (class blob, 255 border).  Setup only (torch slicing, outside the timed region)."""
    import torch
    std = path.standard_mask(unaugmented_logits(model, image_dev, batch), path.sr.output_size)
    m = (std == class_id)

    def spread(mask):                 # (2*ring+1)^2 box dilation, separable
        out = mask.clone()
        for axis in (0, 1):
            acc = out.clone()
            for d in range(1, ring + 1):
                fwd = torch.zeros_like(out)
                bwd = torch.zeros_like(out)
                if axis == 0:
                    fwd[d:], bwd[:-d] = out[:-d], out[d:]
                else:
                    fwd[:, d:], bwd[:, :-d] = out[:, :-d], out[:, d:]
                acc |= fwd | bwd
            out = acc
        return out

    dil, ero = spread(m), ~spread(~m)
    gt = torch.zeros_like(std, dtype=torch.int32)
    gt[dil] = 255
    gt[ero] = class_id
    return gt.contiguous()


def shifted_weights(weights, class_id, delta, name="logits_semantic"):
    """Host copy of the synthetic weights with the bench's class-bias shift applied: what the CPU oracle must use to see
    the model the device runs."""
    w = dict(weights)
    key = name + "/bias" if name + "/bias" in w else "custom_logits_semantic/bias"
    b = np.array(w[key], dtype=np.float32, copy=True)
    b[class_id] += np.float32(delta)
    w[key] = b
    return w


def cpu_baseline(weights, cfg, seed=1234):
    """The CPU oracle (TF-materialised formulation, torch-CPU/numpy f32) on a bounded sample of the same workload: 1 image,
    n copies through the model, SR with N=n for 2 iterations; the per-copy cost is extrapolated to the configuration's
    num_aug / 50 iterations and reported in the metric's unit.  Returns (record, sample) -- the sample (inputs, oracle
    logits, oracle masks) is what `parity` re-runs on the device."""
    import torch
    from oracle import augment as o_aug, sr as o_sr
    from oracle.model import OracleDeeplabV3Plus
    size, out = cfg["img"], cfg["out"]
    feat = size // 4
    n = 4 if size <= 512 else 2
    iters = 2
    rng = np.random.default_rng(seed)
    img = synth_image(rng, size)
    np.random.seed(seed)
    t0 = time.perf_counter()
    copies, angles, shifts = o_aug.create_augmented_copies(img, n, ANGLE_MAX, SHIFT_MAX)
    t_aug = time.perf_counter() - t0
    t0 = time.perf_counter()
    pred = OracleDeeplabV3Plus(weights, last_activation=cfg["activation"]).predict(copies, batch_size=n)
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    masks, _ = o_aug.opm(pred, CLASS_ID, cfg["mode"])
    if cfg["mode"] != "slice":      # load_SR_data's global min-max normalisation (superres_utils.py:183-206)
        stack = np.stack(masks)
        gmin, gmax = stack.min(), stack.max()
        masks = [np.asarray(o_sr.min_max_normalization(m, 0.0, 1.0, gmin, gmax), dtype=np.float32) for m in masks]
    t_opm = time.perf_counter() - t0
    sr_shifts = (shifts * np.float32(out / size)).astype(np.float32)

    def solver():
        opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        return o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n, optimizer=opt,
                                    feature_size=(feat, feat), output_size=(out, out))

    sr = solver()
    t0 = time.perf_counter()
    t_max, _ = sr.max_superresolution(masks, angles, sr_shifts)
    t_mean, _ = sr.mean_superresolution(masks, angles, sr_shifts)
    t_realign = time.perf_counter() - t0
    t0 = time.perf_counter()
    t_asr, _ = sr.augmented_superresolution(masks, angles, sr_shifts)
    t_sr = time.perf_counter() - t0
    per_copy = (t_aug + t_fwd + t_opm + t_realign) / n + (t_sr / (iters * n)) * SR_ITERS
    rec = {
        "value": round(1.0 / per_copy, 4), "unit": "augmented-copies/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": (f"CPU restatement of the reference (TF2 unavailable offline): 1 image {size}x{size}, {n} copies; "
                   f"augment {t_aug:.2f}s + model fwd {t_fwd:.2f}s + OPM {t_opm:.2f}s + max/mean-SR {t_realign:.2f}s, "
                   f"ASR {iters} iters at N={n} {t_sr:.2f}s extrapolated to {SR_ITERS} iters per copy"),
    }
    sample = dict(img=img, copies=copies, angles=angles, shifts=shifts, pred=pred, n=n, iters=iters,
                  targets={"aug": t_asr, "max": t_max, "mean": t_mean})
    return rec, sample


def parity(sample, cfg, model, make_sr, class_id, dev):
    """HIP path against the oracle sample of `cpu_baseline` (same image, same seed-1234 copies, same weights): the class
    maps of the forward pass, then the three SR masks and their IoUs against a ground truth derived from the ORACLE's
    un-augmented prediction.  BASELINE's metric: "IoU delta vs ref" (utils.py:207-230), bar 1e-3."""
    import torch
    from asr_amd import ops
    from asr_amd.pipeline import HotPath
    from oracle import augment as o_aug, sr as o_sr
    n, size, out = sample["n"], cfg["img"], cfg["out"]
    got = model.predict_device(sample["copies"], batch_size=n).cpu().numpy()
    ref = sample["pred"]
    rec = {"copies": n,
           "max_abs_output_diff": float(np.abs(got - ref).max()),
           "output": "softmax probabilities" if cfg["activation"] else "logits",
           "output_scale": float(np.abs(ref).max()),
           "argmax_agreement": float((got.argmax(-1) == ref.argmax(-1)).mean())}
    # ground truth for the IoUs: the oracle's own standard-output mask of copy 0 (upsample + argmax + class filter)
    from oracle import tf_ops
    up = tf_ops.resize_bilinear(ref[:1], (out, out)).numpy()[0]
    gt = np.where(up.argmax(-1) == class_id, class_id, 0).astype(np.int32)
    sr = make_sr(sample["iters"], n)
    path = HotPath(model, sr, class_id=class_id, mode=cfg["mode"], th_factor=cfg["th_factor"], batch_size=n)
    res = path.run_image(ops.to_device(sample["img"], device=dev), sample["angles"], sample["shifts"],
                         gt_dev=ops.to_device(gt, torch.int32, device=dev), adam_start=0)
    deltas, agree = {}, {}
    for t in ("aug", "max", "mean"):
        o_mask = o_sr.threshold_image(sample["targets"][t], class_id, th_factor=cfg["th_factor"])[..., 0]
        h_mask = res[t].cpu().numpy()
        agree[t] = float((o_mask == h_mask).mean())
        o_iou = o_aug.compute_IoU(gt, o_mask, img_size=(out, out), class_id=class_id)
        h_iou = o_aug.compute_IoU(gt, h_mask, img_size=(out, out), class_id=class_id)
        deltas[t] = None if (np.isnan(o_iou) or np.isnan(h_iou)) else abs(h_iou - o_iou)
    rec["mask_agreement"] = {k: round(v, 6) for k, v in agree.items()}
    rec["iou_delta_vs_oracle"] = {k: (None if v is None else float(f"{v:.3e}")) for k, v in deltas.items()}
    vals = [v for v in deltas.values() if v is not None]
    rec["iou_delta_max"] = float(f"{max(vals):.3e}") if vals else None
    rec["bar"] = "BASELINE north_star: IoU within 1e-3 of the reference path for identical augmentation seeds"
    rec["note"] = (f"oracle sample of cpu_baseline ({n} copies of one {size}x{size} image, {sample['iters']} ASR iterations); "
                   "ground truth = the oracle's standard-output mask of copy 0")
    return rec


def build_library_once():
    """Compile libasr_hip.so if it is missing -- under an exclusive file lock, so the ranks of one launch never run
    hipcc over the same objects at once (csrc/build.py takes the lock; the losers find the library up to date)."""
    from asr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as entry
        entry.build()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) with torch.distributed.run and
    relay their output.  This parent never touches the GPU (torch.cuda.device_count() does not initialise HIP on this
    image), it only builds the library, waits and returns the children's exit code.  With fewer visible devices than
    ranks the run is a REHEARSAL of the multi-rank path (gloo collectives, ranks share devices) and says so in its
    line; it is refused beyond 4 ranks (device memory and the per-card process limit of the GPU boxes)."""
    import socket
    import torch
    build_library_once()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        if ndev == 0:
            print(f"bench.py: --gpus {args.gpus} but no GPU is visible", file=sys.stderr)
            return 2
        if args.gpus > 4:
            print(f"bench.py: --gpus {args.gpus} but only {ndev} device(s) visible; a rehearsal on shared devices is limited "
                  f"to 4 ranks", file=sys.stderr)
            return 2
        env["ASR_DIST_BACKEND"] = "gloo"
        print(f"bench.py: {ndev} device(s) for {args.gpus} ranks -> rehearsal (gloo, shared devices)", file=sys.stderr)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# kernel family of the profile -> the translation unit its kernels are compiled from
FAMILY_UNIT = {"pw16": "gemm", "pw16s": "gemm", "pw": "gemm", "dw": "dwconv", "sepconv": "sepconv", "conv": "layers",
               "misc": "layers", "sr": "sr"}


def unit_hashes():
    """sha256 per translation unit of the library: its source, every shared header and its compile flags (csrc/build.py's
    COMMON + per-file flags, which decide the code objects as much as the sources do)."""
    import hashlib
    import importlib.util
    csrc = os.path.join(ROOT, "deeplabv3plus-augmented-superresolution_amd", "csrc")
    spec = importlib.util.spec_from_file_location("asr_csrc_build", os.path.join(csrc, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    hdr = hashlib.sha256()
    for f in b.HEADERS:
        with open(os.path.join(csrc, f), "rb") as fh:
            hdr.update(fh.read())
    out = {}
    for src, flags in b.SOURCES:
        h = hashlib.sha256(hdr.digest())
        h.update(" ".join(b.COMMON + flags + (["postpass"] if src in getattr(b, "POSTPASS", ()) else [])).encode())
        if src in getattr(b, "POSTPASS", ()):
            with open(os.path.join(csrc, "pk_postpass.py"), "rb") as fh:
                h.update(fh.read())
        with open(os.path.join(csrc, src), "rb") as fh:
            h.update(fh.read())
        out[os.path.splitext(src)[0]] = h.hexdigest()[:16]
    return out


def profile_provenance():
    """The committed PMC summaries (profiles/r*_pmc_{traffic,sq}*.json) are quoted only for kernels whose TRANSLATION UNIT is
    unchanged since they were taken: profiles/csrc.sha256 (written by tools/profile_round.sh next to the summaries) holds
    one hash per unit -- source + shared headers + compile flags -- and the commit; a change of sr.hip neither vouches for
    nor invalidates the GEMM counters."""
    now = unit_hashes()
    stamp = os.path.join(ROOT, "profiles", "csrc.sha256")
    rec = {"units": now, "profiled_units": {}, "profiled_at_commit": None}
    if os.path.exists(stamp):
        try:
            with open(stamp) as fh:
                old = json.load(fh)
            rec["profiled_units"] = old.get("units", {})
            rec["profiled_at_commit"] = old.get("commit")
        except ValueError:
            pass                                               # a stamp of an earlier format vouches for nothing
    rec["current"] = {u: rec["profiled_units"].get(u) == h for u, h in now.items()}
    return rec


class Workload:
    """One BASELINE config on this rank: model, hot path, resident inputs, step functions."""

    def __init__(self, cfg_id, args, rank, world, dev, weights, precision, bias_shift=None):
        import torch
        from asr_amd import distributed as D, ops
        from asr_amd.model import DeeplabModel
        from asr_amd.pipeline import HotPath
        self.cfg_id, self.cfg, self.args, self.rank, self.world, self.dev = cfg_id, CONFIGS[cfg_id], args, rank, world, dev
        cfg = self.cfg
        self.img, self.out, self.feat, self.num_aug = cfg["img"], cfg["out"], cfg["img"] // 4, cfg["num_aug"]
        self.batch = min(args.batch_size or cfg["batch"], self.num_aug)
        self.lanes = 1 if args.overlap else args.lanes
        self.model = DeeplabModel(weights, (self.img, self.img, 3), 21, final_upsample=False,
                                  last_activation=cfg["activation"], precision=precision)
        self.path = HotPath(self.model, self.make_sr(SR_ITERS, self.num_aug), class_id=CLASS_ID, mode=cfg["mode"],
                            th_factor=cfg["th_factor"], batch_size=self.batch)
        img0 = ops.to_device(synth_image(np.random.default_rng(1234), self.img), device=dev)
        if bias_shift is None:
            bias_shift = calibrate_class_bias(self.model, img0, CLASS_ID, batch=self.batch)
        else:
            self.model.engine.shift_logit_bias(CLASS_ID, bias_shift)
        self.bias_shift = bias_shift
        self.D = D

    def make_sr(self, iters, n):
        from asr_amd.superresolution_scripts.optimizer import Optimizer
        from asr_amd.superresolution_scripts.superresolution import Superresolution
        opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        return Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n, optimizer=opt,
                               feature_size=(self.feat, self.feat), output_size=(self.out, self.out))

    def prepare(self, timed_globals, warm_globals):
        """Resident inputs: the global images this rank touches (8 distinct images cycled by GLOBAL index: the same global
        image is the same data on any rank count), their model-consistent ground truth, the replayed RNG stream."""
        from asr_amd import ops
        self.timed, self.warm = list(timed_globals), list(warm_globals)
        everything = self.timed + self.warm
        self.params = self.D.replay_augmentation_stream(max(everything) + 1 if everything else 0, self.num_aug, ANGLE_MAX,
                                                        SHIFT_MAX, seed=1234)
        self.imgs, self.gts = {}, {}
        for j in sorted({g % 8 for g in everything}):
            self.imgs[j] = ops.to_device(synth_image(np.random.default_rng(1234 + j), self.img), device=self.dev)
            self.gts[j] = model_gt(self.path, self.model, self.imgs[j], CLASS_ID, batch=self.batch)
        for lane in range(self.lanes):                  # every lane's launch plan and activation pool exist before the
            self.model.engine.plan(self.batch, self.img, self.img, lane)     # timed region, whatever the warm-up count
            if self.num_aug % self.batch:
                self.model.engine.plan(self.num_aug % self.batch, self.img, self.img, lane)

    def _args(self, g):
        angles, shifts = self.params[g]
        return (self.imgs[g % 8], angles, shifts), dict(gt_dev=self.gts[g % 8],
                                                        adam_start=self.D.adam_start_step(g, SR_ITERS, self.cfg["mode"]))

    def step(self, g, profile=None):
        a, kw = self._args(g)
        return self.path.run_image(*a, profile=profile, **kw)

    def run(self, globals_):
        """The images `globals_` in order; returns their IoU records."""
        recs, args = [], self.args
        if self.lanes > 1:
            pending = []
            for i, g in enumerate(globals_):
                a, kw = self._args(g)
                pending.append(self.path.submit_lane(i % self.lanes, *a, **kw))
                if len(pending) >= self.lanes:              # keep `lanes` images in flight, collect in submission order
                    recs.append(pending.pop(0).result()["ious"])
            while pending:
                recs.append(pending.pop(0).result()["ious"])
            return recs
        if not args.overlap:
            return [self.step(g)["ious"] for g in globals_]
        pending = None
        for g in globals_:
            a, kw = self._args(g)
            h = self.path.submit_image(*a, **kw)
            if pending is not None:
                recs.append(pending.result()["ious"])
            pending = h
        if pending is not None:
            recs.append(pending.result()["ious"])
        return recs

    def close(self):
        import torch
        self.path = self.model = self.imgs = self.gts = None
        gc.collect()
        torch.cuda.empty_cache()


def rooflines(prof, provenance, cfg_id=1):
    """roofline objects from the HIP-event profile of one step (engine.forward(profile=...)): algorithmic flops / bytes of
    the launches of a kernel family / their summed launch durations.  PMC figures come from the summaries taken on THIS
    config (profiles/r*_pmc_traffic.json for configs[1], r*_pmc_traffic_cfg<N>.json for configs[N]; null when there is
    none) and only for kernel families whose translation unit is unchanged since (profile_provenance)."""
    import glob
    out = {}
    pmc, sq, pmc_path, sq_path = {}, {}, "", ""
    suffix = "" if cfg_id == 1 else f"_cfg{cfg_id}"
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_traffic{suffix}.json")))
    sq_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_sq{suffix}.json")))
    if pmc_files:       # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh), gfx950-corrected
        pmc_path = pmc_files[-1]
        with open(pmc_path) as fh:
            pmc = json.load(fh)
    if sq_files:        # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... pass (tools/pmc_sq.sh)
        sq_path = sq_files[-1]
        with open(sq_path) as fh:
            sq = json.load(fh)
    current = provenance.get("current", {})

    def vouched(kind):
        return bool(current.get(FAMILY_UNIT.get(kind, ""), False))

    def sq_lookup(kind, prefix):
        """Launch-time-weighted matrix-pipe occupancy of the entries whose kernel name starts with ``prefix``."""
        if not vouched(kind):
            return None
        hits = [(v, v["launches_sampled"] * v["avg_launch_us_profiled"]) for k, v in sq.items()
                if isinstance(v, dict) and k.startswith(prefix) and v.get("mfma_pipe_busy_at_2p4ghz") is not None]
        if not hits:
            return None
        tot = sum(w for _, w in hits)
        return {"mfma_pipe_busy_at_2p4ghz": round(sum(v["mfma_pipe_busy_at_2p4ghz"] * w for v, w in hits) / tot, 4),
                "mfma_util_rocprof_formula": round(sum(v["mfma_util_rocprof_formula"] * w for v, w in hits) / tot, 4),
                "source": f"profiles/{os.path.basename(sq_path)}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 pipes x launch time x "
                          "2.4 GHz) resp. / (GRBM_GUI_ACTIVE x 1024); counts K / N padding, unlike frac"}

    def pmc_lookup(kind, prefix):
        """Launch-weighted mean HBM bytes of the entries whose kernel name starts with ``prefix`` (template
        arguments in the name vary by build)."""
        if not vouched(kind):
            return None
        hits = [(v["hbm_bytes_per_launch"], v.get("launches_sampled", 1)) for k, v in pmc.items()
                if isinstance(v, dict) and k.startswith(prefix) and v.get("hbm_bytes_per_launch") is not None]
        return round(sum(b * n for b, n in hits) / sum(n for _, n in hits)) if hits else None

    def traffic_note(kind):
        unit = FAMILY_UNIT.get(kind, "?")
        if not pmc_path:
            return f"null: no PMC summary for configs[{cfg_id}] under profiles/ (tools/profile_round.sh takes one)"
        if vouched(kind):
            return (f"HBM bytes per launch from profiles/{os.path.basename(pmc_path)} (separate FETCH_SIZE / WRITE_SIZE passes, "
                    f"FETCH doubled per the gfx950 rule), taken on this build of {unit}.hip (unit hash "
                    f"{provenance['units'].get(unit)}, commit {provenance.get('profiled_at_commit')})")
        return (f"null: profiles/{os.path.basename(pmc_path)} was taken on another build of {unit}.hip (unit hash "
                f"{provenance['units'].get(unit)} now, {provenance.get('profiled_units', {}).get(unit)} profiled); re-run "
                "tools/profile_round.sh")

    def gemm_roofline(kind, kernel, pmc_key, peak, peak_note):
        ms, flops, nbytes, launches = prof[kind]
        achieved = flops / (ms * 1e-3) / 1e12
        return {
            "kernel": kernel, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "peak_note": peak_note,
            "mfma_util_pmc": sq_lookup(kind, pmc_key),
            "traffic": pmc_lookup(kind, pmc_key), "traffic_note": traffic_note(kind),
            "algorithmic_bytes_per_launch": round(nbytes / launches),
            "launches": launches, "avg_launch_ms": round(ms / launches, 4),
            "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
            "note": "algorithmic 2*M*K*N flop of the launches / HIP-event time around every launch of one extra "
                    "profiled step after the timed region",
        }

    if "pw16" in prof:
        out["roofline"] = gemm_roofline(
            "pw16", "pw_gemm_f16x3_pre_ring[_persist]_kernel / pw_gemm_f16x3_kernel (asr_pwconv_mfma_f16x3[_presplit], "
                    "v_mfma_f32_16x16x32_f16 x3 / v_mfma_f32_32x32x16_f16 x3)",
            "pw_gemm_f16x3", round(F16_MFMA_PEAK_TFLOPS / 3.0, 1),
            "dense f16 MFMA peak 2500 TFLOP/s / 3 MFMA products per f32-grade product")
        if "pw" in prof:
            out["roofline_f32_mfma_layers"] = gemm_roofline(
                "pw", "pw_gemm_kernel (layers with <= 32 output channels: the logits)", "pw_gemm_kernel<",
                F32_MFMA_PEAK_TFLOPS, "FP32 MFMA peak (spec)")
    elif "pw" in prof:
        out["roofline"] = gemm_roofline("pw", "pw_gemm_kernel (asr_pwconv_mfma_f32, v_mfma_f32_32x32x2_f32)",
                                        "pw_gemm_kernel<2, 2, 2, 2,", F32_MFMA_PEAK_TFLOPS,
                                        "FP32 MFMA peak 157.3 TFLOP/s (spec)")

    def hbm_roofline(kind, kernel, pmc_key, note):
        ms, _fl, by, launches = prof[kind]
        gbs = by / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_lookup(kind, pmc_key), "traffic_note": traffic_note(kind),
                "algorithmic_bytes_per_launch": round(by / launches), "bytes_note": note,
                "launches": launches, "avg_launch_ms": round(ms / launches, 4)}

    if "pw16s" in prof:
        out["roofline_small_n_layers"] = hbm_roofline(
            "pw16s", "pw_gemm_f16x3_kernel<2, 2, 2, 1> (asr_pwconv_mfma_f16x3 with <= 64 output channels: feature_projection0; "
                     "128 x 64 tile)", "pw_gemm_f16x3_kernel<2, 2, 2, 1,",
            "layer input + layer output + weights (these layers are HBM-bound: 256 -> 48 channels reads 5.3x what it writes)")
    if "dw" in prof:
        out["roofline_depthwise"] = hbm_roofline(
            "dw", "dw_stream_full_kernel / aspp_dw3_phase_kernel (asr_dwconv3x3_nhwc_[split_]f32, asr_aspp_dwconv3_nhwc_*)",
            "dw_stream_full_kernel<",
            "bytes moved: layer input once + layer output once (the fused ASPP kernel: input once + its three outputs)")
    if "sepconv" in prof:
        out["roofline_fused_sepconv"] = hbm_roofline(
            "sepconv", "sepconv_fused_kernel (asr_sepconv_fused_f16x3: depthwise -> LDS -> MFMA pointwise, entry-flow block 1)",
            "_ZN12_GLOBAL__N_120sepconv_fused_kernel",
            "layer input + layer output only (the depthwise tensor stays in LDS)")
    out["kernel_time_ms_per_step"] = {k: round(v[0], 3) for k, v in prof.items() if not k.startswith("_")}
    if "_sr_stage_ms" in prof:          # ASR solve + max / mean realign + thresholds + IoU counts of the image (one interval)
        out["kernel_time_ms_per_step"]["sr_stage"] = round(prof["_sr_stage_ms"], 3)
    return out


def measure(cfg_id, args, rank, world, dev, weights, steps, warmup, images=None, bias_shift=None, want_roofline=True,
            want_cpu=True, dump_table=None):
    """One workload, timed as the contract says: W warm-up steps, a barrier + synchronize, EXACTLY K steps (or the rank's
    shard of `images`), the all-gather, synchronize + barrier, MAX over ranks."""
    import torch
    from asr_amd import distributed as D
    wl = Workload(cfg_id, args, rank, world, dev, weights, args.precision, bias_shift)
    cfg = wl.cfg
    if images is None:            # weak scaling: K images per rank; global image g = step * world + rank
        timed = [s * world + rank for s in range(warmup, warmup + steps)]
        warm = [s * world + rank for s in range(warmup)]
        total_images = (warmup + steps) * world
        timed_count = steps * world
    else:                         # strong scaling: `images` in all, ragged shards; warm-up images lie beyond them
        timed = D.shard_indices(images, rank, world)
        warm = [images + s * world + rank for s in range(warmup)]
        total_images = images + warmup * world
        timed_count = images
    wl.prepare(timed, warm)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    warm_records = wl.run(warm)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    records = wl.run(timed)
    table = D.all_gather_iou(warm + timed, warm_records + records, total_images, device=dev)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = D.all_reduce_max(elapsed, dev)           # MAX over ranks

    collective = D.collective_info(dev)                 # after the timed region: one more (untimed) all-reduce
    k_rank = max(len(D.shard_indices(images, r, world)) for r in range(world)) if images is not None else steps
    copies_total = timed_count * wl.num_aug
    rehearsal = world > 1 and torch.distributed.get_backend() != "nccl"
    out = {
        "metric": f"augmented-copies/sec (fwd+realign+SR-iter) @{wl.img}x{wl.img}",
        "value": round(copies_total / elapsed, 3),
        "unit": "augmented-copies/s",
        "n_gpus": world,
        "steps": k_rank,
        "warmup": warmup,
        "ms_per_step": round(1000.0 * elapsed / k_rank, 3),
        "higher_is_better": True,
        "scaling": "weak" if images is None else "strong",
        "vs_baseline": None,
        "dtype": ("f32 (pointwise GEMMs: operands split into f16 hi+lo, hi*hi+hi*lo+lo*hi on f16 MFMA, f32 accumulate)"
                  if args.precision == "f16x3" else "f32"),
        "data": "synthetic",
        "config": {
            "workload": ((f"BASELINE configs[3]: {images} images sharded over {world} GPU(s), per image as " if images is not None
                          else "") + f"{cfg['what']}; ASR {SR_ITERS} AMSGrad iters + max-SR + mean-SR + threshold + 6 IoUs; step = 1 image = "
                         f"{wl.num_aug} copies; DeepLabV3+ Xception-65 OS16, f32 activations/accumulation, seeded synthetic "
                         f"weights (class-{CLASS_ID} logit bias shifted by {wl.bias_shift:+.4f} so that class {CLASS_ID} wins "
                         "30 % of image 0)"),
            "baseline_config": cfg_id,
            "precision": args.precision,
            "images_per_gpu": k_rank, "images_total": timed_count, "num_aug": wl.num_aug, "sr_iters": SR_ITERS,
            "forward_batch": wl.batch, "input": f"{wl.img}x{wl.img}", "sr": f"{wl.feat}x{wl.feat} -> {wl.out}x{wl.out}",
            "opm": cfg["mode"] + (" on softmax probabilities" if cfg["activation"] else ""),
            "th_factor": cfg["th_factor"],      # test_SR.py:44 (0.2) for the argmax masks, SR_single_class.py:32 (0.65) for float maps
            "parallelism": (f"images sharded over {world} GPU(s) (image g on rank g mod {world}), one all-gather of IoU "
                            "records"),
            "overlap": ("SR stage of image i on a side HIP stream under the forward pass of image i+1" if args.overlap else
                        (f"{wl.lanes} images in flight on alternating HIP streams" if wl.lanes > 1 else "none")),
            "ground_truth": "model-derived: the model's own standard-output mask of the un-augmented image, eroded by 4 px "
                            "inside an 8 px void band -- mean_ious are self-consistency figures, not segmentation quality",
        },
    }
    out["config"]["collective"] = {k: collective[k] for k in ("backend", "world_size", "ranks_joined")}
    out["collective"] = dict(collective, op="one all_gather of [ceil(images / ranks), 1 + 6] float64 IoU records per rank "
                                            "(+ the MAX all-reduce of the elapsed time)")
    if rehearsal:
        out["rehearsal"] = (f"{world} ranks on {torch.cuda.device_count()} device(s), gloo collectives: exercises sharding and the "
                            "all-gather, NOT a scaling measurement")
    if rank == 0:
        rows = table[sorted(timed_all(images, steps, warmup, world))]
        if dump_table:
            np.save(dump_table, table if images is None else table[:images])
        valid = rows[~np.isnan(rows[:, 2])]
        out["mean_ious"] = {k: (None if np.isnan(v) else round(v, 6))
                            for k, v in D.mean_ious(valid if len(valid) else rows).items()}
    # ---- roofline of the dominant kernel, HIP events on the launch stream: one extra profiled step ----
    if rank == 0 and want_roofline:
        prof = {}
        wl.step(timed[0] if timed else 0, profile=prof)
        torch.cuda.synchronize()
        out.update(rooflines(prof, profile_provenance(), cfg_id))
        kt = out["kernel_time_ms_per_step"]
        fwd = sum(v for k, v in kt.items() if k != "sr_stage")
        out["overlap"] = {
            "forward_kernels_ms": round(fwd, 3), "sr_stage_ms": kt.get("sr_stage"), "step_ms": out["ms_per_step"],
            "sum_over_step": round((fwd + (kt.get("sr_stage") or 0.0)) / out["ms_per_step"], 3),
            "note": "one image profiled alone after the timed region (launch by launch, HIP events) against the timed step of "
                    f"the {wl.lanes}-lane pipeline: sum_over_step >= 1 means the SR stage of one image ran under the forward pass "
                    "of the next (the step costs less than its parts)"}
    sample = None
    if rank == 0 and world == 1 and want_cpu:
        out["cpu_baseline"], sample = cpu_baseline(shifted_weights(weights, CLASS_ID, wl.bias_shift), cfg)
        out["parity"] = parity(sample, cfg, wl.model, wl.make_sr, CLASS_ID, dev)
    return out, wl


def timed_all(images, steps, warmup, world):
    """Global indices of the timed images of ALL ranks (rows of the gathered table that enter mean_ious)."""
    if images is not None:
        return list(range(images))
    return [s * world + r for s in range(warmup, warmup + steps) for r in range(world)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed images per GPU (default: 64; configs[4]: 8)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1, help="BASELINE.json configs index")
    ap.add_argument("--images", type=int, default=None,
                    help="strong scaling (BASELINE configs[3]): this many images IN ALL, image g on rank g mod N (ragged "
                         "shards); --steps is ignored")
    ap.add_argument("--batch-size", type=int, default=int(os.environ.get("ASR_BATCH", "0")) or None,
                    help="forward batch (default: 100 copies at 512x512, 50 at 1024x1024)")
    ap.add_argument("--precision", choices=["f16x3", "f32"], default=os.environ.get("ASR_PRECISION", "f16x3"),
                    help="pointwise GEMM arithmetic: f16x3 = split-f16 MFMA with f32 accumulation (f32-grade results), "
                         "f32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--overlap", action="store_true", help="run the SR stage of image i on a side HIP stream under the forward pass of image i+1")
    ap.add_argument("--lanes", type=int, default=2, help="images in flight on alternating HIP streams, each with its own "
                    "activation pool (1 = strictly sequential; 2 = the forward pass of one image runs under the SR stage of "
                    "the other and vice versa; results are bit-identical)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the extra exact-f32 (v_mfma_f32_32x32x2_f32) measurement")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the configs[2] / configs[4] measurements that the default single-GPU run appends")
    ap.add_argument("--dump-table", default=None, help="rank 0 writes the gathered [images, 6] IoU table (float64 .npy) here")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world_env} but --gpus {args.gpus}: launch one rank per GPU "
                 f"(python bench.py --gpus N starts them itself) -- refusing to report a line for the wrong rank count")
    if args.images is not None and args.images < args.gpus:
        sys.exit(f"bench.py: --images {args.images} < --gpus {args.gpus}: a rank would have nothing to time")

    import torch
    build_library_once()
    from asr_amd import distributed as D, weights as W

    rank, world, local_rank = D.init_from_env()
    dev = D.local_device(local_rank)
    torch.cuda.set_device(dev)
    rehearsal = world > 1 and torch.distributed.get_backend() != "nccl"
    if rehearsal:           # ranks share devices: keep (ranks per device) x lanes x 30 GB of activation pools inside the HBM
        per_dev = -(-world // max(torch.cuda.device_count(), 1))
        args.lanes = max(1, min(args.lanes, 4 // per_dev))

    steps = args.steps if args.steps is not None else CONFIGS[args.config]["steps"]
    weights = W.make_synthetic_weights(1234, 21)
    single = rank == 0 and world == 1
    out, wl = measure(args.config, args, rank, world, dev, weights, steps, args.warmup, images=args.images,
                      want_roofline=not args.no_roofline, want_cpu=not args.no_cpu_baseline, dump_table=args.dump_table)
    bias_shift = wl.bias_shift
    wl.close()
    del wl

    if single and args.precision != "f32" and not args.no_f32_line and args.config == 1 and args.images is None:
        # the same workload with every GEMM on the exact-f32 MFMA kernels, a few steps, next to the headline
        saved = args.precision
        args.precision = "f32"
        k32 = min(steps, 8)
        o32, w32 = measure(1, args, rank, world, dev, weights, k32, min(args.warmup, 2) or 1, bias_shift=bias_shift,
                           want_roofline=False, want_cpu=False)
        w32.close()
        del w32
        args.precision = saved
        out["config"]["exact_f32_copies_per_s"] = o32["value"]      # (top-level extras are dropped by the driver's record)
        out["config"]["exact_f32_ms_per_step"] = o32["ms_per_step"]
        out["exact_f32"] = {"value": o32["value"], "unit": "augmented-copies/s", "steps": k32, "ms_per_step": o32["ms_per_step"],
                            "note": "--precision f32: every pointwise GEMM on v_mfma_f32_32x32x2_f32, same workload and lanes"}
    if single and args.config == 1 and args.images is None and not args.no_extra_configs:
        # BASELINE's other single-GPU configurations, a few steps each, each with its own roofline / cpu_baseline / parity
        out["configs"] = {}
        for cid, k in ((2, min(steps, 8)), (4, min(steps, 4))):
            oc, wc = measure(cid, args, rank, world, dev, weights, k, 1, want_roofline=not args.no_roofline,
                             want_cpu=not args.no_cpu_baseline)
            wc.close()
            del wc
            out["configs"][str(cid)] = {key: oc[key] for key in
                                        ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "mean_ious",
                                         "roofline", "roofline_depthwise", "kernel_time_ms_per_step", "overlap", "collective", "cpu_baseline", "parity")
                                        if key in oc}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
