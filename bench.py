#!/usr/bin/env python3
"""Benchmark of the Augmented Super-Resolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 512x512 images, num_aug=100, angle +-0.15,
shift +-80, argmax OPM, class 8, 50 AMSGrad SR iterations (lr 1e-3, decay 60/0.3,
lambda = 1/.3/.7/0), plus max-SR and mean-SR, threshold and the 6 IoUs of SR_single_class.py.
One STEP = one image = 100 augmented copies through
    augment -> DeepLabV3+ Xception-65 forward (f32) -> OPM -> {ASR 50 iters, max-SR, mean-SR}
    -> threshold -> IoU counts,
with the image and its ground truth resident in HBM before the timed region.  Default K = 64
(the whole 64-image configuration).  For N > 1 (torch.distributed.run, one rank per GPU) every
rank processes K images of its own (weak scaling), no data-path collective, one RCCL all-gather
of the per-image IoU records at the end (inside the timed region).

Prints ONE JSON line on rank 0: metric/value (+ roofline of the dominant kernel, measured with
HIP events on the launch stream in an extra profiled step, + cpu_baseline = the CPU oracle timed
on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

IMG = 512
FEAT = 128
NUM_AUG = 100
ANGLE_MAX = 0.15
SHIFT_MAX = 80
CLASS_ID = 8
SR_ITERS = 50
TH_FACTOR = 0.2
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 matrix peak (spec), dense
HBM_PEAK_GBS = 8000.0
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak (spec)


def synth_image(rng, size=IMG, box=8):
    """U[0,1) noise low-pass filtered with a box filter so that masks have structure."""
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
    import torch
    copies = image_dev[None].expand(batch, -1, -1, -1).contiguous()
    return model.predict_device(copies, batch_size=batch)[0]


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


def cpu_baseline(weights, seed=1234):
    """The CPU oracle (TF-materialised formulation, torch-CPU/numpy f32) on a bounded sample of the
    same workload: 1 image 512x512, 4 copies through the model, SR with N=4 for 2 iterations; the
    per-copy cost is extrapolated to N=100 / 50 iterations and reported in the metric's unit."""
    import torch
    from oracle import augment as o_aug, sr as o_sr
    from oracle.model import OracleDeeplabV3Plus
    n = 4
    iters = 2
    rng = np.random.default_rng(seed)
    img = synth_image(rng)
    np.random.seed(seed)
    t0 = time.perf_counter()
    copies, angles, shifts = o_aug.create_augmented_copies(img, n, ANGLE_MAX, SHIFT_MAX)
    t_aug = time.perf_counter() - t0
    t0 = time.perf_counter()
    pred = OracleDeeplabV3Plus(weights).predict(copies, batch_size=n)
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    masks, _ = o_aug.opm(pred, CLASS_ID, "argmax")
    t_opm = time.perf_counter() - t0
    opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = o_sr.Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=iters, num_aug=n, optimizer=opt,
                              feature_size=(FEAT, FEAT), output_size=(IMG, IMG))
    t0 = time.perf_counter()
    sr.max_superresolution(masks, angles, shifts)
    sr.mean_superresolution(masks, angles, shifts)
    t_realign = time.perf_counter() - t0
    t0 = time.perf_counter()
    sr.augmented_superresolution(masks, angles, shifts)
    t_sr = time.perf_counter() - t0
    per_copy = (t_aug + t_fwd + t_opm + t_realign) / n + (t_sr / (iters * n)) * SR_ITERS
    return {
        "value": round(1.0 / per_copy, 4), "unit": "augmented-copies/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": (f"CPU restatement of the reference (TF2 unavailable offline): 1 image 512x512, {n} copies; "
                   f"augment {t_aug:.2f}s + model fwd {t_fwd:.2f}s + OPM {t_opm:.2f}s + max/mean-SR {t_realign:.2f}s, "
                   f"ASR {iters} iters at N={n} {t_sr:.2f}s extrapolated to {SR_ITERS} iters per copy"),
    }


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
    import subprocess
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=int(os.environ.get("ASR_BATCH", "100")))
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
    ap.add_argument("--dump-table", default=None, help="rank 0 writes the gathered [images, 6] IoU table (float64 .npy) here")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world_env} but --gpus {args.gpus}: launch one rank per GPU "
                 f"(python bench.py --gpus N starts them itself) -- refusing to report a line for the wrong rank count")

    import torch
    from asr_amd import _lib
    build_library_once()
    from asr_amd import distributed as D, weights as W, ops
    from asr_amd.model import DeeplabModel
    from asr_amd.pipeline import HotPath
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution

    rank, world, local_rank = D.init_from_env()
    dev = D.local_device(local_rank)
    torch.cuda.set_device(dev)
    rehearsal = world > 1 and torch.distributed.get_backend() != "nccl"
    if rehearsal:           # ranks share devices: keep (ranks per device) x lanes x 30 GB of activation pools inside the HBM
        per_dev = -(-world // max(torch.cuda.device_count(), 1))
        args.lanes = max(1, min(args.lanes, 4 // per_dev))

    K, Wm = args.steps, args.warmup
    per_rank = K + Wm
    total_images = per_rank * world
    weights = W.make_synthetic_weights(1234, 21)

    def make_path(precision):
        mdl = DeeplabModel(weights, (IMG, IMG, 3), 21, final_upsample=False, last_activation=None, precision=precision)
        opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
        sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=SR_ITERS, num_aug=NUM_AUG, optimizer=opt,
                             feature_size=(FEAT, FEAT), output_size=(IMG, IMG))
        return mdl, HotPath(mdl, sr, class_id=CLASS_ID, mode="argmax", th_factor=TH_FACTOR, batch_size=args.batch_size)

    model, path = make_path(args.precision)
    cur = {"path": path}                              # the path the step closures drive (swapped for the exact-f32 line)

    # Synthetic inputs, resident in HBM before the timed region.  Global image g = step * world + rank;
    # every rank replays the reference's sequential RNG stream and keeps its own draws.
    params = D.replay_augmentation_stream(total_images, NUM_AUG, ANGLE_MAX, SHIFT_MAX, seed=1234)
    my_globals = [s * world + rank for s in range(per_rank)]
    distinct = 8                                     # 8 distinct images, cycled by GLOBAL index (content does not change the
    imgs, gts = {}, {}                               # work; the same global image is the same data on any rank count)
    setup_batch = min(args.batch_size, NUM_AUG)
    bias_shift = calibrate_class_bias(model, ops.to_device(synth_image(np.random.default_rng(1234)), device=dev), CLASS_ID,
                                      batch=setup_batch)
    for j in sorted({g % distinct for g in my_globals}):
        rng = np.random.default_rng(1234 + j)
        imgs[j] = ops.to_device(synth_image(rng), device=dev)
        gts[j] = model_gt(path, model, imgs[j], CLASS_ID, batch=setup_batch)

    def step(i, profile=None):
        g = my_globals[i]
        angles, shifts = params[g]
        return cur["path"].run_image(imgs[my_globals[i] % distinct], angles, shifts, gt_dev=gts[my_globals[i] % distinct],
                              adam_start=D.adam_start_step(g, SR_ITERS), profile=profile)

    def submit(i):
        """Pipelined step: forward pass on the main stream, SR stage on a side HIP stream (it overlaps the
        next image's forward pass); the per-image result is collected one step later."""
        g = my_globals[i]
        angles, shifts = params[g]
        return cur["path"].submit_image(imgs[my_globals[i] % distinct], angles, shifts, gt_dev=gts[my_globals[i] % distinct],
                                 adam_start=D.adam_start_step(g, SR_ITERS))

    def submit_lane(i):
        g = my_globals[i]
        angles, shifts = params[g]
        return cur["path"].submit_lane(i % args.lanes, imgs[my_globals[i] % distinct], angles, shifts, gt_dev=gts[my_globals[i] % distinct],
                                adam_start=D.adam_start_step(g, SR_ITERS))

    def run_steps(first, count):
        recs = []
        if args.lanes > 1 and not args.overlap:
            pending = []
            for i in range(first, first + count):
                pending.append(submit_lane(i))
                if len(pending) >= args.lanes:              # keep `lanes` images in flight, collect in submission order
                    recs.append(pending.pop(0).result()["ious"])
            while pending:
                recs.append(pending.pop(0).result()["ious"])
            return recs
        if not args.overlap:
            for i in range(first, first + count):
                recs.append(step(i)["ious"])
            return recs
        pending = None
        for i in range(first, first + count):
            h = submit(i)
            if pending is not None:
                recs.append(pending.result()["ious"])
            pending = h
        if pending is not None:
            recs.append(pending.result()["ious"])
        return recs

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def make_plans(mdl):
        for lane in range(args.lanes if not args.overlap else 1):   # every lane's launch plan and activation pool exist
            fb = min(args.batch_size, NUM_AUG)                      # before the timed region, whatever the warm-up count
            mdl.engine.plan(fb, IMG, IMG, lane)
            if NUM_AUG % fb:
                mdl.engine.plan(NUM_AUG % fb, IMG, IMG, lane)

    make_plans(model)
    warm_records = run_steps(0, Wm)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    records = run_steps(Wm, K)
    table = D.all_gather_iou(my_globals, warm_records + records, total_images, device=dev)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = D.all_reduce_max(elapsed, dev)           # MAX over ranks

    copies_total = K * NUM_AUG * world
    out = {
        "metric": "augmented-copies/sec (fwd+realign+SR-iter) @512x512",
        "value": round(copies_total / elapsed, 3),
        "unit": "augmented-copies/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": round(1000.0 * elapsed / K, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32 (pointwise GEMMs: operands split into f16 hi+lo, hi*hi+hi*lo+lo*hi on f16 MFMA, f32 accumulate)"
                  if args.precision == "f16x3" else "f32"),
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]: synthetic 512x512 images, num_aug=100, angle+-0.15 shift+-80, argmax OPM "
                         "class 8, ASR 50 AMSGrad iters + max-SR + mean-SR + threshold + 6 IoUs; step = 1 image = 100 copies; "
                         "DeepLabV3+ Xception-65 OS16, f32 activations/accumulation, seeded synthetic weights (class-8 logit bias "
                         f"shifted by {bias_shift:+.4f} so that class 8 wins 30 % of image 0); ground truth = the model's own "
                         "standard-output mask eroded by 4 px inside an 8 px void band"),
            "precision": args.precision,
            "images_per_gpu": K, "num_aug": NUM_AUG, "sr_iters": SR_ITERS, "forward_batch": args.batch_size,
            "parallelism": f"images sharded over {world} GPU(s), one all-gather of IoU records",
            "overlap": ("SR stage of image i on a side HIP stream under the forward pass of image i+1" if args.overlap else
                        (f"{args.lanes} images in flight on alternating HIP streams" if args.lanes > 1 else "none")),
        },
    }
    if rehearsal:
        out["rehearsal"] = (f"{world} ranks on {torch.cuda.device_count()} device(s), gloo collectives: exercises sharding and the "
                            "all-gather, NOT a scaling measurement")
    if rank == 0:
        if args.dump_table:
            np.save(args.dump_table, table)
        valid = table[~np.isnan(table[:, 2])]
        out["mean_ious"] = {k: (None if np.isnan(v) else round(v, 6))
                            for k, v in D.mean_ious(valid if len(valid) else table).items()}

    # ---- roofline of the dominant kernel (pointwise FP32-MFMA GEMM), HIP events on the launch stream ----
    if rank == 0 and not args.no_roofline:
        prof = {}
        step(Wm, profile=prof)
        torch.cuda.synchronize()
        pmc = {}
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        pmc_path = pmc_files[-1] if pmc_files else ""          # the newest round's PMC summary
        if pmc_path:       # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh), gfx950-corrected
            with open(pmc_path) as fh:
                pmc = json.load(fh)

        sq_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")))
        sq = {}
        if sq_files:       # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... pass (tools/pmc_sq.sh)
            with open(sq_files[-1]) as fh:
                sq = json.load(fh)

        def sq_lookup(prefix):
            """Launch-time-weighted matrix-pipe occupancy of the entries whose kernel name starts with ``prefix``."""
            hits = [(v, v["launches_sampled"] * v["avg_launch_us_profiled"]) for k, v in sq.items()
                    if k.startswith(prefix) and v.get("mfma_pipe_busy_at_2p4ghz") is not None]
            if not hits:
                return None
            tot = sum(w for _, w in hits)
            return {"mfma_pipe_busy_at_2p4ghz": round(sum(v["mfma_pipe_busy_at_2p4ghz"] * w for v, w in hits) / tot, 4),
                    "mfma_util_rocprof_formula": round(sum(v["mfma_util_rocprof_formula"] * w for v, w in hits) / tot, 4),
                    "source": f"profiles/{os.path.basename(sq_files[-1])}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 pipes x launch time x "
                              "2.4 GHz) resp. / (GRBM_GUI_ACTIVE x 1024); counts K / N padding, unlike frac"}

        def pmc_lookup(prefix):
            """Launch-weighted mean HBM bytes of the entries whose kernel name starts with ``prefix`` (template
            arguments in the name vary by build)."""
            hits = [(v["hbm_bytes_per_launch"], v.get("launches_sampled", 1)) for k, v in pmc.items()
                    if k.startswith(prefix) and v.get("hbm_bytes_per_launch") is not None]
            return round(sum(b * n for b, n in hits) / sum(n for _, n in hits)) if hits else None

        def gemm_roofline(kind, kernel, pmc_key, peak, peak_note):
            ms, flops, nbytes, launches = prof[kind]
            achieved = flops / (ms * 1e-3) / 1e12
            return {
                "kernel": kernel, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "peak_note": peak_note,
                "mfma_util_pmc": sq_lookup(pmc_key),
                "traffic": pmc_lookup(pmc_key),
                "traffic_note": f"HBM bytes per launch from profiles/{os.path.basename(pmc_path)} (separate FETCH_SIZE / "
                                "WRITE_SIZE passes, FETCH doubled per the gfx950 rule)",
                "algorithmic_bytes_per_launch": round(nbytes / launches),
                "launches": launches, "avg_launch_ms": round(ms / launches, 4),
                "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
                "note": "algorithmic 2*M*K*N flop of the launches / HIP-event time around every launch of one extra "
                        "profiled step after the timed region",
            }

        if "pw16" in prof:
            out["roofline"] = gemm_roofline(
                "pw16", "pw_gemm_f16x3_pre_kernel / pw_gemm_f16x3_kernel (asr_pwconv_mfma_f16x3[_presplit], v_mfma_f32_16x16x32_f16 x3 / v_mfma_f32_32x32x16_f16 x3)",
                "pw_gemm_f16x3",
                round(F16_MFMA_PEAK_TFLOPS / 3.0, 1),
                "dense f16 MFMA peak 2500 TFLOP/s / 3 MFMA products per f32-grade product")
            if "pw" in prof:
                out["roofline_f32_mfma_layers"] = gemm_roofline(
                    "pw", "pw_gemm_kernel (layers with <= 64 output channels)", "pw_gemm_kernel<2, 2, 2, 1,",
                    F32_MFMA_PEAK_TFLOPS, "FP32 MFMA peak (spec)")
        else:
            out["roofline"] = gemm_roofline("pw", "pw_gemm_kernel (asr_pwconv_mfma_f32, v_mfma_f32_32x32x2_f32)",
                                            "pw_gemm_kernel<2, 2, 2, 2,", F32_MFMA_PEAK_TFLOPS,
                                            "FP32 MFMA peak 157.3 TFLOP/s (spec)")
        if "dw" in prof:
            dms, _dfl, dby, dl = prof["dw"]
            gbs = dby / (dms * 1e-3) / 1e9
            out["roofline_depthwise"] = {
                "kernel": "dw_stream_full_kernel / aspp_dw3_kernel (asr_dwconv3x3_nhwc_f32, asr_aspp_dwconv3_nhwc_f32)", "bound": "hbm",
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": pmc_lookup("dw_stream_full_kernel<"), "algorithmic_bytes_per_launch": round(dby / dl),
                "traffic_note": "launch-weighted mean HBM bytes of the dw_stream_full_kernel launches (PMC, as above)",
                "launches": dl, "avg_launch_ms": round(dms / dl, 4),
            }
        if "sepconv" in prof:
            sms, _sfl, sby, sl = prof["sepconv"]
            gbs = sby / (sms * 1e-3) / 1e9
            out["roofline_fused_sepconv"] = {
                "kernel": "sepconv_fused_kernel (asr_sepconv_fused_f16x3: depthwise -> LDS -> MFMA pointwise, entry-flow block 1)",
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": pmc_lookup("_ZN12_GLOBAL__N_120sepconv_fused_kernel"), "algorithmic_bytes_per_launch": round(sby / sl),
                "traffic_note": "algorithmic bytes = layer input + layer output only (the depthwise tensor stays in LDS); PMC as above",
                "launches": sl, "avg_launch_ms": round(sms / sl, 4),
            }
        out["kernel_time_ms_per_step"] = {k: round(v[0], 3) for k, v in prof.items() if not k.startswith("_")}
    if rank == 0 and world == 1 and args.precision != "f32" and not args.no_f32_line:
        # the same workload with every GEMM on the exact-f32 MFMA kernels, a few steps, next to the headline
        m32, p32 = make_path("f32")
        m32.engine.shift_logit_bias(CLASS_ID, bias_shift)
        make_plans(m32)
        cur["path"] = p32
        k32 = min(K, 8)
        run_steps(0, min(Wm, 2) or 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(Wm, k32)
        torch.cuda.synchronize()
        e32 = time.perf_counter() - t0
        out["exact_f32"] = {"value": round(k32 * NUM_AUG / e32, 3), "unit": "augmented-copies/s", "steps": k32,
                            "ms_per_step": round(1000.0 * e32 / k32, 3),
                            "note": "--precision f32: every pointwise GEMM on v_mfma_f32_32x32x2_f32, same workload and lanes"}
        cur["path"] = path
        del m32, p32
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(weights)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
