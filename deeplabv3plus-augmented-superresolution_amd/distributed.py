"""Image-level sharding over the GPUs of one node and the single collective of the path.

The reference is single-process: it loops over images (generate_augmented_copies.py:88-91,
SR_single_class.py:83) appending per-image IoUs to Python lists and finishes with ``np.mean``
(SR_single_class.py:122-134).  Images are independent, so here image i belongs to rank
i mod world_size, with NO collective on the data path; the list-append + mean becomes ONE
all-gather of the per-image IoU records (<= a few KB, latency-bound) over RCCL/xGMI
(``backend="nccl"`` is RCCL on ROCm; ``gloo`` on CPU for tests).

Identical augmentation seeds under sharding: every rank replays the reference's sequential numpy
RNG stream for ALL images and keeps its own slice; the reference's persistent Adam step counter
 is reproduced by seeding ``iterations = image_index * num_iter * solves_per_image``.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from .superresolution_scripts.augmentation_utils import draw_augmentation_parameters

IOU_FIELDS = ("standard_single", "standard_bg", "aug_single", "aug_bg", "max", "mean")


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # ASR_DIST_BACKEND=gloo rehearses the multi-rank path on a single-GPU box (RCCL refuses two ranks on one
            # device); on a real node the default is nccl (= RCCL over xGMI).
            backend = os.environ.get("ASR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def local_device(local_rank):
    """cuda:<local_rank>.  One rank per GPU is the rule: a LOCAL_RANK beyond the visible devices is an error, except in
    a declared rehearsal (gloo collectives, ASR_DIST_BACKEND=gloo), where ranks wrap onto the visible devices."""
    n = torch.cuda.device_count()
    if local_rank >= n:
        rehearsal = dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl"
        if not rehearsal or n == 0:
            raise RuntimeError(f"LOCAL_RANK={local_rank} but only {n} device(s) are visible (one rank per GPU)")
        return torch.device("cuda", local_rank % n)
    return torch.device("cuda", local_rank)


def collective_device(device):
    """Where collective payloads must live: the GPU for RCCL, host memory for gloo."""
    return device if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")


def all_reduce_max(value, device):
    """MAX over ranks of a host float (the bench's elapsed time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def collective_info(device):
    """What actually carried the collectives of this run: the backend torch.distributed was initialised with, the world
    size it reports, and the number of ranks that answered one all-reduce(SUM) of a 1 -- on a real node "nccl" (= RCCL over
    xGMI) with ranks_joined == n_gpus; "none" / 1 in a single process."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": "none", "world_size": 1, "ranks_joined": 1, "payload_device": "none"}
    dev = collective_device(device)
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_joined": int(one.item()),
            "payload_device": dev.type}


def shard_indices(num_images, rank, world):
    """Global image indices owned by ``rank`` (round-robin, like dealing the reference's loop)."""
    return list(range(rank, num_images, world))


def replay_augmentation_stream(num_images, num_aug, angle_max, shift_max, seed=1234):
    """The (angles, shifts) every image would get from the reference's sequential global RNG
    (np.random.seed once at import, generate_augmented_copies.py:41-44)."""
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        out = [draw_augmentation_parameters(num_aug, angle_max, shift_max) for _ in range(num_images)]
    finally:
        np.random.set_state(state)
    return out


def adam_start_step(image_index, num_iter, mode="argmax"):
    """Global Adam ``iterations`` before the reference's solve of image ``image_index``."""
    return image_index * num_iter * (2 if mode == "slice_max" else 1)


def all_gather_rows(local_indices, local_rows, num_rows, width, device=None):
    """local_rows: [n_local, width] float64 rows of the global indices local_indices.  Returns the full
    [num_rows, width] table (NaN where no rank reported) on every rank via ONE all_gather of equal-sized slots."""
    rec = np.asarray(local_rows, dtype=np.float64).reshape(-1, width)
    idx = np.asarray(local_indices, dtype=np.int64)
    assert rec.shape[0] == idx.shape[0]
    table = np.full((num_rows, width), np.nan)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        table[idx] = rec
        return table
    world = dist.get_world_size()
    cap = -(-num_rows // world)                       # equal-sized slots: ceil(rows / world)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    device = collective_device(device)
    slot = torch.full((cap, 1 + width), -1.0, dtype=torch.float64)
    slot[:len(idx), 0] = torch.from_numpy(idx.astype(np.float64))
    slot[:len(idx), 1:] = torch.from_numpy(rec)
    slot = slot.to(device)
    gathered = [torch.empty_like(slot) for _ in range(world)]
    dist.all_gather(gathered, slot)
    for g in gathered:
        g = g.cpu().numpy()
        valid = g[:, 0] >= 0
        table[g[valid, 0].astype(np.int64)] = g[valid, 1:]
    return table


def all_gather_iou(local_indices, local_records, num_images, device=None):
    """local_records: [n_local, 6] float64 IoUs of the images in local_indices.  Returns the full
    [num_images, 6] table (NaN where no rank reported) on every rank via one all_gather."""
    return all_gather_rows(local_indices, local_records, num_images, len(IOU_FIELDS), device)


def mean_ious(table):
    """np.mean over images of each IoU column (SR_single_class.py:129-134); like the reference, a NaN
    per-image IoU (class absent from both masks) makes that column's mean NaN."""
    return {k: float(np.mean(table[:, i])) for i, k in enumerate(IOU_FIELDS)}
