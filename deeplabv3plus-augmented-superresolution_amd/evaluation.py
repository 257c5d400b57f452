"""The per-image evaluation loop of the reference's SR_single_class.py:72-134: interchange file -> augmented / max /
mean SR -> six IoUs per image -> means over the VALID images.  Images are sharded over the ranks of the node
(asr_amd.distributed), the per-image records are all-gathered once at the end."""
from __future__ import annotations

import os

import numpy as np

from . import distributed as D
from .superresolution_scripts.superres_utils import DATA_EXTS, compute_SR, load_SR_data, probe_SR_data
from .utils import compute_IoU, load_image



def interchange_files(root_dir):
    """Every interchange file under ``root_dir`` in ONE order shared by all ranks: by the integer value of the file stem
    (VOC-style ids, the order SR_single_class.py:72 asks for with sort=True), names that are not integers after them in
    lexical order.  os.walk order is file-system dependent, so it is never used as the order."""
    found = []
    for folder, _dirs, files in os.walk(root_dir):
        found += [os.path.join(folder, f) for f in files if f.endswith(DATA_EXTS)]

    def key(path):
        stem = os.path.basename(path).split(".")[0]
        return (0, int(stem), path) if stem.isdigit() else (1, 0, path)

    # one file per image: a folder that holds both 7.hdf5 and 7.npz (two runs of stage 1 with different ASR_DATA_EXT)
    # must not evaluate image 7 twice -- the first extension of DATA_EXTS wins (.hdf5, the reference's format)
    best = {}
    for path in sorted(found, key=lambda q: (key(q)[:2], DATA_EXTS.index(next(e for e in DATA_EXTS if q.endswith(e))), q)):
        best.setdefault((os.path.dirname(path), os.path.basename(path).split(".")[0]), path)
    return sorted(best.values(), key=key)


def evaluate_precomputed(sr, paths, gt_dir, standard_dir=None, num_aug=100, class_id=8, th_factor=0.65,
                         img_size=(512, 512), out_dir=None, rank=0, world=1, save_final_output=False):
    """Returns (table, valid) on every rank: the [len(paths), 6] IoU table (distributed.IOU_FIELDS order) and the bool
    mask of the files that were evaluated.  The row of an invalid file is all-NaN and never enters a mean, like the
    reference's ``continue`` (SR_single_class.py:85-90); a VALID image whose IoUs are NaN (class absent from both masks)
    keeps its row, so the mean over the valid rows is NaN exactly when the reference's np.mean is.

    Validity is what ``load_SR_data`` would decide, read from each file's HEADERS (``probe_SR_data``: no mask is loaded, no
    rank opens another rank's files), so host memory stays at ONE image's maps whatever the shard size, like the
    reference's loop -- and it is all-gathered BEFORE any solve, together with the number of Adam solves each file will run
    (two for slice_max files: class map and max map).  Each valid file is then loaded once, when its turn comes.  ``sr.optimizer``'s global step
    counter is then set per image to what the reference's sequential loop would have reached: num_iter * (solves of the
    valid files before it); a skipped file runs no solve there, so it does not advance the counter here either, and
    sharding changes no update."""
    mine = D.shard_indices(len(paths), rank, world)
    flags = []
    for g in mine:
        ok, n_solves = probe_SR_data(paths[g], num_aug=num_aug)
        if not ok:
            print(f"File: {paths[g]} is invalid, skipping...")
        flags.append([1.0 if ok else 0.0, float(n_solves)])
    status = D.all_gather_rows(mine, flags, len(paths), 2)                    # [files, (valid, solves)] on every rank
    valid = np.nan_to_num(status[:, 0]) > 0.5
    solves = np.where(valid, np.nan_to_num(status[:, 1]), 0.0).astype(np.int64)
    before = np.concatenate([[0], np.cumsum(solves)[:-1]]) if len(paths) else np.zeros(0, np.int64)
    records = []
    for g in mine:
        if not valid[g]:
            records.append([np.nan] * len(D.IOU_FIELDS))
            continue
        class_masks, max_masks, angles, shifts, filename = load_SR_data(paths[g], num_aug=num_aug)
        sr.optimizer.optimizer.iterations = int(before[g]) * sr.num_iter
        true_mask = load_image(os.path.join(gt_dir, f"{filename}.png"), image_size=img_size, normalize=False, is_png=True,
                               resize_method="nearest")
        mm = max_masks if max_masks is not None else []
        out = {t: compute_SR(sr, class_masks, angles, shifts, filename, max_masks=mm, SR_type=t, class_id=class_id,
                             dest_folder=out_dir, th_factor=th_factor, save_final_output=save_final_output)
               for t in ("aug", "max", "mean")}
        std = [np.nan, np.nan]
        if standard_dir:
            sm = load_image(os.path.join(standard_dir, f"{filename}.png"), image_size=img_size, normalize=False, is_png=True,
                            resize_method="nearest")
            std = [compute_IoU(true_mask, sm, img_size=img_size, class_id=class_id),
                   compute_IoU(true_mask, sm, img_size=img_size, class_id=class_id, include_bg=True)]
        records.append(std + [compute_IoU(true_mask, out["aug"], img_size=img_size, class_id=class_id),
                              compute_IoU(true_mask, out["aug"], img_size=img_size, class_id=class_id, include_bg=True),
                              compute_IoU(true_mask, out["max"], img_size=img_size, class_id=class_id),
                              compute_IoU(true_mask, out["mean"], img_size=img_size, class_id=class_id)])
    return D.all_gather_iou(mine, records, len(paths)), valid


def valid_rows(table, valid=None):
    """Rows of the images that were evaluated: by the explicit mask evaluate_precomputed returns; without one, every row
    that is not all-NaN (a table from elsewhere)."""
    table = np.asarray(table, dtype=np.float64)
    if valid is not None:
        return table[np.asarray(valid, dtype=bool)]
    return table[~np.isnan(table).all(axis=1)]


def mean_over_valid(table, valid=None):
    """The six means SR_single_class.py:129-134 prints: np.mean over the images that were not skipped (NaN where a valid
    image has a NaN IoU, as there)."""
    return D.mean_ious(valid_rows(table, valid))
