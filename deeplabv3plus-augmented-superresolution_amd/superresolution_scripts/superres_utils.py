"""SR glue with the reference's surface (superresolution_scripts/superres_utils.py):
``min_max_normalization`` (:56-62), ``threshold_image`` (:118-139), ``load_SR_data`` (:154-210),
``compute_SR`` (:213-273) and ``check_hdf5_validity`` (:108-115), plus the thin host-side list helpers the reference's
scripts import from this module (``get_img_paths`` :9-29, ``class_in_image`` :32-38, ``filter_images_by_class`` :41-53,
``load_precomputed_images`` :65-78, ``get_precomputed_folders_path`` :81-90, ``list_precomputed_data_paths`` :93-105,
``normalize_coefficients`` :142-151): pure file plumbing, no device work, kept so that code written against the
reference's module still imports.  The interchange file is the reference's HDF5 file (same dataset and
attribute names, written / read by ``hdf5_lite``: files from the reference's h5py writer load here and vice versa);
an ``.npz`` with the same keys is accepted too (``ASR_DATA_EXT=.npz`` makes it the written format).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib, hdf5_lite, ops

DATA_EXT = os.environ.get("ASR_DATA_EXT", ".hdf5")          # what save_SR_data writes: ".hdf5" (reference) or ".npz"
DATA_EXTS = (".hdf5", ".h5", ".npz")                         # what the readers accept


# ---- list helpers (host only) ------------------------------------------------------------------------
def _stem_as_int(path):
    return int(os.path.basename(path).split(".")[0])


def get_img_paths(image_list_path, image_folder, is_png=False, sort=True):
    """One image id per line of ``image_list_path`` -> ``image_folder/<id>.jpg`` (``.png`` with is_png); ``sort`` orders
    the result by the integer value of the id (VOC ids are integers)."""
    suffix = ".png" if is_png else ".jpg"
    with open(image_list_path) as fh:
        ids = [line.rstrip() for line in fh]
    paths = [os.path.join(image_folder, i + suffix) for i in ids]
    return sorted(paths, key=_stem_as_int) if sort else paths


def class_in_image(image_path, class_id, image_size=(512, 512)):
    """Does the label map that belongs to ``image_path`` (same VOC tree: JPEGImages -> SegmentationClassAug, jpg -> png)
    contain ``class_id``?  The map is resized with nearest neighbour like every label map of the path."""
    from ..utils import load_image
    label_path = image_path.replace("JPEGImages", "SegmentationClassAug").replace("jpg", "png")
    labels = load_image(label_path, image_size=image_size, normalize=False, is_png=True, resize_method="nearest")
    return bool((np.asarray(labels) == class_id).any())


def filter_images_by_class(path_list, filter_class_id, num_images=None, image_size=(512, 512)):
    """The paths whose label map contains ``filter_class_id``, in list order, at most ``num_images`` of them."""
    limit = len(path_list) if num_images is None else num_images
    kept = []
    for path in path_list:
        if len(kept) == limit:
            break
        if class_in_image(path, class_id=filter_class_id, image_size=image_size):
            kept.append(path)
    return kept


def load_precomputed_images(img_folder):
    """The PNG copies ``0.png, 1.png, ...`` of a pre-HDF5 output folder, in numeric order (its ``.npy`` side files are
    not images)."""
    from ..utils import load_image
    numbers = sorted(int(name[:-len(".png")]) for name in os.listdir(img_folder) if ".npy" not in name)
    return [load_image(os.path.join(img_folder, f"{k}.png"), normalize=False, is_png=True) for k in numbers]


def get_precomputed_folders_path(root_dir, num_aug=100):
    """Sub-folders of ``root_dir`` that are complete: ``num_aug`` copies plus the two side files (angles, shifts).
    Incomplete ones are reported and left out."""
    complete = []
    for name in os.listdir(root_dir):
        folder = os.path.join(root_dir, name)
        if len(os.listdir(folder)) != num_aug + 2:
            print(f"Skipped folder named {name} as it is not valid")
            continue
        complete.append(folder)
    return complete


def list_precomputed_data_paths(root_dir, sort=False):
    """Every ``.hdf5`` interchange file below ``root_dir`` (os.walk order, or by integer stem with ``sort``).  The
    evaluation loop of this package uses ``asr_amd.evaluation.interchange_files`` instead: one order on every rank,
    ``.npz`` files too, one file per stem."""
    found = [os.path.join(folder, f) for folder, _dirs, files in os.walk(root_dir) for f in files if f.endswith(".hdf5")]
    return sorted(found, key=_stem_as_int) if sort else found


def normalize_coefficients(coeff_dict):
    """The same keys with the values scaled to sum to one."""
    total = np.sum(list(coeff_dict.values()))
    return {key: value / total for key, value in coeff_dict.items()}


def min_max_normalization(image, new_min=0.0, new_max=255.0, global_min=None, global_max=None):
    """Host arrays: numpy float32 arithmetic in the reference's order.  Device tensors stay on the device."""
    if isinstance(image, torch.Tensor) and image.is_cuda:
        mm = ops.minmax(image.contiguous())[0]
        mn = mm[0] if global_min is None else torch.as_tensor(global_min, dtype=torch.float32, device=image.device)
        mx = mm[1] if global_max is None else torch.as_tensor(global_max, dtype=torch.float32, device=image.device)
        den = mx - mn
        den = torch.where(den != 0, den, torch.ones_like(den))
        return new_min + ((image - mn) * (new_max - new_min)) / den
    image = np.asarray(image)
    mn = image.min() if global_min is None else global_min
    mx = image.max() if global_max is None else global_max
    num = (image - mn) * (new_max - new_min)
    den = (mx - mn) if (mx - mn) != 0 else 1.0
    return new_min + (num / den)


def threshold_image(image, th_value, th_factor=.15, th_mask=None):
    """{0, th_value} int32 mask: image >= th_mask, or image > th_factor * max(image).  Runs on the
    device (asr_threshold_f32); returns a host array for host input, a device tensor otherwise."""
    was_tensor = isinstance(image, torch.Tensor)
    dev = _lib.require_gpu()
    img = image if was_tensor else torch.as_tensor(np.asarray(image, dtype=np.float32))
    img = img.to(device=dev, dtype=torch.float32).contiguous()
    tm = None
    if th_mask is not None:
        tm = th_mask if isinstance(th_mask, torch.Tensor) else torch.as_tensor(np.asarray(th_mask, dtype=np.float32))
        tm = tm.to(device=dev, dtype=torch.float32).contiguous()
    out = ops.threshold(img, th_value, th_factor=th_factor, th_mask=tm)
    return out if was_tensor else out.cpu().numpy()


# ---- interchange file (augmentation_utils.py:117-136 writer, superres_utils.py:154-210 reader) -------
def save_SR_data(path_without_ext, class_masks, max_masks, angles, shifts, filename, mode, angle_max, shift_max, ext=None):
    ext = ext or DATA_EXT
    os.makedirs(os.path.dirname(path_without_ext) or ".", exist_ok=True)
    data = dict(class_masks=np.asarray(class_masks, dtype=np.float32), angles=np.asarray(angles, dtype=np.float32),
                shifts=np.asarray(shifts, dtype=np.float32))
    if max_masks is not None and len(max_masks):
        data["max_masks"] = np.asarray(max_masks, dtype=np.float32)
    attrs = dict(filename=str(filename), mode=str(mode), angle_max=angle_max, shift_max=shift_max)
    if ext == ".npz":
        np.savez(path_without_ext + ext, **data, **{k: np.array(v) for k, v in attrs.items()})
    elif ext in (".hdf5", ".h5"):
        hdf5_lite.write(path_without_ext + ext, data, attrs)
    else:
        raise ValueError(f"unknown interchange format {ext!r} (.hdf5 or .npz)")
    return path_without_ext + ext


def _open_SR_file(filepath):
    """-> dict with the datasets plus the attributes 'filename' and 'mode' (either container format)."""
    if str(filepath).endswith(".npz"):
        with np.load(filepath, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    datasets, attrs = hdf5_lite.read(filepath)
    out = dict(datasets)
    out.update(attrs.get("/", {}))
    return out


def check_validity(file, num_aug=100):
    """Every array dataset has at least num_aug entries (superres_utils.py:108-115)."""
    for key in ("class_masks", "max_masks", "angles", "shifts"):
        if key in file and file[key].shape[0] < num_aug:
            return False
    return True


check_hdf5_validity = check_validity          # the reference's name (superres_utils.py:108)


def probe_SR_data(filepath, num_aug=100):
    """What load_SR_data WOULD do with this file, from its headers alone (no mask is read): (valid, solves) -- valid exactly
    when load_SR_data would not raise (readable, every array dataset with at least num_aug entries, name and mode present),
    solves = the Adam solves an evaluation of it runs (2 for slice_max files: class map and max map, else 1).  Used by the
    sharded evaluation loop, which must know both for EVERY file before its first solve (evaluation.py) and must not hold a
    shard's worth of masks in host memory to find out."""
    try:
        if str(filepath).endswith(".npz"):
            file = _open_SR_file(filepath)                      # (the .npz container has no cheap header; stage-1 default is .hdf5)
            shapes = {k: np.shape(v) for k, v in file.items()}
            mode = str(file["mode"])
            str(file["filename"])
        else:
            shapes, attrs = hdf5_lite.header(filepath)
            root = attrs.get("/", {})
            mode = str(root["mode"])
            str(root["filename"])
        for key in ("class_masks", "angles", "shifts") + (("max_masks",) if mode == "slice_max" else ()):
            if key not in shapes:
                return False, 0
        for key in ("class_masks", "max_masks", "angles", "shifts"):
            if key in shapes and (len(shapes[key]) == 0 or shapes[key][0] < num_aug):
                return False, 0
        return True, 2 if mode == "slice_max" else 1
    except Exception:
        return False, 0


def load_SR_data(filepath, num_aug=100, global_normalize=True):
    """Returns (class_masks [N,h,w,1], max_masks | None, angles, shifts, filename) as host arrays;
    argmax / slice_max masks are min-max normalised to [0,1] (superres_utils.py:183-206)."""
    file = _open_SR_file(filepath)
    if not check_validity(file, num_aug=num_aug):
        raise Exception(f"File: {filepath} is invalid")
    filename = str(file["filename"])
    mode = str(file["mode"])
    angles = file["angles"][:num_aug]
    shifts = file["shifts"][:num_aug]
    class_masks = file["class_masks"][:num_aug].astype(np.float32)
    max_masks = file["max_masks"][:num_aug].astype(np.float32) if mode == "slice_max" else None

    def normalise(stack):
        if global_normalize:
            gmin, gmax = stack.min(), stack.max()
            return np.stack([min_max_normalization(im, 0.0, 1.0, gmin, gmax) for im in stack]).astype(np.float32)
        return np.stack([min_max_normalization(im, 0.0, 1.0) for im in stack]).astype(np.float32)

    if mode != "slice":
        class_masks = normalise(class_masks)
    if max_masks is not None:
        max_masks = normalise(max_masks)
    return class_masks, max_masks, angles, shifts, filename


def _save_png(path, image, scale=True):
    from PIL import Image
    a = np.asarray(image, dtype=np.float32)
    a = a[..., 0] if a.ndim == 3 else a
    if scale:                                   # tf.keras.utils.save_img(scale=True)
        a = a - a.min()
        mx = a.max()
        if mx != 0:
            a = a / mx
        a = a * 255.0
    Image.fromarray(a.astype(np.uint8)).save(path)


def compute_SR(superresolution_obj, class_masks, angles, shifts, filename, dest_folder,
               SR_type="aug", max_masks=[], save_intermediate_output=False, save_final_output=False, class_id=8,
               th_factor=0.15):
    """Dispatch aug / mean / max SR, then threshold to a {0, class_id} mask [H,W,1] (host int32)."""
    out_folder = os.path.join(dest_folder, f"{SR_type}_SR")
    if not os.path.exists(out_folder):
        os.makedirs(out_folder)
    if SR_type == "aug":
        SR_function = superresolution_obj.augmented_superresolution
    elif SR_type == "mean":
        SR_function = superresolution_obj.mean_superresolution
    elif SR_type == "max":
        SR_function = superresolution_obj.max_superresolution
    else:
        raise ValueError("SR_type must be either 'aug', 'mean' or 'max'")

    target_image_class, _ = SR_function(class_masks, angles, shifts)
    target_image_max = None
    # the max mask is super-resolved only when it was produced, i.e. in slice_max OPM
    if max_masks is not None and len(max_masks) == len(class_masks):
        target_image_max, _ = SR_function(max_masks, angles, shifts)
        th_mask = threshold_image(target_image_class, class_id, th_mask=target_image_max)
    else:
        th_mask = threshold_image(target_image_class, class_id, th_factor=th_factor)

    if save_intermediate_output:
        _save_png(f"{out_folder}/{filename}_class.png", target_image_class)
        if target_image_max is not None:
            _save_png(f"{out_folder}/{filename}_max.png", target_image_max)
    if save_final_output:
        _save_png(f"{out_folder}/{filename}_{SR_type}_SR.png", th_mask)
    return th_mask
