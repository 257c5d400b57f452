#!/usr/bin/env python3
"""Batch producer: augmented copies -> DeepLabV3+ -> OPM -> one interchange file per image.  Counterpart of
the reference's generate_augmented_copies.py (same flags); images are listed with --images (a folder or
a text file of paths) instead of the VOC file lists; files are the reference's {image}.hdf5 (asr_amd/hdf5_lite.py).
With one process per GPU (torch.distributed.run) images are dealt round-robin over the ranks while every
rank replays the reference's sequential RNG stream, so each image gets the reference's draws."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

parser = argparse.ArgumentParser()
parser.add_argument("--images", required=True, help="folder of .jpg images or a text file with one path per line")
parser.add_argument("--num_aug", help="Number of augmented copies created for each image", type=int, default=100)
parser.add_argument("--num_samples", help="Number of samples taken from the list", type=int, default=500)
parser.add_argument("--mode", type=str, choices=["slice_max", "slice", "argmax"], default="argmax")
parser.add_argument("--angle_max", help="Max angle value (in radians) used for rotations", type=float, default=0.3)
parser.add_argument("--shift_max", help="Max shift value used for traslations", type=int, default=30)
parser.add_argument("--backbone", type=str, choices=["mobilenet", "xception"], default="xception")
parser.add_argument("--use_validation", action="store_true")
parser.add_argument("--class_id", type=int, default=8, choices=range(21), required=True)
parser.add_argument("--weights", default=None, help="local Keras .h5 checkpoint or .npz of Keras weights")
parser.add_argument("--out_root", default=os.path.join(ROOT, "data", "superres_root", "augmented_copies"))

SEED = 1234
IMG_SIZE = (512, 512)
BATCH_SIZE = 16


def list_images(spec, limit):
    if os.path.isdir(spec):
        paths = sorted(os.path.join(spec, f) for f in os.listdir(spec) if f.lower().endswith((".jpg", ".jpeg")))
    else:
        paths = [line.strip() for line in open(spec) if line.strip()]
    return paths[:limit]


def main():
    args = parser.parse_args()
    import torch
    from asr_amd import distributed as D
    from asr_amd.model import DeeplabV3Plus
    from asr_amd.utils import load_image
    from asr_amd.superresolution_scripts import augmentation_utils as au, superres_utils as su

    rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    out_dir = os.path.join(args.out_root, f"{args.backbone}_{args.mode}_{args.class_id}_{args.num_aug}"
                                          f"{'_validation' if args.use_validation else ''}")
    paths = list_images(args.images, args.num_samples)
    print(f"[rank {rank}/{world}] Valid images: {len(paths)}")
    model = DeeplabV3Plus(input_shape=IMG_SIZE + (3,), classes=21, OS=16, last_activation=None, load_weights=True,
                          backbone=args.backbone, weights_path=args.weights).build_model(final_upsample=False)
    params = D.replay_augmentation_stream(len(paths), args.num_aug, args.angle_max, args.shift_max, seed=SEED)
    for g in D.shard_indices(len(paths), rank, world):
        name = os.path.splitext(os.path.basename(paths[g]))[0]
        image = load_image(paths[g], image_size=IMG_SIZE, normalize=True)
        angles, shifts = params[g]
        cls, mx = au.feature_maps_on_device(au._image_to_device(image), model, args.class_id, args.mode, angles, shifts,
                                            batch_size=BATCH_SIZE)
        su.save_SR_data(os.path.join(out_dir, name), cls.cpu().numpy()[..., None],
                        mx.cpu().numpy()[..., None] if mx is not None else None, angles, shifts, name, args.mode,
                        args.angle_max, args.shift_max)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
