#!/usr/bin/env python3
"""Standard (non-SR) segmentation output used as the comparison baseline -- counterpart of the reference's
generate_standard_output.py (:52-65): model with the final bilinear upsample, argmax, keep `class_id`, save a
label PNG per image.  Images come from --images (folder or list file) instead of the VOC lists."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IMG_SIZE = (512, 512)
BATCH_SIZE = 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", required=True)
    ap.add_argument("--num_samples", type=int, default=500)
    ap.add_argument("--backbone", type=str, choices=["mobilenet", "xception"], default="xception")
    ap.add_argument("--use_validation", action="store_true")
    ap.add_argument("--class_id", type=int, default=8, choices=range(21))
    ap.add_argument("--weights", default=None)
    ap.add_argument("--out_root", default=os.path.join(ROOT, "data", "superres_root", "standard_output"))
    args = ap.parse_args()

    import torch
    from PIL import Image
    from asr_amd import distributed as D
    from asr_amd.model import DeeplabV3Plus
    from asr_amd.utils import load_image, create_mask

    rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    if os.path.isdir(args.images):
        paths = sorted(os.path.join(args.images, f) for f in os.listdir(args.images) if f.lower().endswith((".jpg", ".jpeg")))
    else:
        paths = [line.strip() for line in open(args.images) if line.strip()]
    paths = paths[:args.num_samples]
    out_dir = os.path.join(args.out_root, f"{args.backbone}_{args.class_id}{'_validation' if args.use_validation else ''}")
    os.makedirs(out_dir, exist_ok=True)
    model = DeeplabV3Plus(input_shape=IMG_SIZE + (3,), classes=21, OS=16, last_activation=None, load_weights=True,
                          backbone=args.backbone, weights_path=args.weights).build_model(final_upsample=True)
    mine = D.shard_indices(len(paths), rank, world)
    for start in range(0, len(mine), BATCH_SIZE):
        idx = mine[start:start + BATCH_SIZE]
        batch = np.stack([load_image(paths[g], image_size=IMG_SIZE, normalize=True) for g in idx])
        masks = create_mask(model.predict_device(batch, batch_size=BATCH_SIZE))[..., 0]        # device, int64
        masks = torch.where(masks == args.class_id, masks, torch.zeros_like(masks)).to(torch.uint8).cpu().numpy()
        for g, m in zip(idx, masks):
            name = os.path.splitext(os.path.basename(paths[g]))[0]
            Image.fromarray(m).save(os.path.join(out_dir, f"{name}.png"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
