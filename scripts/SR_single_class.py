#!/usr/bin/env python3
"""Batch consumer: interchange files -> ASR / max-SR / mean-SR -> IoUs per image -> means.  Counterpart of
the reference's SR_single_class.py (same hyper-parameters and IoU record); images sharded over the GPUs of
the node, one all-gather of the per-image IoU records at the end (asr_amd.distributed)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IMG_SIZE = (512, 512)
FEATURE_SIZE = (128, 128)
HYPER = dict(lambda_df=1, lambda_tv=0.3, lambda_L2=0.7, lambda_L1=0.0, num_iter=300, optimizer="adam",
             learning_rate=1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="folder of interchange files written by generate_augmented_copies.py")
    ap.add_argument("--gt", required=True, help="folder of ground-truth label PNGs named <filename>.png")
    ap.add_argument("--standard", default=None, help="folder of standard-output PNGs (optional)")
    ap.add_argument("--num_aug", type=int, default=10)
    ap.add_argument("--num_samples", type=int, default=500)
    ap.add_argument("--class_id", type=int, default=8)
    ap.add_argument("--th_factor", type=float, default=0.65)
    ap.add_argument("--feature_size", type=int, default=FEATURE_SIZE[0],
                    help="side of the stored model outputs: 128 for the Xception copies, 64 for MobileNet (OS 8)")
    ap.add_argument("--out", default=os.path.join(ROOT, "data", "superres_root", "superres_output"))
    args = ap.parse_args()

    import torch
    from asr_amd import distributed as D
    from asr_amd.evaluation import evaluate_precomputed, interchange_files, mean_over_valid
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution

    rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    optimizer_obj = Optimizer(optimizer=HYPER["optimizer"], learning_rate=HYPER["learning_rate"], amsgrad=HYPER["amsgrad"],
                              lr_scheduler=HYPER["lr_scheduler"], decay_steps=HYPER["decay_steps"],
                              decay_rate=HYPER["decay_rate"])
    sr = Superresolution(lambda_df=HYPER["lambda_df"], lambda_tv=HYPER["lambda_tv"], lambda_L2=HYPER["lambda_L2"],
                         lambda_L1=HYPER["lambda_L1"], num_iter=HYPER["num_iter"], num_aug=args.num_aug,
                         optimizer=optimizer_obj, feature_size=(args.feature_size, args.feature_size))
    paths = interchange_files(args.data)[:args.num_samples]
    table, valid = evaluate_precomputed(sr, paths, args.gt, args.standard, num_aug=args.num_aug, class_id=args.class_id,
                                        th_factor=args.th_factor, img_size=IMG_SIZE, out_dir=args.out, rank=rank, world=world)
    if rank == 0:
        m = mean_over_valid(table, valid)
        print(f"Avg. Standard IoUs (No bg): {m['standard_single']},  Avg. Augmented SR IoUs (No bg): {m['aug_single']}")
        print(f"Avg. Standard IoUs (with bg): {m['standard_bg']},  Avg. Augmented SR IoUs (with bg): {m['aug_bg']}")
        print(f"Avg. Max SR IoUs: {m['max']}, Avg. Mean SR IoUs: {m['mean']}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
