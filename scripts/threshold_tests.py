#!/usr/bin/env python3
"""Threshold study -- counterpart of the reference's threshold_tests.py (:48-171) without wandb: one augmented-SR solve
per image (hyper-parameters of `hyperparamters_default`, prior weights normalised to sum 1), then the IoU of the
thresholded result for th_factor = 0.10 ... 0.90, averaged over the images; prints the table and writes the CSV."""
import argparse
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED = 1234
np.random.seed(SEED)
IMG_SIZE = (512, 512)
# threshold_tests.py:49-71
HYPERPARAMETERS_DEFAULT = {
    "lambda_df": 1.0, "lambda_tv": 0.84, "lambda_L2": 0.047, "lambda_L1": 0.0065, "num_iter": 300, "num_aug": 100,
    "num_samples": 500, "copy_dropout": 0.2, "use_BTV": False, "optimizer": "adam", "learning_rate": 1e-1, "beta_1": 0.9,
    "beta_2": 0.999, "epsilon": 1e-7, "amsgrad": False, "initial_accumulator_value": 0.1, "nesterov": True,
    "momentum": 0.2, "lr_scheduler": True, "decay_steps": 100, "decay_rate": 0.65,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="folder of interchange files written by generate_augmented_copies.py")
    ap.add_argument("--gt", required=True)
    ap.add_argument("--standard", default=None)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--class_id", type=int, default=8)
    ap.add_argument("--feature_size", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(ROOT, "data", "threshold_test"))
    args = ap.parse_args()
    config = dict(HYPERPARAMETERS_DEFAULT)
    for item in args.set:
        k, _, v = item.partition("=")
        if k not in config:
            raise SystemExit(f"unknown hyper-parameter {k}")
        try:
            config[k] = json.loads(v)
        except json.JSONDecodeError:
            config[k] = {"true": True, "false": False}.get(v.lower(), v)

    from asr_amd.utils import load_image, compute_IoU
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.superres_utils import (list_precomputed_data_paths, load_SR_data,
                                                                normalize_coefficients, threshold_image)
    coeff_dict = normalize_coefficients({"lambda_tv": config["lambda_tv"], "lambda_L2": config["lambda_L2"],
                                         "lambda_L1": config["lambda_L1"]})
    print(coeff_dict)
    optimizer_obj = Optimizer(optimizer=config["optimizer"], learning_rate=config["learning_rate"], epsilon=config["epsilon"],
                              beta_1=config["beta_1"], beta_2=config["beta_2"], amsgrad=config["amsgrad"],
                              initial_accumulator_value=config["initial_accumulator_value"], momentum=config["momentum"],
                              nesterov=config["nesterov"], lr_scheduler=config["lr_scheduler"],
                              decay_steps=config["decay_steps"], decay_rate=config["decay_rate"])
    sr = Superresolution(lambda_df=config["lambda_df"], **coeff_dict, num_iter=config["num_iter"], num_aug=config["num_aug"],
                         optimizer=optimizer_obj, use_BTV=config["use_BTV"], copy_dropout=config["copy_dropout"],
                         feature_size=(args.feature_size, args.feature_size))
    path_list = list_precomputed_data_paths(args.data, sort=True)
    paths = path_list if config["num_samples"] is None else path_list[:config["num_samples"]]
    th_values = [round(float(v), 2) for v in np.arange(0.1, 0.95, step=0.05)]
    ious_th = [[] for _ in th_values]
    standard_ious = []
    for filepath in paths:
        try:
            class_masks, _, angles, shifts, filename = load_SR_data(filepath, num_aug=config["num_aug"], global_normalize=True)
        except Exception:
            print(f"File: {filepath} is invalid, skipping...")
            continue
        ground_truth = load_image(os.path.join(args.gt, f"{filename}.png"), image_size=IMG_SIZE, normalize=False, is_png=True,
                                  resize_method="nearest")
        if args.standard:
            sm = load_image(os.path.join(args.standard, f"{filename}.png"), image_size=IMG_SIZE, normalize=False, is_png=True,
                            resize_method="nearest")
            standard_ious.append(compute_IoU(ground_truth, sm, img_size=IMG_SIZE, class_id=args.class_id))
        target, _ = sr.augmented_superresolution(class_masks, angles, shifts)
        for k, value in enumerate(th_values):
            th_mask = threshold_image(target, args.class_id, th_factor=value)
            ious_th[k].append(compute_IoU(ground_truth, th_mask, img_size=IMG_SIZE, class_id=args.class_id))
    rows = [{"Th_Value": v, "IoU": float(np.mean(ious_th[k])) if ious_th[k] else float("nan")} for k, v in enumerate(th_values)]
    for r in rows:
        print(f"{r['Th_Value']:.2f}  {r['IoU']}")
    best = max(rows, key=lambda r: (-1 if np.isnan(r["IoU"]) else r["IoU"]))
    print(f"Best record: {best}")
    if standard_ious:
        print(f"Standard IoU: {np.mean(standard_ious)}")
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, f"th_{len(paths)}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Th_Value", "IoU"])
        for r in rows:
            w.writerow([r["Th_Value"], r["IoU"]])
    print("Done")


if __name__ == "__main__":
    main()
