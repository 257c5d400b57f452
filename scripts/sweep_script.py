#!/usr/bin/env python3
"""One evaluation of a hyper-parameter setting over the precomputed augmented copies -- counterpart of the reference's
sweep_script.py (the function a wandb agent calls per trial, sweep_script.py:50-171) without wandb: the setting comes from
--config (JSON or YAML with the keys of `hyperparamters_default`, e.g. one sample of configs/sweep_configs/sweep_all.yaml)
and/or --set key=value, and the six averages wandb.log receives are printed as one JSON line.  Every optimiser
(adam / adagrad / adadelta / adamax / sgd), use_BTV and copy_dropout run inside the HIP solver."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED = 1234
np.random.seed(SEED)             # sweep_script.py:19 -- copy_dropout's mask comes from this stream

IMG_SIZE = (512, 512)
FEATURE_SIZE = (128, 128)
# sweep_script.py:53-74
HYPERPARAMETERS_DEFAULT = {
    "lambda_df": 1.0, "lambda_tv": 0.5, "lambda_L2": 0.5, "lambda_L1": 0.0, "num_iter": 300, "num_aug": 100,
    "num_samples": 500, "use_BTV": False, "copy_dropout": 0.0, "optimizer": "adam", "learning_rate": 1e-3,
    "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7, "amsgrad": False, "initial_accumulator_value": 0.1,
    "momentum": 0.6, "nesterov": False, "lr_scheduler": True, "decay_steps": 50, "decay_rate": 0.5,
}


def _parse_value(text):
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        return {"true": True, "false": False, "none": None}.get(text.lower(), text)


def load_config(path, overrides):
    cfg = dict(HYPERPARAMETERS_DEFAULT)
    if path:
        with open(path) as fh:
            if path.endswith((".yaml", ".yml")):
                import yaml
                loaded = yaml.safe_load(fh)
            else:
                loaded = json.load(fh)
        cfg.update({k: (v["value"] if isinstance(v, dict) and "value" in v else v) for k, v in loaded.items()})
    for item in overrides:
        k, _, v = item.partition("=")
        cfg[k] = _parse_value(v)
    unknown = set(cfg) - set(HYPERPARAMETERS_DEFAULT)
    if unknown:
        raise SystemExit(f"unknown hyper-parameters: {sorted(unknown)}")
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="folder of interchange files written by generate_augmented_copies.py")
    ap.add_argument("--gt", required=True, help="folder of ground-truth label PNGs named <filename>.png")
    ap.add_argument("--standard", default=None, help="folder of standard-output PNGs (optional)")
    ap.add_argument("--config", default=None, help="JSON / YAML file with hyper-parameters")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override one hyper-parameter")
    ap.add_argument("--class_id", type=int, default=8)
    ap.add_argument("--th_factor", type=float, default=0.65)
    ap.add_argument("--feature_size", type=int, default=FEATURE_SIZE[0],
                    help="side of the stored model outputs: 128 for the Xception copies, 64 for MobileNet (OS 8)")
    ap.add_argument("--out", default=os.path.join(ROOT, "data", "superres_root", "superres_output"))
    args = ap.parse_args()
    config = load_config(args.config, args.set)

    import torch
    from asr_amd import distributed as D
    from asr_amd.evaluation import evaluate_precomputed, sweep_metrics
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.superres_utils import list_precomputed_data_paths

    rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    optimizer_obj = Optimizer(optimizer=config["optimizer"], learning_rate=config["learning_rate"], epsilon=config["epsilon"],
                              beta_1=config["beta_1"], beta_2=config["beta_2"], amsgrad=config["amsgrad"],
                              initial_accumulator_value=config["initial_accumulator_value"], momentum=config["momentum"],
                              nesterov=config["nesterov"], lr_scheduler=config["lr_scheduler"],
                              decay_steps=config["decay_steps"], decay_rate=config["decay_rate"])
    sr = Superresolution(lambda_df=config["lambda_df"], lambda_tv=config["lambda_tv"], lambda_L2=config["lambda_L2"],
                         lambda_L1=config["lambda_L1"], num_iter=config["num_iter"], num_aug=config["num_aug"],
                         optimizer=optimizer_obj, use_BTV=config["use_BTV"], copy_dropout=config["copy_dropout"],
                         feature_size=(args.feature_size, args.feature_size))
    path_list = list_precomputed_data_paths(args.data, sort=True)
    paths = path_list if config["num_samples"] is None else path_list[:config["num_samples"]]
    table = evaluate_precomputed(sr, paths, args.gt, args.standard, num_aug=config["num_aug"], class_id=args.class_id,
                                 th_factor=args.th_factor, img_size=IMG_SIZE, out_dir=args.out, rank=rank, world=world)
    if rank == 0:
        print(json.dumps({**sweep_metrics(table), "config": config}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
