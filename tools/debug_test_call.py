import os, sys, tempfile, pathlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from asr_amd import weights as W
import asr_amd.superresolution_scripts.augmentation_utils as au
orig = au.feature_maps_on_device
def wrapped(image_dev, model, cid, mode, angles, shifts, batch_size=16, profile=None):
    print("image stats", float(image_dev.mean()), image_dev.shape, "bs", batch_size, "cid", cid, mode, angles[:3])
    cls, mx = orig(image_dev, model, cid, mode, angles, shifts, batch_size=batch_size, profile=profile)
    print("cls frac8", float((cls == 8).float().mean()))
    return cls, mx
au.feature_maps_on_device = wrapped
from oracle import augment as OA
_opm = OA.opm
def opm_w(pred, cid, mode):
    cm, mm = _opm(pred, cid, mode)
    print("oracle frac8", float((np.stack(cm) == 8).mean()), "pred mean", float(np.mean(pred)))
    return cm, mm
OA.opm = opm_w
_pd = au.output_processing
def op_w(preds, cid, mode):
    print("product pred mean", float(preds.mean()))
    return _pd(preds, cid, mode)
au.output_processing = op_w
syn = W.make_synthetic_weights(seed=1234, classes=21)
try:
    T.test_hot_path_config1_test_cat.__wrapped__ if hasattr(T.test_hot_path_config1_test_cat, "__wrapped__") else None
    T.test_hot_path_config1_test_cat(torch.device("cuda", 0), syn, os.path.join(ROOT, "tests", "golden"), pathlib.Path(tempfile.mkdtemp()))
    print("PASS")
except AssertionError as e:
    print("FAIL", str(e)[:300])
