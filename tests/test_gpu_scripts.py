"""The two-stage command-line flow (generate_augmented_copies -> SR_single_class) and the single-image
demo (test_SR), run as child processes on the GPU like a user of the reference would."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    r = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_two_stage_scripts(dev, golden_dir, tmp_path):
    imgs = tmp_path / "JPEGImages"
    gts = tmp_path / "gt"
    imgs.mkdir()
    gts.mkdir()
    shutil.copy(os.path.join(golden_dir, "test_cat.jpg"), imgs / "7.jpg")          # numeric names like VOC ids
    shutil.copy(os.path.join(golden_dir, "test_cat_gt.png"), gts / "7.png")
    out_root = tmp_path / "copies"
    _run([os.path.join(ROOT, "scripts", "generate_augmented_copies.py"), "--images", str(imgs), "--num_aug", "4",
          "--mode", "argmax", "--angle_max", "0.15", "--shift_max", "80", "--class_id", "8", "--out_root", str(out_root)],
         str(tmp_path))
    data_dir = out_root / "xception_argmax_8_4"
    assert (data_dir / "7.hdf5").exists()
    std_root = tmp_path / "standard"
    _run([os.path.join(ROOT, "scripts", "generate_standard_output.py"), "--images", str(imgs), "--class_id", "8",
          "--out_root", str(std_root)], str(tmp_path))
    std_dir = std_root / "xception_8"
    assert (std_dir / "7.png").exists()
    out = _run([os.path.join(ROOT, "scripts", "SR_single_class.py"), "--data", str(data_dir), "--gt", str(gts),
                "--standard", str(std_dir), "--num_aug", "4", "--class_id", "8", "--out", str(tmp_path / "sr_out")],
               str(tmp_path))
    assert "Avg. Max SR IoUs" in out and "Avg. Augmented SR IoUs" in out and "Avg. Standard IoUs (No bg): nan" not in out
    # one sweep trial (sweep_script.py) with the sweep-only options: Adagrad + bilateral TV + copy dropout
    import json
    out = _run([os.path.join(ROOT, "scripts", "sweep_script.py"), "--data", str(data_dir), "--gt", str(gts),
                "--standard", str(std_dir), "--out", str(tmp_path / "sweep_out"), "--set", "optimizer=adagrad",
                "--set", "learning_rate=0.01", "--set", "use_BTV=true", "--set", "copy_dropout=0.25", "--set", "num_aug=4",
                "--set", "num_iter=12"], str(tmp_path))
    out_th = _run([os.path.join(ROOT, "scripts", "threshold_tests.py"), "--data", str(data_dir), "--gt", str(gts),
                   "--standard", str(std_dir), "--out", str(tmp_path / "th_out"), "--set", "num_aug=4", "--set", "num_iter=8",
                   "--set", "learning_rate=0.001", "--set", "copy_dropout=0.25"], str(tmp_path))
    assert "Best record:" in out_th and out_th.count("\n0.") >= 17 and (tmp_path / "th_out" / "th_1.csv").exists()
    rec = json.loads(out.strip().splitlines()[-1])
    assert set(rec) == {"aug_iou_single", "aug_iou_multiple", "standard_iou_single", "standard_iou_multiple", "mean_iou",
                        "max_iou", "config"}
    assert rec["config"]["optimizer"] == "adagrad" and 0.0 <= rec["aug_iou_single"] <= 1.0


def test_single_image_demo(dev, tmp_path):
    out = _run([os.path.join(ROOT, "scripts", "test_SR.py"), "--num-aug", "4", "--num-iter", "20", "--out",
                str(tmp_path / "SR_output")], str(tmp_path))
    assert "Aug. SR (argmax OPM) IoU" in out
    for t in ("aug", "max", "mean"):
        assert (tmp_path / "SR_output" / f"{t}_SR" / f"test_cat_{t}_SR.png").exists()


def test_robustness_grid_script(dev, golden_dir, tmp_path):
    """check_robustness.py on a 2 x 2 x 2 grid with the MobileNet backbone (the small model keeps the test short)."""
    imgs = tmp_path / "JPEGImages"
    gts = tmp_path / "gt"
    imgs.mkdir()
    gts.mkdir()
    shutil.copy(os.path.join(golden_dir, "test_cat.jpg"), imgs / "7.jpg")
    shutil.copy(os.path.join(golden_dir, "test_cat_gt.png"), gts / "7.png")
    out = _run([os.path.join(ROOT, "scripts", "check_robustness.py"), "--images", str(imgs), "--gt", str(gts), "--backbone",
                "mobilenet", "--image_size", "128", "--angles", "0.0", "0.3", "--shifts", "0", "20", "--out",
                str(tmp_path / "rob")], str(tmp_path))
    assert out.count("mIoU:") == 8 and "Done:" in out
    import csv
    rows = list(csv.reader(open(tmp_path / "rob" / "robustness_1_class_all_small.csv")))
    assert rows[0] == ["Angle", "Shift_X", "Shift_Y", "mIoU"] and len(rows) == 9
    assert all(0.0 <= float(r[3]) <= 1.0 for r in rows[1:])


def test_bench_contract_line(dev):
    """bench.py prints ONE JSON line with the driver's keys, the roofline of the dominant kernel and the CPU baseline
    (a short run: 3 steps, bounded oracle sample)."""
    import json
    out = _run([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2"], ROOT)
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 3 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 50 and abs(d["value"] - 100 * 1000.0 / d["ms_per_step"]) < 1e-3 * d["value"]   # 100 copies per step
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert d["roofline_depthwise"]["bound"] == "hbm" and 0.3 < d["roofline_depthwise"]["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"]
