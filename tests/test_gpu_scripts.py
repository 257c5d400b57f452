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
    assert d["exact_f32"]["value"] > 50 and d["exact_f32"]["ms_per_step"] > 0
    assert d["config"]["exact_f32_copies_per_s"] == d["exact_f32"]["value"]     # inside `config`: the driver's record keeps it
    assert d["config"]["collective"] == {"backend": "none", "world_size": 1, "ranks_joined": 1}
    # PMC traffic is quoted per config from ITS summary, and only while that kernel family's translation unit is unchanged
    for obj in (d["roofline"], d["configs"]["4"]["roofline"]):
        assert obj["traffic"] is None or obj["traffic"] >= 0.5 * obj["algorithmic_bytes_per_launch"], obj["traffic_note"]
    m = d["mean_ious"]                  # the SR stage solves a non-empty problem: class 8 is present and recovered
    assert m["aug_single"] > 0.3 and m["mean"] > 0.3
    assert "model-derived" in d["config"]["ground_truth"]
    # "IoU delta vs ref" half of the metric: the HIP path against the oracle sample of the cpu_baseline leg
    pr = d["parity"]
    assert pr["argmax_agreement"] >= 0.999 and pr["max_abs_output_diff"] <= 2e-4 * pr["output_scale"]
    assert pr["iou_delta_max"] is not None and pr["iou_delta_max"] <= 1e-3, pr
    assert min(pr["mask_agreement"].values()) >= 0.999
    # BASELINE configs[2] and configs[4] next to the headline, each with its own roofline / cpu_baseline / parity
    for cid, copies, img in (("2", 100, 512), ("4", 200, 1024)):
        c = d["configs"][cid]
        assert c["config"]["baseline_config"] == int(cid) and c["config"]["num_aug"] == copies and str(img) in c["metric"]
        assert c["value"] > 20 and abs(c["value"] - copies * 1000.0 / c["ms_per_step"]) < 1e-3 * c["value"]
        assert c["roofline"]["bound"] == "mfma" and 0.1 < c["roofline"]["frac"] < 1.0
        assert c["cpu_baseline"]["value"] > 0 and c["parity"]["iou_delta_max"] is not None and c["parity"]["iou_delta_max"] <= 1e-3
        assert c["parity"]["argmax_agreement"] >= 0.999
    assert d["configs"]["2"]["config"]["opm"].startswith("slice") and d["configs"]["4"]["config"]["sr"] == "256x256 -> 512x512"


def test_bench_strong_scaling_odd_image_count(dev, tmp_path):
    """`bench.py --images M` (BASELINE configs[3] in its strong-scaling form, SR_single_class.py:83-134 over a fixed image
    set): M images IN ALL, image g on rank g mod N -- ragged shards for an odd M on two ranks -- and the gathered table
    equals the single-rank table row for row."""
    import json
    import numpy as np
    bench = os.path.join(ROOT, "bench.py")
    t2, t1 = str(tmp_path / "s2.npy"), str(tmp_path / "s1.npy")
    common = ["--no-cpu-baseline", "--no-roofline", "--no-f32-line", "--no-extra-configs", "--warmup", "1", "--images", "5"]
    out2 = _run([bench, "--gpus", "2", "--dump-table", t2] + common, ROOT)
    d2 = json.loads([l for l in out2.strip().splitlines() if l.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["steps"] == 3 and d2["config"]["images_total"] == 5
    assert d2["config"]["workload"].startswith("BASELINE configs[3]: 5 images sharded over 2 GPU(s)")
    assert d2["config"]["collective"] == {"backend": "gloo", "world_size": 2, "ranks_joined": 2}
    out1 = _run([bench, "--gpus", "1", "--dump-table", t1] + common, ROOT)
    d1 = json.loads([l for l in out1.strip().splitlines() if l.startswith("{")][0])
    assert d1["steps"] == 5 and d1["scaling"] == "strong"
    assert abs(d1["value"] - 5 * 100 * 1000.0 / (d1["ms_per_step"] * 5)) < 1e-3 * d1["value"]
    a, b = np.load(t2), np.load(t1)
    assert a.shape == b.shape == (5, 6) and not np.isnan(a[:, 2:]).any()
    np.testing.assert_array_equal(a, b)
    assert d2["mean_ious"] == d1["mean_ious"]


def test_bench_two_rank_rehearsal_equals_single_rank(dev, tmp_path):
    """`bench.py --gpus 2` starts its own two ranks (gloo rehearsal on the one visible GPU): image g goes to rank g mod 2,
    one all-gather brings the full [images, 6] IoU table to rank 0, and every per-image record equals the single-rank
    run of the same global images (sharding must not change any result: SR_single_class.py:66-70,83,122-134)."""
    import json
    import numpy as np
    bench = os.path.join(ROOT, "bench.py")
    t2, t1 = str(tmp_path / "t2.npy"), str(tmp_path / "t1.npy")
    common = ["--no-cpu-baseline", "--no-roofline", "--no-f32-line", "--no-extra-configs"]
    out2 = _run([bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dump-table", t2] + common, ROOT)
    lines = [l for l in out2.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["steps"] == 2 and "rehearsal" in d2
    _run([bench, "--gpus", "1", "--steps", "4", "--warmup", "2", "--dump-table", t1] + common, ROOT)
    a, b = np.load(t2), np.load(t1)
    assert a.shape == (6, 6) and not np.isnan(a[:, 2:]).any()
    np.testing.assert_array_equal(a, b)
    assert (a[:, 2] > 0.3).all()          # non-trivial masks (class 8 present), not 0/0
