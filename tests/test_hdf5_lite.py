"""hdf5_lite (pure-Python HDF5 subset) against files written by the REAL h5py with the reference writer's calls
(tests/golden/*.hdf5, *.h5 + make_hdf5_golden.py), and -- where this image's /opt/conda python with
h5py is present -- files written here opened by the real h5py the way the reference's load_SR_data does."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from asr_amd import hdf5_lite, weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONDA_PY = "/opt/conda/bin/python3.9"


@pytest.mark.parametrize("mode", ["argmax", "slice_max"])
def test_reads_files_written_by_h5py(mode):
    ds, attrs = hdf5_lite.read(os.path.join(GOLDEN, f"sr_data_{mode}.hdf5"))
    exp = np.load(os.path.join(GOLDEN, f"sr_data_{mode}_expected.npz"))
    assert sorted(ds) == sorted(k for k in exp.files if exp[k].size)
    for k in ds:
        assert ds[k].dtype == np.float32 and np.array_equal(ds[k], exp[k])
    root = attrs["/"]
    assert root["filename"] == "2007_000033" and root["mode"] == mode and isinstance(root["mode"], str)
    assert root["angle_max"] == 0.15 and root["shift_max"] == 80


def test_load_SR_data_reads_the_reference_format():
    """load_SR_data (superres_utils.py:154-210) on an h5py-written file: slicing to num_aug, global min-max
    normalisation of argmax masks, validity check."""
    from asr_amd.superresolution_scripts import superres_utils as su
    path = os.path.join(GOLDEN, "sr_data_argmax.hdf5")
    exp = np.load(os.path.join(GOLDEN, "sr_data_argmax_expected.npz"))
    cm, mm, angles, shifts, name = su.load_SR_data(path, num_aug=4)
    raw = exp["class_masks"][:4]
    np.testing.assert_allclose(cm, (raw - raw.min()) / (raw.max() - raw.min()), rtol=0, atol=1e-7)
    assert mm is None and name == "2007_000033"
    assert np.array_equal(angles, exp["angles"][:4]) and np.array_equal(shifts, exp["shifts"][:4])
    with pytest.raises(Exception, match="invalid"):
        su.load_SR_data(path, num_aug=6)                       # the file holds 5 copies
    cm2, mm2, *_ = su.load_SR_data(os.path.join(GOLDEN, "sr_data_slice_max.hdf5"), num_aug=4)
    assert mm2 is not None and mm2.shape == cm2.shape and float(mm2.max()) == 1.0 and float(mm2.min()) == 0.0


def test_write_read_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    data = dict(class_masks=rng.random((3, 4, 5, 1), dtype=np.float32), angles=rng.random(3).astype(np.float32),
                shifts=rng.random((3, 2)), counts=np.arange(-3, 4, dtype=np.int32), empty=np.zeros((0, 2), np.float32))
    attrs = dict(filename="2008_000123", mode="slice", angle_max=0.5, shift_max=30, note="café")
    p = hdf5_lite.write(str(tmp_path / "x.hdf5"), data, attrs)
    ds, at = hdf5_lite.read(p)
    assert sorted(ds) == sorted(data)
    for k in data:
        assert ds[k].dtype == np.asarray(data[k]).dtype and ds[k].shape == np.asarray(data[k]).shape
        assert np.array_equal(ds[k], data[k])
    assert at["/"] == attrs
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.write(str(tmp_path / "y.hdf5"), {"a/b": np.zeros(2)})
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.write(str(tmp_path / "y.hdf5"), {"s": np.array(["x"])})


def test_save_and_load_SR_data_both_containers(tmp_path):
    from asr_amd.superresolution_scripts import superres_utils as su
    rng = np.random.default_rng(5)
    cm = [rng.random((4, 4, 1), dtype=np.float32) for _ in range(3)]
    mx = [rng.random((4, 4, 1), dtype=np.float32) for _ in range(3)]
    ang, sh = rng.random(3).astype(np.float32), rng.random((3, 2)).astype(np.float32)
    got = {}
    for ext in (".hdf5", ".npz"):
        p = su.save_SR_data(str(tmp_path / "7"), cm, mx, ang, sh, "7", "slice_max", 0.15, 80, ext=ext)
        assert p.endswith(ext)
        got[ext] = su.load_SR_data(p, num_aug=3)
    for a, b in zip(got[".hdf5"], got[".npz"]):
        assert (a == b) if isinstance(a, str) else np.array_equal(a, b)
    assert sorted(os.path.basename(p) for p in __import__("asr_amd.evaluation", fromlist=["x"]).interchange_files(str(tmp_path))) == ["7.hdf5"]          # one file per image: .hdf5 wins


def test_keras_weight_file_layout():
    """<layer>/<layer>/<variable>:0 groups of Model.save_weights(.h5) -> '<layer>/<variable>' keys (by-name loading)."""
    w = W.load_weights(os.path.join(GOLDEN, "keras_like_weights.h5"))
    exp = np.load(os.path.join(GOLDEN, "keras_like_weights_expected.npz"))
    assert sorted(w) == sorted(exp.files) and len(w) == 21
    for k in w:
        assert np.array_equal(w[k], exp[k])
    _, attrs = hdf5_lite.read(os.path.join(GOLDEN, "keras_like_weights.h5"))
    assert list(attrs["logits_semantic"]["weight_names"]) == ["logits_semantic/kernel:0", "logits_semantic/bias:0"]
    with pytest.raises(ValueError, match="local file"):
        W.load_weights("https://example.com/w.h5")


def test_unsupported_layouts_are_refused():
    with pytest.raises(hdf5_lite.Hdf5Error, match="not supported"):
        hdf5_lite.read(os.path.join(GOLDEN, "chunked_gzip.hdf5"))
    with pytest.raises(hdf5_lite.Hdf5Error, match="signature"):
        hdf5_lite.read(os.path.join(GOLDEN, "sr_data_argmax_expected.npz"))


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no h5py-capable interpreter in this image")
def test_files_written_here_open_in_real_h5py(tmp_path):
    """The reference's reader calls (superres_utils.py:170-181) executed by the real h5py on a file written here."""
    probe = subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("h5py not importable there")
    rng = np.random.default_rng(9)
    data = dict(class_masks=rng.random((3, 4, 5, 1), dtype=np.float32), max_masks=rng.random((3, 4, 5, 1), dtype=np.float32),
                angles=rng.random(3).astype(np.float32), shifts=rng.random((3, 2)).astype(np.float32))
    p = hdf5_lite.write(str(tmp_path / "w.hdf5"), data, dict(filename="2007_000033", mode="slice_max", angle_max=0.15, shift_max=80))
    code = ("import h5py, json, sys\n"
            "f = h5py.File(sys.argv[1], 'r')\n"
            "out = dict(keys=sorted(f.keys()), filename=f.attrs['filename'], mode=f.attrs['mode'],\n"
            "           mode_is_str=isinstance(f.attrs['mode'], str), not_slice=bool(f.attrs['mode'] != 'slice'),\n"
            "           angle_max=float(f.attrs['angle_max']), shift_max=int(f.attrs['shift_max']),\n"
            "           shapes={k: list(f[k].shape) for k in f.keys()}, dtypes={k: str(f[k].dtype) for k in f.keys()},\n"
            "           sums={k: float(f[k][:2].astype('float64').sum()) for k in f.keys()})\n"
            "print(json.dumps(out))\n")
    r = subprocess.run([CONDA_PY, "-c", code, p], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["keys"] == sorted(data) and out["filename"] == "2007_000033" and out["mode"] == "slice_max"
    assert out["mode_is_str"] and out["not_slice"] and out["angle_max"] == 0.15 and out["shift_max"] == 80
    for k, v in data.items():
        assert out["shapes"][k] == list(v.shape) and out["dtypes"][k] == "float32"
        assert abs(out["sums"][k] - float(v[:2].astype(np.float64).sum())) < 1e-12
