"""CPU tests of the host-side mirror: argument validation and error behaviour identical to the reference,
float32 parameter math, plan bookkeeping -- nothing here touches a GPU."""
import os

import numpy as np
import pytest


def test_deeplab_constructor_validation_matches_reference():
    """model.py:20-30, 66-68: same ValueErrors; out-of-scope options are refused loudly."""
    from asr_amd.model import DeeplabV3Plus
    with pytest.raises(ValueError, match="weights"):
        DeeplabV3Plus(weights="imagenet")
    with pytest.raises(ValueError, match="last_activation"):
        DeeplabV3Plus(last_activation="relu")
    with pytest.raises(ValueError, match="Backbone"):
        DeeplabV3Plus(backbone="resnet")
    assert DeeplabV3Plus(backbone="mobilenet", OS=16).OS == 8         # model.py:53-55: mobilenet forces OS = 8
    assert DeeplabV3Plus(OS=8).OS == 8
    with pytest.raises(ValueError, match="OS"):
        DeeplabV3Plus(OS=32)
    # model.py:57-62: an input tensor replaces input_shape (its static shape is what carries over)
    assert DeeplabV3Plus(input_tensor=np.zeros((1, 256, 384, 3), np.float32)).input_shape == (256, 384, 3)
    with pytest.raises(ValueError, match="input_tensor"):
        DeeplabV3Plus(input_tensor=np.zeros((3,), np.float32))
    m = DeeplabV3Plus(input_shape=(512, 512, 3), classes=21, OS=16, last_activation=None, load_weights=True,
                      backbone="xception")
    with pytest.raises(ValueError, match="only_DCNN_output"):
        m.build_model(only_DCNN_output=True, only_ASPP_output=True)
    # the modified decoders change the layer inventory (model.py:261-294): 2048 -> 48 projection / no skip concat
    from asr_amd import weights as W
    shapes = {d: {n: i for k, n, i in W.layer_inventory(decoder=d)} for d in ("full", "dcnn", "aspp")}
    assert shapes["full"]["feature_projection0"]["cin"] == 256 and shapes["full"]["decoder_conv0_depthwise"]["c"] == 304
    assert shapes["dcnn"]["feature_projection0"]["cin"] == 2048 and shapes["dcnn"]["decoder_conv0_depthwise"]["c"] == 48
    assert "feature_projection0" not in shapes["aspp"] and shapes["aspp"]["decoder_conv0_depthwise"]["c"] == 256
    assert "logits_semantic" not in {n for _k, n, _i in W.layer_inventory(class_prediction=False)}
    with pytest.raises(ValueError, match="decoder"):
        W.layer_inventory(decoder="unet")


def test_mobilenet_inventory_follows_the_reference_graph():
    """model.py:308-379, 426-461: 16 inverted residual blocks (x6 expansion), strides (2, 1, 2, 1, 1, then 1 with dilation
    2 / 4), 320 encoder channels, ASPP = image pooling + aspp0 only."""
    from asr_amd import weights as W
    blocks = W.mobilenet_blocks()
    assert [b[2] for b in blocks] == [24, 24, 32, 32, 32, 64, 64, 64, 64, 96, 96, 96, 160, 160, 160, 320]
    assert [b[3] for b in blocks] == [2, 1, 2] + [1] * 13 and [b[4] for b in blocks][-3:] == [4, 4, 4]
    assert [b[5] for b in blocks] == [False, True, False, True, True, False, True, True, True, False, True, True, False,
                                      True, True, False]
    inv = W.layer_inventory(21, backbone="mobilenet")
    names = [n for _k, n, _d in inv]
    assert names[:6] == ["Conv", "Conv_BN", "expanded_conv_depthwise", "expanded_conv_depthwise_BN", "expanded_conv_project",
                         "expanded_conv_project_BN"]
    assert "expanded_conv_16_project_BN" in names and "aspp1_depthwise" not in names and "decoder_conv0_depthwise" not in names
    convs = {n: d for k, n, d in inv if k == "conv"}
    assert (convs["expanded_conv_1_expand"]["cin"], convs["expanded_conv_1_expand"]["cout"]) == (16, 96)
    assert convs["concat_projection"]["cin"] == 512 and convs["aspp0"]["cin"] == 320
    assert W.make_divisible(32 * 0.35, 8) == 16 and W.make_divisible(int(24 * 0.35), 8) == 8        # model.py:544-556
    assert 2.0e6 < W.count_params(21, backbone="mobilenet") < 2.3e6


def test_optimizer_schedule_and_persistent_counter():
    """optimizer.py:37-52 + SURVEY 3.3: lr_t = 1e-3 * 0.3^(i/60), alpha_t with the GLOBAL step that keeps
    growing across solves."""
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd import transforms as T
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    a1 = opt.schedule_alphas(3)
    assert opt.optimizer.iterations == 3 and a1.dtype == np.float32
    for i in range(3):
        lr = np.float32(1e-3) * np.float32(0.3) ** (np.float32(i) / np.float32(60))
        exp = lr * np.sqrt(1 - np.float32(0.999) ** (i + 1)) / (1 - np.float32(0.9) ** (i + 1))
        assert abs(a1[i] - exp) < 1e-9
    a2 = opt.schedule_alphas(2)                      # second solve: schedule restarts, bias correction continues at t = 4
    assert opt.optimizer.iterations == 5
    assert abs(a2[0] - T.adam_alpha(1e-3, 0.9, 0.999, 4)) < 1e-12
    assert float(T.exponential_decay_lr(1e-3, 60, 0.3, 60)) == pytest.approx(3e-4, rel=1e-6)
    # the sweep-only optimisers (optimizer.py:21-35): per-step scalar and kernel configuration
    from asr_amd import _lib
    for name, kind in (("adadelta", _lib.OPT_ADADELTA), ("adagrad", _lib.OPT_ADAGRAD), ("adamax", _lib.OPT_ADAMAX),
                       ("sgd", _lib.OPT_SGD)):
        o = Optimizer(name, 1e-2, momentum=0.9, nesterov=True)
        assert o.optimizer.config(use_btv=True).optimizer == kind and o.optimizer.config(use_btv=True).prior == _lib.PRIOR_BTV
        a = o.schedule_alphas(2)
        exp = [1e-2 / (1 - 0.9), 1e-2 / (1 - 0.81)] if name == "adamax" else [1e-2, 1e-2]
        np.testing.assert_allclose(a, exp, rtol=1e-6)
    assert Optimizer("adagrad", initial_accumulator_value=0.25).optimizer.slot_init == {"v": 0.25}
    sgd = Optimizer("sgd", momentum=0.5, nesterov=True).optimizer.config()
    assert (sgd.flag, sgd.c0) == (1, 0.5)
    assert Optimizer("whatever").optimizer.amsgrad is False     # unknown names fall through to Adam, like the reference


def test_transform_vectors_and_inverse():
    from asr_amd import transforms as T
    r = T.rotation_transforms(np.array([0.0, 0.3], np.float32), 512, 512)
    assert r.shape == (2, 8) and r.dtype == np.float32
    np.testing.assert_array_equal(r[0], [1, 0, 0, 0, 1, 0, 0, 0])
    c, s = np.cos(np.float32(0.3)), np.sin(np.float32(0.3))
    np.testing.assert_allclose(r[1, :6], [c, -s, (511 - (c * 511 - s * 511)) / 2, s, c, (511 - (s * 511 + c * 511)) / 2],
                               rtol=1e-6)
    t = T.translation_transforms(np.array([[3.5, -2.0]], np.float32))
    np.testing.assert_array_equal(t[0], [1, 0, -3.5, 0, 1, 2.0, 0, 0])
    np.testing.assert_array_equal(T.inverse_transforms(t)[0], [1, 0, 3.5, 0, 1, -2.0, 0, 0])
    inv = T.inverse_transforms(r[1:2])
    np.testing.assert_allclose(inv, T.rotation_transforms(np.array([-0.3], np.float32), 512, 512), atol=5e-5)


def test_sr_object_surface_and_errors():
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    from asr_amd.superresolution_scripts.augmentation_utils import create_augmented_copies_chunked
    s = Superresolution(1, 0.3, 0.7, 0.0)
    assert (s.num_iter, s.num_aug, s.feature_size, s.output_size, s.optimizer) == (200, 100, (64, 64), (512, 512), None)
    from asr_amd import _lib
    b = Superresolution(1, 0, 0, 0, num_aug=10, use_BTV=True, copy_dropout=0.25)
    cfg = b._config()
    assert (cfg.prior, cfg.btv_shift, round(cfg.btv_alpha, 6)) == (_lib.PRIOR_BTV, 2, 0.6)     # bilateral_tv defaults
    np.random.seed(3)
    m = b._drop_mask(int(10 * 0.25))
    assert m.sum() == 8 and b._drop_mask(2) is m                # superresolution.py:47-50, frozen after the first draw
    with pytest.raises(Exception, match="multiple"):          # augmentation_utils.py:31-32, raised before any GPU work
        create_augmented_copies_chunked(np.zeros((8, 8, 3), np.float32), 150, 0.1, 3, chunk_size=100)


def test_interchange_file_roundtrip_and_validity(tmp_path):
    from asr_amd.superresolution_scripts import superres_utils as su
    masks = np.zeros((6, 4, 4, 1), np.float32)
    masks[:, 1:3, 1:3] = 8.0
    mx = np.random.default_rng(0).random((6, 4, 4, 1)).astype(np.float32)
    a, sh = np.arange(6, dtype=np.float32), np.ones((6, 2), np.float32)
    p = su.save_SR_data(str(tmp_path / "x" / "17"), masks, mx, a, sh, "17", "slice_max", 0.15, 80)
    from asr_amd.evaluation import interchange_files
    assert interchange_files(str(tmp_path)) == [p]
    cm, mm, a2, s2, name = su.load_SR_data(p, num_aug=5)
    assert name == "17" and cm.shape == (5, 4, 4, 1) and mm.shape == (5, 4, 4, 1)
    assert cm.max() == 1.0 and cm.min() == 0.0 and mm.max() <= 1.0          # min-max normalised (mode != slice)
    np.testing.assert_array_equal(a2, a[:5])
    with pytest.raises(Exception, match="invalid"):
        su.load_SR_data(p, num_aug=7)
    p2 = su.save_SR_data(str(tmp_path / "x" / "18"), masks, None, a, sh, "18", "slice", 0.15, 80)
    cm2, mm2, *_ = su.load_SR_data(p2, num_aug=6)
    assert mm2 is None and cm2.max() == 8.0                                  # slice mode is stored already normalised
    np.testing.assert_allclose(su.min_max_normalization(np.array([2.0, 4.0, 6.0]), 0.0, 1.0), [0, 0.5, 1])


def test_weight_inventory_and_folding():
    from asr_amd import weights as W
    inv = W.layer_inventory(21)
    names = [n for _k, n, _d in inv]
    assert len(names) == len(set(names))
    assert "entry_flow_block2_separable_conv2_pointwise" in names and "logits_semantic" in names
    assert [n for n in names if n.endswith("_shortcut")] == ["entry_flow_block1_shortcut", "entry_flow_block2_shortcut",
                                                             "entry_flow_block3_shortcut", "exit_flow_block1_shortcut"]
    assert W.layer_inventory(5)[-1][1] == "custom_logits_semantic"          # model.py:298-301
    eps = {n: d["eps"] for k, n, d in inv if k == "bn"}
    assert eps["aspp1_depthwise_BN"] == 1e-3 and eps["aspp0_BN"] == 1e-5 and eps["decoder_conv1_pointwise_BN"] == 1e-5
    w = W.make_synthetic_weights(7, 21)
    assert w["entry_flow_conv1_1/kernel"].shape == (3, 3, 3, 32) and w["logits_semantic/bias"].shape == (21,)
    k, b = W.fold_conv_bn(w, "aspp0", "aspp0_BN", 1e-5)
    assert k.shape == (2048, 256) and b.shape == (256,)
    with pytest.raises(ValueError):
        W.load_weights("https://example.com/w.h5")
    with pytest.raises(FileNotFoundError):                       # a local .h5 is opened (hdf5_lite), not refused
        W.load_weights("/tmp/definitely_missing_weights.h5")


def test_evaluation_order_validity_and_means(tmp_path):
    """SR_single_class.py:72-90,129-134: files in integer order on every rank, an invalid file is skipped -- it does not
    enter the means and does not advance the optimiser's persistent step counter."""
    from asr_amd import distributed as D, evaluation as E
    from asr_amd.superresolution_scripts import superres_utils as su
    masks = np.zeros((6, 4, 4, 1), np.float32)
    a, sh = np.arange(6, dtype=np.float32), np.ones((6, 2), np.float32)
    for name, n in (("10", 6), ("9", 6), ("100", 3), ("2007_000032", 6)):
        su.save_SR_data(str(tmp_path / "d" / name), masks[:n], None, a[:n], sh[:n], name, "argmax", 0.15, 80)
    (tmp_path / "d" / "11.hdf5").write_bytes(b"not an hdf5 file")
    paths = E.interchange_files(str(tmp_path))
    assert [os.path.basename(p) for p in paths] == ["9.hdf5", "10.hdf5", "11.hdf5", "100.hdf5", "2007_000032.hdf5"]

    def loads(p):
        try:
            su.load_SR_data(p, num_aug=6)
            return True
        except Exception:
            return False

    valid = [loads(p) for p in paths]
    assert valid == [True, True, False, False, True]          # 11: unreadable, 100: only 3 copies
    table = np.full((5, 6), np.nan)
    table[[0, 1, 4]] = [[0.5] * 6, [0.7] * 6, [0.9] * 6]
    assert E.valid_rows(table).shape == (3, 6) and E.valid_rows(table, valid).shape == (3, 6)
    m = E.mean_over_valid(table, valid)
    assert abs(m["aug_single"] - 0.7) < 1e-12 and not any(np.isnan(v) for v in m.values())
    assert np.isnan(D.mean_ious(table)["aug_single"])         # the plain mean over all rows would be NaN
    # a VALID image whose IoUs are all NaN (class absent from both masks) stays a row: the mean is NaN, as np.mean over
    # the reference's lists would be -- with the explicit mask; NaN inference alone would drop it
    table[1] = np.nan
    assert np.isnan(E.mean_over_valid(table, valid)["aug_single"]) and E.valid_rows(table).shape == (2, 6)
    # one file per image when a folder holds two container formats of the same stem
    su.save_SR_data(str(tmp_path / "d" / "9"), masks, None, a, sh, "9", "argmax", 0.15, 80, ext=".npz")
    assert [os.path.basename(p) for p in E.interchange_files(str(tmp_path))][:2] == ["9.hdf5", "10.hdf5"]
    # the reference-named helpers are importable from the reference's module path
    assert su.list_precomputed_data_paths(str(tmp_path), sort=False) and su.normalize_coefficients({"a": 1.0, "b": 3.0}) == {"a": 0.25, "b": 0.75}
    lst = tmp_path / "ids.txt"
    lst.write_text("12\n7\n")
    assert su.get_img_paths(str(lst), "/x") == ["/x/7.jpg", "/x/12.jpg"] and su.get_img_paths(str(lst), "/x", is_png=True, sort=False)[0] == "/x/12.png"


def test_bench_refuses_a_wrong_rank_count():
    """`bench.py --gpus N` must either start N ranks or fail: a launcher that set WORLD_SIZE to something else, or a box
    without enough devices, never yields a line that claims N GPUs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr and "{" not in r.stdout
    import torch
    if torch.cuda.device_count() == 0:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "no GPU is visible" in r.stderr and "{" not in r.stdout


def test_weights_merge_by_name_skips_mismatches():
    """model.py:145 load_weights(by_name=True, skip_mismatch=True): matching variables are taken, a missing layer or a
    shape mismatch (e.g. a 5-class logits layer against a 21-class checkpoint) keeps the initial value and is reported."""
    from asr_amd import weights as W
    tmpl = W.make_synthetic_weights(7, 5)
    ckpt = W.make_synthetic_weights(8, 21)
    del ckpt["aspp0_BN/gamma"]
    ckpt["not_a_layer/kernel"] = np.zeros((1, 1, 2, 2), np.float32)
    merged, skipped = W.merge_by_name(tmpl, ckpt)
    names = dict(skipped)
    assert set(merged) == set(tmpl)
    assert "absent" in names["aspp0_BN/gamma"] and np.array_equal(merged["aspp0_BN/gamma"], tmpl["aspp0_BN/gamma"])
    # per LAYER, like Keras: aspp0_BN lost one variable -> all four of its variables keep their initial values
    bn = {"aspp0_BN/gamma", "aspp0_BN/beta", "aspp0_BN/moving_mean", "aspp0_BN/moving_variance"}
    assert set(names) == bn | {"custom_logits_semantic/kernel", "custom_logits_semantic/bias"}
    assert all(np.array_equal(merged[k], tmpl[k]) for k in bn) and "as a whole" in names["aspp0_BN/beta"]
    assert np.array_equal(merged["entry_flow_conv1_1/kernel"], ckpt["entry_flow_conv1_1/kernel"])
    assert W.loaded_fraction(tmpl, skipped) > 0.98
    same, none = W.merge_by_name(ckpt, dict(ckpt))
    assert none == [] and all(np.array_equal(same[k], ckpt[k]) for k in ckpt if k != "not_a_layer/kernel")
    # a checkpoint of another backbone matches (almost) nothing by name
    mob = W.make_synthetic_weights(7, 21, backbone="mobilenet")
    _m, sk = W.merge_by_name(mob, ckpt)
    assert W.loaded_fraction(mob, sk) < 0.2


def test_bench_quotes_pmc_traffic_per_config_and_per_translation_unit(tmp_path, monkeypatch):
    """bench.rooflines: `traffic` comes from the PMC summary of THE SAME config (r*_pmc_traffic.json for configs[1],
    r*_pmc_traffic_cfg<N>.json for configs[N]; none -> null), and only while the kernel family's translation unit is unchanged
    since the summary was taken (round 3 pasted the configs[1] figure into configs[4]'s line and stamped all units with one hash)."""
    import json
    import bench
    prof_dir = tmp_path / "profiles"
    prof_dir.mkdir()
    entry = lambda b: {"hbm_bytes_per_launch": b, "launches_sampled": 4}
    (prof_dir / "r09_pmc_traffic.json").write_text(json.dumps({"pw_gemm_f16x3_pre_ring_kernel": entry(1000), "dw_stream_full_kernel<1>": entry(10)}))
    (prof_dir / "r09_pmc_traffic_cfg4.json").write_text(json.dumps({"pw_gemm_f16x3_pre_ring_kernel": entry(4000), "dw_stream_full_kernel<1>": entry(40)}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    prof = {"pw16": [10.0, 1e12, 8e8, 10], "dw": [5.0, 1e9, 6e8, 5]}
    units = {u: "h" for u in ("gemm", "dwconv", "sepconv", "layers", "sr")}
    fresh = {"units": units, "profiled_units": units, "profiled_at_commit": "abc", "current": {u: True for u in units}}
    assert bench.rooflines(dict(prof), fresh, 1)["roofline"]["traffic"] == 1000
    assert bench.rooflines(dict(prof), fresh, 4)["roofline"]["traffic"] == 4000
    r2 = bench.rooflines(dict(prof), fresh, 2)                                  # no summary for configs[2]
    assert r2["roofline"]["traffic"] is None and "no PMC summary for configs[2]" in r2["roofline"]["traffic_note"]
    stale = dict(fresh, current=dict(fresh["current"], gemm=False))           # gemm.hip changed since: its counters are not quoted ...
    rs = bench.rooflines(dict(prof), stale, 1)
    assert rs["roofline"]["traffic"] is None and "another build of gemm.hip" in rs["roofline"]["traffic_note"]
    assert rs["roofline_depthwise"]["traffic"] == 10                            # ... the depthwise unit's still are


def test_bench_unit_hashes_cover_sources_headers_and_flags():
    import bench
    h = bench.unit_hashes()
    assert set(h) == {"core", "warp", "sr", "reduce", "gemm", "dwconv", "layers", "sepconv"} and len(set(h.values())) == len(h)
    p = bench.profile_provenance()
    assert set(p["current"]) == set(h) and all(isinstance(v, bool) for v in p["current"].values())
