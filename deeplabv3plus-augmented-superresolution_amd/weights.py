"""DeepLabV3+ (Xception-65, OS16) parameter inventory, synthetic initialisation and BN folding.

Names, shapes and BatchNorm epsilons follow the reference's Keras graph (model.py:149-306,
381-424, 463-541) so that a weight file exported from the reference model
(``{layer_name}/{variable}`` -> array, Keras layouts: Conv2D HWIO, DepthwiseConv2D [3,3,C,1])
loads by name, like ``load_weights(by_name=True)`` (model.py:145).  The pretrained bonlime .h5
(model.py:9) is a network download and is not available offline; ``make_synthetic_weights``
produces seeded random parameters of the same architecture instead.
"""
from __future__ import annotations

import numpy as np

XCEPTION_BN_EPS = 1e-3   # Keras default: every Xception-flow BN and aspp1-3 (model.py:214-221)
HEAD_BN_EPS = 1e-5       # image_pooling, aspp0, concat_projection, feature_projection0, decoder_conv*


def xception_blocks():
    """(prefix, cin, filters, skip_type, last_stride, rate, depth_activation) in graph order."""
    blocks = [
        ("entry_flow_block1", 64, [128, 128, 128], "conv", 2, 1, False),
        ("entry_flow_block2", 128, [256, 256, 256], "conv", 2, 1, False),
        ("entry_flow_block3", 256, [728, 728, 728], "conv", 2, 1, False),
    ]
    for i in range(16):
        blocks.append((f"middle_flow_unit_{i + 1}", 728, [728, 728, 728], "sum", 1, 1, False))
    blocks.append(("exit_flow_block1", 728, [728, 1024, 1024], "conv", 1, 1, False))
    blocks.append(("exit_flow_block2", 1024, [1536, 1536, 2048], None, 1, 2, True))
    return blocks


def make_divisible(value, divisor, min_value=None):
    """model.py:544-556."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(value + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * value:
        new_v += divisor
    return new_v


def mobilenet_blocks(alpha=1.0):
    """MobileNet_Backbone_Encoder (model.py:339-379): (block_id, in_channels, out_channels, stride, rate, skip)
    of the 16 inverted residual blocks (expansion factor 6, OS = 8: strides of blocks 6 and 13 replaced by dilation)."""
    spec = [(1, 24, 2, 1, False), (2, 24, 1, 1, True),
            (3, 32, 2, 1, False), (4, 32, 1, 1, True), (5, 32, 1, 1, True),
            (6, 64, 1, 1, False), (7, 64, 1, 2, True), (8, 64, 1, 2, True), (9, 64, 1, 2, True),
            (10, 96, 1, 2, False), (11, 96, 1, 2, True), (12, 96, 1, 2, True),
            (13, 160, 1, 2, False), (14, 160, 1, 4, True), (15, 160, 1, 4, True),
            (16, 320, 1, 4, False)]
    cin = make_divisible(int(16 * alpha), 8)
    out = []
    for bid, filters, stride, rate, skip in spec:
        cout = make_divisible(int(filters * alpha), 8)
        out.append((bid, cin, cout, stride, rate, skip))
        cin = cout
    return out


def mobilenet_inventory(classes=21, alpha=1.0):
    """Layers of the MobileNetV2 variant (model.py:308-337 EntryBlockMobile, :426-461 _inverted_res_block,
    :192-233 ASPP without the atrous branches, :296-306 logits), Keras names."""
    inv = []

    def conv(name, kh, cin, cout, bias=False):
        inv.append(("conv", name, dict(kh=kh, cin=cin, cout=cout, bias=bias)))

    def bn(name, c, eps):
        inv.append(("bn", name, dict(c=c, eps=eps)))

    first = make_divisible(32 * alpha, 8)
    conv("Conv", 3, 3, first)
    bn("Conv_BN", first, XCEPTION_BN_EPS)
    inv.append(("dw", "expanded_conv_depthwise", dict(c=first)))
    bn("expanded_conv_depthwise_BN", first, XCEPTION_BN_EPS)
    c0 = make_divisible(int(16 * alpha), 8)
    conv("expanded_conv_project", 1, first, c0)
    bn("expanded_conv_project_BN", c0, XCEPTION_BN_EPS)
    for bid, cin, cout, _stride, _rate, _skip in mobilenet_blocks(alpha):
        p = f"expanded_conv_{bid}_"
        conv(p + "expand", 1, cin, 6 * cin)
        bn(p + "expand_BN", 6 * cin, XCEPTION_BN_EPS)
        inv.append(("dw", p + "depthwise", dict(c=6 * cin)))
        bn(p + "depthwise_BN", 6 * cin, XCEPTION_BN_EPS)
        conv(p + "project", 1, 6 * cin, cout)
        bn(p + "project_BN", cout, XCEPTION_BN_EPS)
    c_enc = mobilenet_blocks(alpha)[-1][2]
    conv("image_pooling", 1, c_enc, 256)
    bn("image_pooling_BN", 256, HEAD_BN_EPS)
    conv("aspp0", 1, c_enc, 256)
    bn("aspp0_BN", 256, HEAD_BN_EPS)
    conv("concat_projection", 1, 512, 256)
    bn("concat_projection_BN", 256, HEAD_BN_EPS)
    conv("logits_semantic" if classes == 21 else "custom_logits_semantic", 1, 256, classes, bias=True)
    return inv


def layer_inventory(classes=21, backbone="xception", alpha=1.0, decoder="full", class_prediction=True):
    """Ordered list of (kind, name, shape-info dict).  kinds: conv, dw, bn.
    decoder: "full" (Decoder, model.py:235-259), "dcnn" (Decoder_only_DCNN, :261-280: feature_projection0 takes the
    2048-channel encoder output, decoder_conv0 48 channels) or "aspp" (Decoder_only_ASPP, :282-294: decoder_conv0 takes
    the 256-channel ASPP output); Xception only."""
    if decoder not in ("full", "dcnn", "aspp"):
        raise ValueError(f"decoder must be 'full', 'dcnn' or 'aspp', got {decoder!r}")
    if backbone == "mobilenet":
        return mobilenet_inventory(classes, alpha)
    inv = []

    def conv(name, kh, cin, cout, bias=False):
        inv.append(("conv", name, dict(kh=kh, cin=cin, cout=cout, bias=bias)))

    def bn(name, c, eps):
        inv.append(("bn", name, dict(c=c, eps=eps)))

    def sep(prefix, cin, cout, eps):
        inv.append(("dw", prefix + "_depthwise", dict(c=cin)))
        bn(prefix + "_depthwise_BN", cin, eps)
        conv(prefix + "_pointwise", 1, cin, cout)
        bn(prefix + "_pointwise_BN", cout, eps)

    conv("entry_flow_conv1_1", 3, 3, 32)
    bn("entry_flow_conv1_1_BN", 32, XCEPTION_BN_EPS)
    conv("entry_flow_conv1_2", 3, 32, 64)
    bn("entry_flow_conv1_2_BN", 64, XCEPTION_BN_EPS)
    for prefix, cin, filters, skip, _stride, _rate, _da in xception_blocks():
        c = cin
        for i, f in enumerate(filters):
            sep(f"{prefix}_separable_conv{i + 1}", c, f, XCEPTION_BN_EPS)
            c = f
        if skip == "conv":
            conv(prefix + "_shortcut", 1, cin, filters[-1])
            bn(prefix + "_shortcut_BN", filters[-1], XCEPTION_BN_EPS)
    conv("image_pooling", 1, 2048, 256)
    bn("image_pooling_BN", 256, HEAD_BN_EPS)
    conv("aspp0", 1, 2048, 256)
    bn("aspp0_BN", 256, HEAD_BN_EPS)
    for i in (1, 2, 3):
        sep(f"aspp{i}", 2048, 256, XCEPTION_BN_EPS)
    conv("concat_projection", 1, 1280, 256)
    bn("concat_projection_BN", 256, HEAD_BN_EPS)
    if decoder != "aspp":
        conv("feature_projection0", 1, 2048 if decoder == "dcnn" else 256, 48)
        bn("feature_projection0_BN", 48, HEAD_BN_EPS)
    sep("decoder_conv0", {"full": 304, "dcnn": 48, "aspp": 256}[decoder], 256, HEAD_BN_EPS)
    sep("decoder_conv1", 256, 256, HEAD_BN_EPS)
    if class_prediction:
        conv("logits_semantic" if classes == 21 else "custom_logits_semantic", 1, 256, classes, bias=True)
    return inv


def count_params(classes=21, backbone="xception", alpha=1.0, decoder="full"):
    n = 0
    for kind, _name, d in layer_inventory(classes, backbone, alpha, decoder):
        if kind == "conv":
            n += d["kh"] * d["kh"] * d["cin"] * d["cout"] + (d["cout"] if d["bias"] else 0)
        elif kind == "dw":
            n += 9 * d["c"]
        else:
            n += 4 * d["c"]
    return n


def make_synthetic_weights(seed=1234, classes=21, backbone="xception", alpha=1.0, decoder="full", class_prediction=True):
    """Seeded random parameters with variance-preserving scales so that activations stay O(1)
    through the 65+ layers (He-style std for kernels that follow a ReLU, 1/sqrt(fan_in)
    otherwise; BN statistics near identity; residual branches damped)."""
    rng = np.random.default_rng(seed)
    w = {}
    inv = layer_inventory(classes, backbone, alpha, decoder, class_prediction)
    relu_before = set()        # conv layers whose input passed through a ReLU
    if backbone == "mobilenet":
        relu_before |= {"expanded_conv_depthwise", "expanded_conv_project"}
        for bid, *_ in mobilenet_blocks(alpha):
            relu_before |= {f"expanded_conv_{bid}_depthwise", f"expanded_conv_{bid}_project"}
    for prefix, _cin, _f, _skip, _s, _r, depth_act in xception_blocks():
        for i in range(3):
            p = f"{prefix}_separable_conv{i + 1}"
            relu_before.add(p + "_depthwise" if not depth_act else p + "_pointwise")
    for p in ("aspp1", "aspp2", "aspp3", "decoder_conv0", "decoder_conv1"):
        relu_before.add(p + "_pointwise")
    relu_before |= {"entry_flow_conv1_2", "aspp0", "image_pooling", "concat_projection", "logits_semantic",
                    "custom_logits_semantic"}
    damped = {f"middle_flow_unit_{i + 1}_separable_conv3_pointwise_BN" for i in range(16)}
    damped |= {f"expanded_conv_{bid}_project_BN" for bid, _ci, _co, _s, _r, skip in mobilenet_blocks(alpha) if skip}
    for kind, name, d in inv:
        if kind == "conv":
            fan_in = d["kh"] * d["kh"] * d["cin"]
            gain = 2.0 if name in relu_before else 1.0
            w[name + "/kernel"] = (rng.standard_normal((d["kh"], d["kh"], d["cin"], d["cout"])) *
                                   np.sqrt(gain / fan_in)).astype(np.float32)
            if d["bias"]:
                b = (rng.standard_normal(d["cout"]) * 0.1).astype(np.float32)
                if d["cout"] > 8:
                    b[8] += 0.8       # make the default filter class (cat, id 8) win a sizeable region
                w[name + "/bias"] = b
        elif kind == "dw":
            gain = 2.0 if name in relu_before else 1.0
            w[name + "/depthwise_kernel"] = (rng.standard_normal((3, 3, d["c"], 1)) *
                                             np.sqrt(gain / 9.0)).astype(np.float32)
        else:
            c = d["c"]
            g = rng.uniform(0.8, 1.2, c)
            if name in damped:
                g *= 0.3
            w[name + "/gamma"] = g.astype(np.float32)
            w[name + "/beta"] = (rng.standard_normal(c) * 0.05).astype(np.float32)
            w[name + "/moving_mean"] = (rng.standard_normal(c) * 0.05).astype(np.float32)
            w[name + "/moving_variance"] = rng.uniform(0.8, 1.2, c).astype(np.float32)
    return w


def save_weights(path, weights):
    np.savez(path, **weights)


def keras_variable_name(dataset_path):
    """HDF5 dataset path of a Keras weight file -> "{layer}/{variable}", or None for what is not a layer variable.
    save_weights() layout: <layer>/<layer>/<variable>:0; a full-model file keeps the same tree under model_weights/ (and
    the optimizer state under optimizer_weights/, which is ignored)."""
    parts = dataset_path.strip("/").split("/")
    if parts[0] == "optimizer_weights":
        return None
    if parts[0] == "model_weights":
        parts = parts[1:]
    if len(parts) < 2:
        return None
    return f"{parts[0]}/{parts[-1].split(':')[0]}"


def load_weights(path):
    """Local file only: an .npz with ``{layer}/{variable}`` keys, or a Keras ``.h5`` weight file (the format of the
    bonlime checkpoint the reference downloads, model.py:129-145: groups ``<layer>/<layer>/<variable>:0``), read with
    hdf5_lite and matched by layer name like ``load_weights(by_name=True)``.  Never a URL: the reference's get_file()
    download (model.py:134-143) has no offline counterpart."""
    if str(path).startswith(("http://", "https://")):
        raise ValueError("weights must be a local file; network downloads are not supported")
    if str(path).endswith((".h5", ".hdf5")):
        from . import hdf5_lite
        datasets, _ = hdf5_lite.read(path)
        out = {}
        for name, value in datasets.items():
            key = keras_variable_name(name)
            if key is not None:
                out[key] = np.asarray(value)
        if not out:
            raise ValueError(f"{path}: no '<layer>/.../<variable>' datasets found")
        return out
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def merge_by_name(template, loaded):
    """Keras ``load_weights(by_name=True, skip_mismatch=True)`` (model.py:145) on ``{layer}/{variable}`` dictionaries.
    Matching is PER LAYER, like Keras: a layer takes its weights from the file only when the file holds every variable of
    it with the right shape; a layer with a missing or mismatched variable keeps ALL its initial values (Keras skips the
    whole layer with a warning) and each of its variables is reported.  Loaded entries the model has no layer for are
    ignored, as Keras does.  Returns (params, [(variable name, reason)])."""
    by_layer = {}
    for name in template:
        by_layer.setdefault(name.rsplit("/", 1)[0], []).append(name)
    out, skipped = {}, []
    for layer, names in by_layer.items():
        problems = {}
        for name in names:
            got = loaded.get(name)
            if got is None:
                problems[name] = "absent from the file"
            elif tuple(np.shape(got)) != tuple(template[name].shape):
                problems[name] = f"shape {tuple(np.shape(got))} != {tuple(template[name].shape)}"
        for name in names:
            if problems:
                out[name] = template[name]
                skipped.append((name, problems.get(name, f"layer {layer} skipped as a whole ({next(iter(problems.values()))})")))
            else:
                out[name] = np.asarray(loaded[name], dtype=np.float32)
    return out, skipped


def loaded_fraction(template, skipped):
    """Fraction of the model's convolution / depthwise kernels that came from the file (0 = the file matched nothing)."""
    kernels = [n for n in template if n.endswith("/kernel") or n.endswith("/depthwise_kernel")]
    missed = {n for n, _ in skipped}
    return (sum(1 for n in kernels if n not in missed) / len(kernels)) if kernels else 0.0


def bn_scale_shift(weights, name, eps):
    g = weights[name + "/gamma"].astype(np.float32)
    b = weights[name + "/beta"].astype(np.float32)
    m = weights[name + "/moving_mean"].astype(np.float32)
    v = weights[name + "/moving_variance"].astype(np.float32)
    scale = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    return scale, shift


def fold_conv_bn(weights, conv_name, bn_name, eps):
    """Conv2D (no bias) followed by inference BatchNorm -> ([K, cout] kernel, [cout] bias)."""
    k = weights[conv_name + "/kernel"].astype(np.float32)
    kh, kw, cin, cout = k.shape
    scale, shift = bn_scale_shift(weights, bn_name, eps)
    return (k * scale).reshape(kh * kw * cin, cout).astype(np.float32), shift


def fold_dw_bn(weights, dw_name, bn_name, eps):
    """DepthwiseConv2D followed by inference BatchNorm -> ([3,3,C] kernel, [C] bias)."""
    k = weights[dw_name + "/depthwise_kernel"].astype(np.float32)[:, :, :, 0]
    scale, shift = bn_scale_shift(weights, bn_name, eps)
    return np.ascontiguousarray(k * scale, dtype=np.float32), shift
