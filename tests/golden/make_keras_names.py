#!/usr/bin/env python3
"""Keras variable names and shapes of the reference's DeepLabV3+ graph, derived from its NAMING RULES alone.

    python tests/golden/make_keras_names.py            # writes tests/golden/keras_variables.json

What a user brings to this framework is the checkpoint the reference downloads (model.py:9, loaded with
``load_weights(by_name=True, skip_mismatch=True)``, model.py:145): a name that the loader does not expect is skipped
silently, layer by layer.  This script walks the reference's graph assembly the way model.py does -- one function per
block, channel counts PROPAGATED from layer to layer instead of written down -- and records every trainable / BatchNorm
variable a Keras layer of that name and type owns:

    Conv2D              <name>/kernel [kh, kw, cin, cout] (+ <name>/bias [cout] unless use_bias=False)
    DepthwiseConv2D     <name>/depthwise_kernel [3, 3, cin, 1]
    BatchNormalization  <name>/gamma, beta, moving_mean, moving_variance [c]

It is written independently of asr_amd/weights.py (which lists the same layers with their input widths spelled out);
tests/test_weight_names.py compares the two.  When /root/reference is present (this container, not the GPU box) the script
also checks itself against the text of model.py: every layer-name literal there must have been used here.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_MODEL = "/root/reference/model.py"


class Trace:
    """Records variables; a "tensor" is just its channel count."""

    def __init__(self):
        self.vars = {}
        self.literals = set()        # name fragments used, for the cross-check against the reference's text

    def _use(self, *fragments):
        self.literals.update(fragments)

    def conv(self, c, filters, k, name, use_bias=False):
        self.vars[name + "/kernel"] = [k, k, c, filters]
        if use_bias:
            self.vars[name + "/bias"] = [filters]
        return filters

    def depthwise(self, c, name):
        self.vars[name + "/depthwise_kernel"] = [3, 3, c, 1]
        return c

    def bn(self, c, name):
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            self.vars[f"{name}/{v}"] = [c]
        return c


def sepconv_bn(t, c, filters, prefix):
    """_SepConv_BN, model.py:463-508: depthwise -> BN -> pointwise (use_bias=False) -> BN."""
    t._use("_depthwise", "_depthwise_BN", "_pointwise", "_pointwise_BN")
    c = t.depthwise(c, prefix + "_depthwise")
    c = t.bn(c, prefix + "_depthwise_BN")
    c = t.conv(c, filters, 1, prefix + "_pointwise")
    return t.bn(c, prefix + "_pointwise_BN")


def xception_block(t, c, filter_list, prefix, skip):
    """_Xception_block, model.py:381-424: three separable convs; "conv" skip = _conv2d_same 1x1 (model.py:510-541: the layer is
    named by the prefix itself) + BN."""
    t._use("_separable_conv", "_shortcut", "_shortcut_BN")
    x = c
    for i in range(3):
        x = sepconv_bn(t, x, filter_list[i], f"{prefix}_separable_conv{i + 1}")
    if skip == "conv":
        s = t.conv(c, filter_list[-1], 1, prefix + "_shortcut")
        t.bn(s, prefix + "_shortcut_BN")
    return x


def xception_encoder(t):
    """EntryFlowBlock model.py:149-170, MiddleFlowBlocks :172-179, ExitFlowBlock :181-190.  Returns (features, skip)."""
    t._use("entry_flow_conv1_1", "entry_flow_conv1_1_BN", "entry_flow_conv1_2", "entry_flow_conv1_2_BN", "entry_flow_block1",
           "entry_flow_block2", "entry_flow_block3", "middle_flow_unit_", "exit_flow_block1", "exit_flow_block2")
    x = t.conv(3, 32, 3, "entry_flow_conv1_1")
    x = t.bn(x, "entry_flow_conv1_1_BN")
    x = t.conv(x, 64, 3, "entry_flow_conv1_2")
    x = t.bn(x, "entry_flow_conv1_2_BN")
    x = xception_block(t, x, [128, 128, 128], "entry_flow_block1", "conv")
    x = skip = xception_block(t, x, [256, 256, 256], "entry_flow_block2", "conv")   # skip: block2's second sepconv, 256 wide too
    x = xception_block(t, x, [728, 728, 728], "entry_flow_block3", "conv")
    for i in range(16):
        x = xception_block(t, x, [728, 728, 728], f"middle_flow_unit_{i + 1}", "sum")
    x = xception_block(t, x, [728, 1024, 1024], "exit_flow_block1", "conv")
    x = xception_block(t, x, [1536, 1536, 2048], "exit_flow_block2", None)
    return x, skip


def aspp(t, c, atrous_branches):
    """AtrousSpatialPyramidPooling, model.py:192-233."""
    t._use("image_pooling", "image_pooling_BN", "aspp0", "aspp0_BN", "aspp1", "aspp2", "aspp3", "concat_projection",
           "concat_projection_BN")
    branches = [t.bn(t.conv(c, 256, 1, "image_pooling"), "image_pooling_BN"), t.bn(t.conv(c, 256, 1, "aspp0"), "aspp0_BN")]
    if atrous_branches:
        branches += [sepconv_bn(t, c, 256, f"aspp{i}") for i in (1, 2, 3)]
    x = t.conv(sum(branches), 256, 1, "concat_projection")
    return t.bn(x, "concat_projection_BN")


def decoder(t, x, skip):
    """Decoder, model.py:235-259."""
    t._use("feature_projection0", "feature_projection0_BN", "decoder_conv0", "decoder_conv1")
    s = t.bn(t.conv(skip, 48, 1, "feature_projection0"), "feature_projection0_BN")
    x = sepconv_bn(t, x + s, 256, "decoder_conv0")
    return sepconv_bn(t, x, 256, "decoder_conv1")


def logits(t, c, classes=21, pascal=True):
    """Final_Class_Prediction, model.py:296-306: the only Conv2D WITH a bias."""
    t._use("logits_semantic", "custom_logits_semantic")
    return t.conv(c, classes, 1, "logits_semantic" if (classes == 21 and pascal) else "custom_logits_semantic", use_bias=True)


def make_divisible(v, divisor, min_value=None):
    """_make_divisible, model.py:544-556."""
    min_value = divisor if min_value is None else min_value
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


def inverted_res_block(t, c, filters, block_id, alpha=1.0, expansion=6):
    """_inverted_res_block, model.py:426-461."""
    t._use("expanded_conv_", "expand", "expand_BN", "depthwise", "depthwise_BN", "project", "project_BN")
    p = f"expanded_conv_{block_id}_"
    out = make_divisible(int(filters * alpha), 8)
    x = t.bn(t.conv(c, expansion * c, 1, p + "expand"), p + "expand_BN")
    x = t.bn(t.depthwise(x, p + "depthwise"), p + "depthwise_BN")
    return t.bn(t.conv(x, out, 1, p + "project"), p + "project_BN")


def mobilenet_encoder(t, alpha=1.0):
    """EntryBlockMobile model.py:308-337, MobileNet_Backbone_Encoder :339-379."""
    t._use("Conv", "Conv_BN")
    x = t.bn(t.conv(3, make_divisible(32 * alpha, 8), 3, "Conv"), "Conv_BN")
    x = t.bn(t.depthwise(x, "expanded_conv_depthwise"), "expanded_conv_depthwise_BN")
    x = t.bn(t.conv(x, make_divisible(int(16 * alpha), 8), 1, "expanded_conv_project"), "expanded_conv_project_BN")
    for block_id, filters in enumerate([24, 24, 32, 32, 32, 64, 64, 64, 64, 96, 96, 96, 160, 160, 160, 320], start=1):
        x = inverted_res_block(t, x, filters, block_id, alpha)
    return x


def build(backbone, classes=21):
    t = Trace()
    if backbone == "xception":
        x, skip = xception_encoder(t)
        x = decoder(t, aspp(t, x, True), skip)
    else:
        x = aspp(t, mobilenet_encoder(t), False)           # the MobileNet decoder adds no layers (model.py:235-259)
    logits(t, x, classes)
    return t


def literals_of_reference(path=REFERENCE_MODEL):
    """Layer-name string literals in the reference's model.py: name='...' / name="..." / name=prefix + '...' and the block
    prefixes passed to _Xception_block / _SepConv_BN / _conv2d_same."""
    text = open(path).read()
    found = set(re.findall(r"name\s*=\s*(?:prefix\s*\+\s*)?['\"]([A-Za-z0-9_]+)['\"]", text))
    found |= set(re.findall(r"prefix\s*\+\s*\n?\s*['\"]([A-Za-z0-9_]+)['\"]", text))
    found |= set(re.findall(r"[\"'](entry_flow_block\d|exit_flow_block\d|aspp\d|decoder_conv\d)[\"']", text))
    found |= set(re.findall(r"f[\"'](middle_flow_unit_|expanded_conv_)\{", text))
    # names of layers without variables (activations, adds) and of the Model itself
    return {f for f in found if not re.search(r"relu|Relu|add$", f)}


def main():
    out = {}
    used = set()
    for key, backbone, classes in (("xception", "xception", 21), ("mobilenet", "mobilenet", 21), ("xception_5_classes", "xception", 5)):
        t = build(backbone, classes)
        out[key] = t.vars
        used |= t.literals
    if os.path.exists(REFERENCE_MODEL):
        ref = literals_of_reference()
        names = " ".join(n for b in out.values() for n in b)
        missing = sorted(f for f in ref if f not in names)
        assert not missing, f"layer-name literals of the reference that no derived variable contains: {missing}"
        unknown = sorted(f for f in used if f not in open(REFERENCE_MODEL).read())
        assert not unknown, f"name fragments used here that do not occur in the reference: {unknown}"
        print(f"cross-checked against {REFERENCE_MODEL}: {len(ref)} layer-name literals, all used")
    params = {b: sum(int(__import__('math').prod(s)) for n, s in v.items() if not n.endswith(("moving_mean", "moving_variance")))
              for b, v in out.items()}
    print({b: (len(v), params[b]) for b, v in out.items()})
    with open(os.path.join(HERE, "keras_variables.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
