"""Layer-by-layer, UNFUSED torch-CPU float32 restatement of the reference's DeepLabV3+
(Xception-65, OS16; and the MobileNetV2 variant, OS8) inference graph.  TEST INFRASTRUCTURE ONLY.

Follows model.py:64-147 (build_model), :149-190 (entry/middle/exit flow), :192-233 (ASPP),
:235-259 (Decoder), :296-306 (logits), :381-424 (_Xception_block), :463-508 (_SepConv_BN),
:510-541 (_conv2d_same); MobileNetV2: :308-337 (EntryBlockMobile), :339-379 (MobileNet_Backbone_Encoder),
:426-461 (_inverted_res_block) of the reference.  Keras layer semantics restated from keras==2.7.0:
Conv2D/DepthwiseConv2D 'same' padding (pad_total = max((ceil(in/s)-1)*s + k_eff - in, 0),
before = pad_total // 2), BatchNormalization inference y = gamma*(x-mean)/sqrt(var+eps)+beta,
Resizing(bilinear) = tf.image.resize half-pixel, GlobalAveragePooling2D(keepdims=True).

Weights: dict keyed '<keras layer name>/<kernel|depthwise_kernel|bias|gamma|beta|moving_mean|
moving_variance>' in Keras layouts (HWIO; depthwise [3,3,C,1]).  PARITY UNPINNED.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import tf_ops

F32 = torch.float32


def _same_pad(in_size, k_eff, stride):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return total // 2, total - total // 2


class OracleDeeplabV3Plus:
    """Xception backbone, OS=16 only (model.py:48-52: entry_block3_stride 2, middle rate 1,
    exit rates (1,2), atrous rates (6,12,18))."""

    def __init__(self, weights: dict, classes=21, last_activation=None, backbone="xception", OS=16, decoder="full",
                 first_upsample_size=(128, 128), class_prediction=True):
        """decoder: "full" (Decoder, model.py:235-259) | "dcnn" (Decoder_only_DCNN, :261-280) | "aspp"
        (Decoder_only_ASPP, :282-294); class_prediction=False: no logits layer (model.py:100-106)."""
        self.decoder, self.first_upsample_size, self.class_prediction = decoder, tuple(first_upsample_size), class_prediction
        self.w = {k: torch.as_tensor(np.asarray(v, dtype=np.float32)) for k, v in weights.items()}
        self.classes = classes
        self.last_activation = last_activation
        self.backbone = backbone
        # model.py:42-52
        self.entry_block3_stride, self.middle_block_rate, self.exit_block_rates, self.atrous_rates = \
            ((1, 2, (2, 4), (12, 24, 36)) if OS == 8 else (2, 1, (1, 2), (6, 12, 18)))

    # ---- Keras layers (NCHW inside) ---------------------------------------------------
    def conv(self, x, name, stride=1, padding="same", rate=1):
        k = self.w[name + "/kernel"]                       # [kh,kw,cin,cout]
        kh = k.shape[0]
        wt = k.permute(3, 2, 0, 1).contiguous()
        if padding == "same":
            k_eff = kh + (kh - 1) * (rate - 1)
            pt, pb = _same_pad(x.shape[2], k_eff, stride)
            pl, pr = _same_pad(x.shape[3], k_eff, stride)
            x = F.pad(x, (pl, pr, pt, pb))
        bias = self.w.get(name + "/bias")
        return F.conv2d(x, wt, bias=bias, stride=stride, dilation=rate)

    def dwconv(self, x, name, stride=1, padding="same", rate=1):
        k = self.w[name + "/depthwise_kernel"]             # [3,3,C,1]
        c = k.shape[2]
        wt = k.permute(2, 3, 0, 1).contiguous()            # [C,1,3,3]
        if padding == "same":
            k_eff = 3 + 2 * (rate - 1)
            pt, pb = _same_pad(x.shape[2], k_eff, stride)
            pl, pr = _same_pad(x.shape[3], k_eff, stride)
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, wt, stride=stride, dilation=rate, groups=c)

    def bn(self, x, name, eps=1e-3):
        g = self.w[name + "/gamma"].reshape(1, -1, 1, 1)
        b = self.w[name + "/beta"].reshape(1, -1, 1, 1)
        m = self.w[name + "/moving_mean"].reshape(1, -1, 1, 1)
        v = self.w[name + "/moving_variance"].reshape(1, -1, 1, 1)
        return (x - m) * (g / torch.sqrt(v + eps)) + b

    @staticmethod
    def resize(x, size):
        return tf_ops.resize_bilinear(x.permute(0, 2, 3, 1), size).permute(0, 3, 1, 2).contiguous()

    # ---- model.py:463-508 ---------------------------------------------------------------
    def sepconv_bn(self, x, prefix, stride=1, rate=1, depth_activation=False, eps=1e-3):
        if stride == 1:
            pad = "same"
        else:
            k_eff = 3 + 2 * (rate - 1)
            beg = (k_eff - 1) // 2
            end = (k_eff - 1) - beg
            x = F.pad(x, (beg, end, beg, end))
            pad = "valid"
        if not depth_activation:
            x = F.relu(x)
        x = self.dwconv(x, prefix + "_depthwise", stride=stride, padding=pad, rate=rate)
        x = self.bn(x, prefix + "_depthwise_BN", eps)
        if depth_activation:
            x = F.relu(x)
        x = self.conv(x, prefix + "_pointwise")
        x = self.bn(x, prefix + "_pointwise_BN", eps)
        if depth_activation:
            x = F.relu(x)
        return x

    # ---- model.py:381-424 ---------------------------------------------------------------
    def xception_block(self, inputs, prefix, skip_type, last_stride, rate=1, depth_activation=False,
                       return_skip=False):
        r = inputs
        skip = None
        for i in range(3):
            r = self.sepconv_bn(r, f"{prefix}_separable_conv{i + 1}", stride=last_stride if i == 2 else 1,
                                rate=rate, depth_activation=depth_activation)
            if i == 1:
                skip = r
        if skip_type == "conv":
            # model.py:510-541 with kernel_size=1: no padding; stride-2 samples x[::2, ::2]
            sc = self.conv(inputs, prefix + "_shortcut", stride=last_stride, padding="valid")
            sc = self.bn(sc, prefix + "_shortcut_BN")
            out = r + sc
        elif skip_type == "sum":
            out = r + inputs
        else:
            out = r
        return (out, skip) if return_skip else out

    # ---- model.py:426-461 ---------------------------------------------------------------
    def inverted_res_block(self, inputs, block_id, stride, rate, skip_connection):
        p = f"expanded_conv_{block_id}_"
        x = F.relu6(self.bn(self.conv(inputs, p + "expand"), p + "expand_BN"))
        x = F.relu6(self.bn(self.dwconv(x, p + "depthwise", stride=stride, rate=rate), p + "depthwise_BN"))
        x = self.bn(self.conv(x, p + "project"), p + "project_BN")
        return inputs + x if skip_connection else x

    def forward_mobilenet(self, x, stages):
        """model.py:94-101, 308-379: entry block, 16 inverted residual blocks (OS 8), ASPP = image pooling + aspp0."""
        x = F.relu6(self.bn(self.conv(x, "Conv", stride=2), "Conv_BN"))
        x = F.relu6(self.bn(self.dwconv(x, "expanded_conv_depthwise"), "expanded_conv_depthwise_BN"))
        x = self.bn(self.conv(x, "expanded_conv_project"), "expanded_conv_project_BN")
        stages["entry"] = x
        spec = [(1, 2, 1, False), (2, 1, 1, True), (3, 2, 1, False), (4, 1, 1, True), (5, 1, 1, True), (6, 1, 1, False),
                (7, 1, 2, True), (8, 1, 2, True), (9, 1, 2, True), (10, 1, 2, False), (11, 1, 2, True), (12, 1, 2, True),
                (13, 1, 2, False), (14, 1, 4, True), (15, 1, 4, True), (16, 1, 4, False)]
        for bid, stride, rate, skip in spec:
            x = self.inverted_res_block(x, bid, stride, rate, skip)
            if bid in (2, 5, 12):
                stages[f"block{bid}"] = x
        stages["exit"] = x
        fh, fw = x.shape[2:]
        pool = x.mean(dim=(2, 3), keepdim=True)
        pool = F.relu(self.bn(self.conv(pool, "image_pooling"), "image_pooling_BN", 1e-5))
        pool = self.resize(pool, (fh, fw))
        b0 = F.relu(self.bn(self.conv(x, "aspp0"), "aspp0_BN", 1e-5))
        x = torch.cat([pool, b0], dim=1)
        x = F.relu(self.bn(self.conv(x, "concat_projection"), "concat_projection_BN", 1e-5))
        stages["aspp"] = x
        return x

    # ---- model.py:64-147 ------------------------------------------------------------------
    def forward(self, images_nhwc, final_upsample=False, return_stages=False):
        x = torch.as_tensor(np.asarray(images_nhwc, dtype=np.float32)).permute(0, 3, 1, 2).contiguous()
        in_hw = x.shape[2:]
        stages = {}
        if self.backbone == "mobilenet":
            x = self.forward_mobilenet(x, stages)
            x = self.conv(x, "logits_semantic" if "logits_semantic/kernel" in self.w else "custom_logits_semantic")
            if final_upsample:
                x = self.resize(x, in_hw)
            if self.last_activation == "softmax":
                x = torch.softmax(x, dim=1)
            elif self.last_activation == "sigmoid":
                x = torch.sigmoid(x)
            out = x.permute(0, 2, 3, 1).contiguous().numpy()
            if return_stages:
                return out, {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in stages.items()}
            return out
        # entry flow (model.py:149-170)
        x = F.relu(self.bn(self.conv(x, "entry_flow_conv1_1", stride=2), "entry_flow_conv1_1_BN"))
        stages["conv1_1"] = x
        x = F.relu(self.bn(self.conv(x, "entry_flow_conv1_2"), "entry_flow_conv1_2_BN"))
        stages["conv1_2"] = x
        x = self.xception_block(x, "entry_flow_block1", "conv", 2)
        stages["block1"] = x
        x, skip = self.xception_block(x, "entry_flow_block2", "conv", 2, return_skip=True)
        x = self.xception_block(x, "entry_flow_block3", "conv", self.entry_block3_stride)
        stages["entry"] = x
        # middle flow (model.py:172-179)
        for i in range(16):
            x = self.xception_block(x, f"middle_flow_unit_{i + 1}", "sum", 1, rate=self.middle_block_rate)
        stages["middle"] = x
        # exit flow (model.py:181-190)
        x = self.xception_block(x, "exit_flow_block1", "conv", 1, rate=self.exit_block_rates[0])
        x = self.xception_block(x, "exit_flow_block2", None, 1, rate=self.exit_block_rates[1], depth_activation=True)
        stages["exit"] = x
        if self.decoder == "dcnn":
            # Decoder_only_DCNN (model.py:261-280): projection of the encoder output, Resizing(first_upsample_size)
            x = F.relu(self.bn(self.conv(x, "feature_projection0"), "feature_projection0_BN", 1e-5))
            x = self.resize(x, self.first_upsample_size)
        else:
            # ASPP (model.py:192-233)
            fh, fw = x.shape[2:]
            pool = x.mean(dim=(2, 3), keepdim=True)
            pool = F.relu(self.bn(self.conv(pool, "image_pooling"), "image_pooling_BN", 1e-5))
            pool = self.resize(pool, (fh, fw))
            b0 = F.relu(self.bn(self.conv(x, "aspp0"), "aspp0_BN", 1e-5))
            b1 = self.sepconv_bn(x, "aspp1", rate=self.atrous_rates[0], depth_activation=True)
            b2 = self.sepconv_bn(x, "aspp2", rate=self.atrous_rates[1], depth_activation=True)
            b3 = self.sepconv_bn(x, "aspp3", rate=self.atrous_rates[2], depth_activation=True)
            x = torch.cat([pool, b0, b1, b2, b3], dim=1)
            x = F.relu(self.bn(self.conv(x, "concat_projection"), "concat_projection_BN", 1e-5))
            stages["aspp"] = x
            if self.decoder == "aspp":
                x = self.resize(x, self.first_upsample_size)          # Decoder_only_ASPP (model.py:282-294)
            else:
                # decoder (model.py:235-259)
                x = self.resize(x, skip.shape[2:])
                dskip = F.relu(self.bn(self.conv(skip, "feature_projection0"), "feature_projection0_BN", 1e-5))
                x = torch.cat([x, dskip], dim=1)
        x = self.sepconv_bn(x, "decoder_conv0", depth_activation=True, eps=1e-5)
        x = self.sepconv_bn(x, "decoder_conv1", depth_activation=True, eps=1e-5)
        stages["decoder"] = x
        # logits (model.py:296-306)
        if self.class_prediction:
            x = self.conv(x, "logits_semantic" if "logits_semantic/kernel" in self.w else "custom_logits_semantic")
        if final_upsample:
            x = self.resize(x, in_hw)
        if self.last_activation == "softmax":
            x = torch.softmax(x, dim=1)
        elif self.last_activation == "sigmoid":
            x = torch.sigmoid(x)
        out = x.permute(0, 2, 3, 1).contiguous().numpy()
        if return_stages:
            return out, {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in stages.items()}
        return out

    def predict(self, x, batch_size=16):
        x = np.asarray(x, dtype=np.float32)
        outs = [self.forward(x[i:i + batch_size]) for i in range(0, x.shape[0], batch_size)]
        return np.concatenate(outs, axis=0)
