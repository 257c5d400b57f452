"""Device execution plan for the DeepLabV3+ forward pass (Xception-65 OS16; MobileNetV2 OS8).

The reference runs 147 Keras layers as separate library kernels with an HBM round trip each
(``model.predict``, superresolution_scripts/augmentation_utils.py:76).  Here the graph of
model.py:64-147 is lowered once into a flat list of C-ABI launches over pre-allocated NHWC
buffers:
  * every BatchNorm is folded into the preceding conv's weights + bias (inference affine);
  * ReLUs, the residual ``Add`` and the stride-2 row gather of shortcut convs live in the
    epilogue / prologue of the kernel that produces or consumes the tensor;
  * ``Concatenate`` never copies: producers write straight into channel slices of the concat
    buffer (``ldy`` = total channels);
  * activation buffers are recycled by liveness so the working set of the middle flow stays
    inside the 256 MiB Infinity Cache.
Replaying the plan is a loop of ctypes calls on the current stream (hipGraph-capturable: the
launch functions allocate nothing and never synchronise).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib, ops, weights as W

f32 = torch.float32

# Range of the split-f16 GEMM operands (hi = f16(v), lo = f16(v - hi)): above 2^15 the f16 halves are a factor 2 from
# overflow (the kernels saturate at 65504 instead of producing infinities, but the result is then inexact); a tensor whose
# LARGEST magnitude is below 2^-12 keeps fewer than 13 of its 22 bits (lo falls under f16's subnormal spacing 2^-24).
SPLIT_MAX = 2.0 ** 15
SPLIT_W_MIN = 2.0 ** -12
SPLIT_ACT_MIN = 2.0 ** -10
SPLIT_HEADROOM = 2.0          # calibration routes a layer whose probe maximum is within this factor of SPLIT_MAX


class _Buf:
    """NHWC activation: logical shape (b,h,w,c) stored with a pixel stride ``ld`` >= c that is a
    multiple of 32 floats, so every pixel row starts on a 128-byte line (728 -> 736, 304 -> 320):
    the GEMM's 128-byte A-row pieces and epilogue stores then map to whole cache lines."""
    __slots__ = ("t", "shape", "ld")

    def __init__(self, t, shape, ld):
        self.t = t
        self.shape = shape
        self.ld = ld

    @property
    def ptr(self):
        return self.t.data_ptr()


class _Pool:
    """Exact-size free list: the net repeats a handful of activation sizes."""

    def __init__(self, device, zero_fill=False):
        self.device = device
        self.zero_fill = zero_fill          # range calibration: unwritten lanes must read as 0, not as garbage
        self.free = {}
        self.total_bytes = 0
        self.owned = []          # every tensor ever handed out: the plan stores raw pointers, so the pool
                                 # (kept alive by the plan) must own the memory for the plan's lifetime

    def get(self, numel):
        lst = self.free.get(numel)
        if lst:
            return lst.pop()
        self.total_bytes += 4 * numel
        if os.environ.get("ASR_POISON"):            # debugging aid: fill fresh buffers with a sentinel value
            t = torch.full((numel,), float(os.environ["ASR_POISON"]), dtype=f32, device=self.device)
        elif self.zero_fill:
            t = torch.zeros(numel, dtype=f32, device=self.device)
        else:
            t = torch.empty(numel, dtype=f32, device=self.device)
        self.owned.append(t)
        return t

    def put(self, t):
        self.free.setdefault(t.numel(), []).append(t)


def _same_pad(in_size, k_eff, stride):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return out, total // 2


class DeeplabEngine:
    """Folded / packed parameters on the device + per-(batch, H, W) launch plans."""

    FUSIONS = ("presplit", "fused_stem", "fused_sepconv", "fused_aspp")

    def __init__(self, weights: dict, classes=21, device=None, precision=None, backbone="xception", alpha=1.0, OS=16,
                 decoder="full", first_upsample_size=(128, 128), class_prediction=True, disable=None):
        """decoder: "full" | "dcnn" | "aspp" = Decoder / Decoder_only_DCNN / Decoder_only_ASPP (model.py:235-294; the
        last two resize to first_upsample_size instead of the skip's size); class_prediction=False returns the decoder's
        256-channel features instead of the logits (model.py:104-106).
        disable: names from FUSIONS to leave out of the plan (profiling / A-B runs; default: $ASR_DISABLE, comma separated) --
        "presplit" (depthwise hands split-f16 operands to the LDS-DMA GEMM), "fused_stem" (conv1_1 + conv1_2 in one kernel),
        "fused_sepconv" (entry-flow block 1 separable convs in one kernel), "fused_aspp" (three dilation rates from one LDS
        plane).  Every combination computes the same layers; results agree to f32 rounding.
        precision: 'f32' = v_mfma_f32_32x32x2_f32 everywhere (exact f32 fmaf chains);
        'f16x3' = split-f16 MFMA (hi*hi + hi*lo + lo*hi, f32 accumulate; f32-grade results, ~2.4x faster)
        for the pointwise GEMMs with more than 32 output channels (the logits stay on exact f32).  Default: $ASR_PRECISION or 'f16x3'."""
        self.device = device or _lib.require_gpu()
        if disable is None:
            disable = [v for v in os.environ.get("ASR_DISABLE", "").split(",") if v]
        unknown = set(disable) - set(self.FUSIONS)
        if unknown:
            raise ValueError(f"unknown fusion name(s) {sorted(unknown)}; choose from {self.FUSIONS}")
        self.disabled = frozenset(disable)
        self.precision = precision or os.environ.get("ASR_PRECISION", "f16x3")
        if self.precision not in ("f32", "f16x3"):
            raise ValueError(f"precision must be 'f32' or 'f16x3', got {self.precision!r}")
        if backbone not in ("xception", "mobilenet"):
            raise ValueError("Backbone must be either xception or mobilenet")
        self.backbone = backbone
        self.alpha = alpha
        if backbone == "xception" and OS not in (8, 16):
            raise ValueError("OS must be 8 or 16 for the Xception backbone")
        # model.py:42-52: OS 8 trades the stride of entry block 3 for dilation in everything after it
        self.OS = OS if backbone == "xception" else 8
        self.entry_block3_stride, self.middle_block_rate, self.exit_block_rates, self.atrous_rates = \
            ((1, 2, (2, 4), (12, 24, 36)) if self.OS == 8 else (2, 1, (1, 2), (6, 12, 18)))
        self.output_stride = 4 if backbone == "xception" else 8      # input size / logits size
        self.classes = classes
        if decoder not in ("full", "dcnn", "aspp"):
            raise ValueError(f"decoder must be 'full', 'dcnn' or 'aspp', got {decoder!r}")
        if backbone != "xception" and decoder != "full":
            raise ValueError("the modified decoders exist for the xception backbone only (model.py:80)")
        self.decoder = decoder
        self.first_upsample_size = tuple(int(v) for v in first_upsample_size)
        self.class_prediction = bool(class_prediction)
        self.logits_name = "logits_semantic" if "logits_semantic/kernel" in weights else "custom_logits_semantic"
        self.p = {}
        self._plans = {}
        self.routed_f32 = {}          # conv layer name -> reason: layers taken off the split-f16 kernels (range guard)
        self._host_weights = weights  # kept by reference: a routed layer is re-packed for the exact-f32 kernel
        self._upload(weights)

    # -- parameters -----------------------------------------------------------------------------
    def _dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def _put_conv(self, name, bn, eps, pack=True):
        self._conv_meta = getattr(self, "_conv_meta", {})
        self._conv_meta[name] = (bn, eps, pack)
        k, b = (W.fold_conv_bn(self._w, name, bn, eps) if bn else
                (self._w[name + "/kernel"].reshape(-1, self._w[name + "/kernel"].shape[-1]).astype(np.float32),
                 self._w.get(name + "/bias")))
        kd = self._dev(k)
        conv = name == "entry_flow_conv1_2"                       # the one dense 3x3 on the matrix path (implicit GEMM)
        # <= 32 output channels (the logits) stay on the exact-f32 kernel: HBM-bound there and faster (401 against 510 us at
        # 21 channels); from 33 up the split-f16 kernel wins on its 128 x 64 tile (feature_projection0, 48 channels: 630 -> 417 us)
        split = pack and self.precision == "f16x3" and (k.shape[1] > 32 or conv)
        if split and name not in self.routed_f32:
            # range guard of the split-f16 arithmetic, weight side (hi = f16(w) must be finite and carry bits): a folded
            # kernel outside [2^-12, 2^15) in magnitude goes to the exact-f32 MFMA kernel
            wmax = float(np.abs(k).max())
            if not (SPLIT_W_MIN <= wmax < SPLIT_MAX):
                self.routed_f32[name] = f"max |w| = {wmax:.3g} outside [{SPLIT_W_MIN:.3g}, {SPLIT_MAX:.3g})"
        if name in self.routed_f32:
            split = False
        wdev = (ops.pack_pw_weights_f16x3(kd) if split else ops.pack_pw_weights(kd)) if pack else kd
        fn = ("asr_conv3x3_mfma_f16x3" if split else "asr_conv3x3_mfma_f32") if conv else \
             ("asr_pwconv_mfma_f16x3" if split else "asr_pwconv_mfma_f32")
        self.p[name] = dict(w=wdev, b=self._dev(b) if b is not None else None, k=k.shape[0], n=k.shape[1], fn=fn)

    def _put_dw(self, name, bn, eps):
        k, b = W.fold_dw_bn(self._w, name, bn, eps)
        self.p[name] = dict(w=self._dev(k), b=self._dev(b), c=k.shape[-1])

    def _put_sep(self, prefix, eps):
        self._put_dw(prefix + "_depthwise", prefix + "_depthwise_BN", eps)
        self._put_conv(prefix + "_pointwise", prefix + "_pointwise_BN", eps)

    def _upload_mobilenet(self, weights):
        """model.py:308-379, 426-461: BN folded into every conv / depthwise; ReLU6 lives in the kernels' epilogues."""
        self._w = weights
        e3, e5 = W.XCEPTION_BN_EPS, W.HEAD_BN_EPS
        k, b = W.fold_conv_bn(weights, "Conv", "Conv_BN", e3)         # HWIO for the direct stem kernel
        self.p["Conv"] = dict(w=self._dev(k), b=self._dev(b), k=k.shape[0], n=k.shape[1])
        self._put_dw("expanded_conv_depthwise", "expanded_conv_depthwise_BN", e3)
        self._put_conv("expanded_conv_project", "expanded_conv_project_BN", e3)
        for bid, *_ in W.mobilenet_blocks(self.alpha):
            p = f"expanded_conv_{bid}_"
            self._put_conv(p + "expand", p + "expand_BN", e3)
            self._put_dw(p + "depthwise", p + "depthwise_BN", e3)
            self._put_conv(p + "project", p + "project_BN", e3)
        self._put_conv("image_pooling", "image_pooling_BN", e5)
        self._put_conv("aspp0", "aspp0_BN", e5)
        self._put_conv("concat_projection", "concat_projection_BN", e5)
        self._put_conv(self.logits_name, None, None)
        torch.cuda.synchronize(self.device)
        del self._w

    def _upload(self, weights):
        if self.backbone == "mobilenet":
            return self._upload_mobilenet(weights)
        self._w = weights
        e3, e5 = W.XCEPTION_BN_EPS, W.HEAD_BN_EPS
        # entry_flow_conv1_1 keeps its HWIO layout for the direct kernel
        k, b = W.fold_conv_bn(weights, "entry_flow_conv1_1", "entry_flow_conv1_1_BN", e3)
        self.p["entry_flow_conv1_1"] = dict(w=self._dev(k), b=self._dev(b), k=k.shape[0], n=k.shape[1])
        self._put_conv("entry_flow_conv1_2", "entry_flow_conv1_2_BN", e3)
        for prefix, _cin, _f, skip, _s, _r, _da in W.xception_blocks():
            for i in range(3):
                self._put_sep(f"{prefix}_separable_conv{i + 1}", e3)
            if skip == "conv":
                self._put_conv(prefix + "_shortcut", prefix + "_shortcut_BN", e3)
        if self.decoder != "dcnn":      # Decoder_only_DCNN never reaches the ASPP (Keras drops layers off the output's path)
            self._put_conv("image_pooling", "image_pooling_BN", e5)
            self._put_conv("aspp0", "aspp0_BN", e5)
            for i in (1, 2, 3):
                self._put_sep(f"aspp{i}", e3)
            # branch-major stack of the three folded ASPP depthwise kernels for the fused kernel
            self.p["aspp_dw3"] = dict(w=torch.stack([self.p[f"aspp{i}_depthwise"]["w"] for i in (1, 2, 3)]).contiguous(),
                                      b=torch.stack([self.p[f"aspp{i}_depthwise"]["b"] for i in (1, 2, 3)]).contiguous())
            self._put_conv("concat_projection", "concat_projection_BN", e5)
        if self.decoder != "aspp":
            self._put_conv("feature_projection0", "feature_projection0_BN", e5)
        self._put_sep("decoder_conv0", e5)
        self._put_sep("decoder_conv1", e5)
        if self.class_prediction:
            self._put_conv(self.logits_name, None, None)
        want = {"full": 304, "dcnn": 48, "aspp": 256}[self.decoder]
        if self.p["decoder_conv0_depthwise"]["c"] != want:
            raise ValueError(f"decoder_conv0_depthwise has {self.p['decoder_conv0_depthwise']['c']} channels; the "
                             f"'{self.decoder}' decoder needs {want} (weights built for another decoder?)")
        torch.cuda.synchronize(self.device)
        del self._w

    def shift_logit_bias(self, class_id, delta):
        """Add ``delta`` to the bias of one class of the logits layer (model.py:296-306), in place on the device."""
        self.p[self.logits_name]["b"][class_id] += float(delta)

    # -- plan construction ----------------------------------------------------------------------
    def _build_plan(self, B, H, Wd, zero_fill=False):
        pool = _Pool(self.device, zero_fill)
        steps = []          # (name, args, kind, flops, bytes)
        live = []           # buffers to release after a given step index

        def new(shape, pad=True):
            c = shape[-1]
            ld = -(-c // 32) * 32 if (pad and c >= 32) else c
            n = int(np.prod(shape[:-1])) * ld
            return _Buf(pool.get(n), tuple(shape), ld)

        def release(buf):
            pool.put(buf.t)

        outs = []           # output buffer of every step (debug capture)

        def add(name, args, kind, flops=0, nbytes=0, label="", out=None):
            steps.append((name, tuple(args), kind, float(flops), float(nbytes), label))
            outs.append(out)

        def pw(x, name, out=None, out_off=0, relu=False, res=None, sub=1, pad_out=True):
            p = self.p[name]
            b, h, w, c = x.shape
            ho, wo = (-(-h // sub), -(-w // sub)) if sub > 1 else (h, w)
            if out is None:
                out = new((b, ho, wo, p["n"]), pad=pad_out)
            ldy = out.ld
            m = b * ho * wo
            add(p.get("fn", "asr_pwconv_mfma_f32"),
                (x.ptr, p["w"].data_ptr(), p["b"].data_ptr() if p["b"] is not None else None,
                 res.ptr if res is not None else None, out.ptr + 4 * out_off, m, p["k"], p["n"], x.ld, ldy,
                 res.ld if res is not None else 0, int(relu), sub, h if sub > 1 else 0, w if sub > 1 else 0),
                # kind: split-f16 GEMMs are "pw16"; the <= 64-channel ones ("pw16s": the 128 x 64 tile, HBM-bound) and the
                # exact-f32 ones ("pw") are booked apart, so that bench.py prices each family against its own bound
                ("pw16" if p["n"] > 64 else "pw16s") if p.get("fn", "").endswith("f16x3") else "pw", 2.0 * m * p["k"] * p["n"],
                4.0 * (m * p["k"] + m * p["n"] * (2 if res is not None else 1) + p["k"] * p["n"]),
                label=f"{name} M={m} K={p['k']} N={p['n']}", out=out)
            return out

        def dw(x, name, stride, rate, pre_relu, post_relu, tf_same=False):
            p = self.p[name]
            b, h, w, c = x.shape
            pad = rate                      # stride 1 'same' and the explicit ZeroPadding2D both give `rate`
            ho, wo = (h, w) if stride == 1 else ((h + 2 * pad - (2 * rate + 1)) // stride + 1,
                                                 (w + 2 * pad - (2 * rate + 1)) // stride + 1)
            pad_t = pad_l = pad
            if tf_same and stride > 1:      # Keras padding='same' on a strided conv: TF SAME puts the odd pixel after
                ho, pad_t = _same_pad(h, 2 * rate + 1, stride)
                wo, pad_l = _same_pad(w, 2 * rate + 1, stride)
            out = new((b, ho, wo, c))
            add("asr_dwconv3x3_nhwc_f32",
                (x.ptr, p["w"].data_ptr(), p["b"].data_ptr(), out.ptr, b, h, w, c, stride, rate, pad_t, pad_l, ho, wo, x.ld,
                 out.ld, int(pre_relu), int(post_relu), 0),
                "dw", 18.0 * b * ho * wo * c, 4.0 * (b * h * w * c + b * ho * wo * c + 10 * c),
                label=f"{name} {h}x{w}x{c} s{stride} r{rate}", out=out)
            return out

        def sepconv(x, prefix, stride=1, rate=1, depth_act=False, **pw_kw):
            pp, pd = self.p[prefix + "_pointwise"], self.p[prefix + "_depthwise"]
            b, h, w, c = x.shape
            ho, wo = (h, w) if stride == 1 else ((h + 2 * rate - (2 * rate + 1)) // stride + 1,
                                                 (w + 2 * rate - (2 * rate + 1)) // stride + 1)
            srows = 16 if ho <= 64 else 32           # csrc/dwconv.hip: ASR_DW_SMALL_MAX
            split_ok = (pp.get("fn", "").endswith("f16x3") and (-(-pp["n"] // 128) * 128) % 256 == 0 and not pw_kw.get("out_off")
                        and pw_kw.get("sub", 1) == 1 and ((stride == 1 and rate in (1, 2)) or (stride == 2 and rate == 1))
                        and ho % srows == 0 and c % 8 == 0 and b <= 65535 and b * ho * wo >= 256 and "presplit" not in self.disabled)
            fused_ok = (pp.get("fn", "").endswith("f16x3") and stride == 1 and rate == 1 and c in (64, 128) and pp["n"] == 128
                        and not pw_kw.get("out_off") and pw_kw.get("sub", 1) == 1 and pw_kw.get("res") is None
                        and pp["b"] is not None and x.ld % 4 == 0 and "fused_sepconv" not in self.disabled)
            if fused_ok:
                # both halves in one kernel: the depthwise output lives in LDS only (entry-flow block 1 at 256 x 256)
                out = pw_kw.get("out")
                if out is None:
                    out = new((b, ho, wo, pp["n"]), pad=pw_kw.get("pad_out", True))
                add("asr_sepconv_fused_f16x3",
                    (x.ptr, pd["w"].data_ptr(), pd["b"].data_ptr(), pp["w"].data_ptr(), pp["b"].data_ptr(), out.ptr, b, h, w, c,
                     pp["n"], x.ld, out.ld, int(not depth_act), int(depth_act), int(depth_act)),
                    "sepconv", 18.0 * b * h * w * c + 2.0 * b * h * w * c * pp["n"], 4.0 * (b * h * w * c + b * h * w * pp["n"]),
                    label=f"{prefix} fused dw+pw {h}x{w}x{c}->{pp['n']}", out=out)
                return out
            if split_ok:
                # depthwise writes the pointwise GEMM's A operand directly as split-f16 chunks; the GEMM takes both operands
                # by LDS-DMA on a 256 x 256 tile (bit-identical to the f32 hand-off, see include/asr_hip.h)
                chunks = -(-c // 32)
                t = new((b, ho, wo, chunks * 32))
                add("asr_dwconv3x3_nhwc_split_f16",
                    (x.ptr, pd["w"].data_ptr(), pd["b"].data_ptr(), t.ptr, b, h, w, c, stride, rate, rate, rate, ho, wo, x.ld, chunks,
                     int(not depth_act), int(depth_act)),
                    "dw", 18.0 * b * ho * wo * c, 4.0 * (b * h * w * c + b * ho * wo * c + 10 * c),
                    label=f"{prefix}_depthwise {h}x{w}x{c} s{stride} r{rate} split", out=t)
                out, res = pw_kw.get("out"), pw_kw.get("res")
                if out is None:
                    out = new((b, ho, wo, pp["n"]), pad=pw_kw.get("pad_out", True))
                m = b * ho * wo
                add("asr_pwconv_mfma_f16x3_presplit",
                    (t.ptr, pp["w"].data_ptr(), pp["b"].data_ptr() if pp["b"] is not None else None,
                     res.ptr if res is not None else None, out.ptr, m, pp["k"], pp["n"], chunks, out.ld,
                     res.ld if res is not None else 0, int(depth_act)),
                    "pw16", 2.0 * m * pp["k"] * pp["n"],
                    4.0 * (m * pp["k"] + m * pp["n"] * (2 if res is not None else 1) + pp["k"] * pp["n"]),
                    label=f"{prefix}_pointwise M={m} K={pp['k']} N={pp['n']} presplit", out=out)
                release(t)
                return out
            t = dw(x, prefix + "_depthwise", stride, rate, pre_relu=not depth_act, post_relu=depth_act)
            y = pw(t, prefix + "_pointwise", relu=depth_act, **pw_kw)
            release(t)
            return y

        def block(x, prefix, skip, last_stride, rate, depth_act, return_skip=False, keep_input=False):
            sc = None
            if skip == "conv":
                sc = pw(x, prefix + "_shortcut", sub=last_stride)
            elif skip == "sum":
                sc = x
            r1 = sepconv(x, prefix + "_separable_conv1", 1, rate, depth_act)
            r2 = sepconv(r1, prefix + "_separable_conv2", 1, rate, depth_act)
            release(r1)
            r3 = sepconv(r2, prefix + "_separable_conv3", last_stride, rate, depth_act, res=sc)
            if not return_skip:
                release(r2)
            if skip == "conv":
                release(sc)
            if not keep_input:
                release(x)
            return (r3, r2) if return_skip else r3

        x_in = new((B, H, Wd, 3), pad=False)
        h1, pt = _same_pad(H, 3, 2)
        w1, pl = _same_pad(Wd, 3, 2)
        if self.backbone == "mobilenet":
            # ---- EntryBlockMobile (model.py:308-337) ----
            p = self.p["Conv"]
            c0 = p["n"]
            a1 = new((B, h1, w1, c0))
            add("asr_conv3x3_stem_f16x3" if (self.precision == "f16x3" and c0 == 32) else "asr_conv3x3_direct_f32",
                (x_in.ptr, p["w"].data_ptr(), p["b"].data_ptr(), a1.ptr, B, H, Wd, 3, c0, 2, pt, pl,
                                           h1, w1, 3, a1.ld, 2), "conv", 2.0 * B * h1 * w1 * 27 * c0,
                4.0 * (B * H * Wd * 3 + B * h1 * w1 * c0), label="Conv", out=a1)
            t = dw(a1, "expanded_conv_depthwise", 1, 1, False, 2)
            release(a1)
            x = pw(t, "expanded_conv_project")
            release(t)
            # ---- MobileNet_Backbone_Encoder (model.py:339-379): 16 inverted residual blocks ----
            for bid, _cin, _cout, stride, rate, skip in W.mobilenet_blocks(self.alpha):
                pfx = f"expanded_conv_{bid}_"
                e = pw(x, pfx + "expand", relu=2)
                d = dw(e, pfx + "depthwise", stride, rate, False, 2, tf_same=True)
                release(e)
                y = pw(d, pfx + "project", res=x if skip else None)
                release(d)
                release(x)
                x = y
            # ---- ASPP without atrous branches (model.py:192-210, 224-231); no decoder (model.py:94-101) ----
            b, fh, fw, fc = x.shape
            cat = new((b, fh, fw, 512))
            pooled = new((b, 1, 1, fc), pad=False)
            add("asr_gap_f32", (x.ptr, pooled.ptr, b, fh * fw, fc, x.ld), "misc", b * fh * fw * fc, 4.0 * b * fh * fw * fc,
                label="gap", out=pooled)
            pp = pw(pooled, "image_pooling", relu=True)
            add("asr_resize_bilinear_f32", (pp.ptr, cat.ptr, b, 1, 1, 256, fh, fw, pp.ld, cat.ld), "misc", 0,
                4.0 * b * fh * fw * 256, out=cat)
            release(pooled)
            release(pp)
            pw(x, "aspp0", out=cat, out_off=256, relu=True)
            release(x)
            x = pw(cat, "concat_projection", relu=True)
            release(cat)
            logits = pw(x, self.logits_name, pad_out=False)
            release(x)
            return dict(steps=steps, outs=outs, pool=pool, x_in=x_in, logits=logits, pool_bytes=pool.total_bytes,
                        out_shape=(B, fh, fw, self.classes))
        # ---- entry flow (model.py:149-170) ----
        p1, p2 = self.p["entry_flow_conv1_1"], self.p["entry_flow_conv1_2"]
        a2 = new((B, h1, w1, 64))
        if (self.precision == "f16x3" and p2["fn"].endswith("f16x3") and H % 2 == 0 and Wd % 2 == 0
                and "fused_stem" not in self.disabled):
            # conv1_1 + conv1_2 in one kernel: the 32-channel intermediate stays in LDS
            add("asr_entry_stem_f16x3", (x_in.ptr, p1["w"].data_ptr(), p1["b"].data_ptr(), p2["w"].data_ptr(), p2["b"].data_ptr(),
                                         a2.ptr, B, H, Wd, 3, a2.ld), "conv", 2.0 * B * h1 * w1 * (27 * 32 + 288 * 64),
                4.0 * (B * H * Wd * 3 + B * h1 * w1 * 64), label="entry_flow_conv1_1+conv1_2 fused", out=a2)
        else:
            a1 = new((B, h1, w1, 32))
            add("asr_conv3x3_stem_f16x3" if self.precision == "f16x3" else "asr_conv3x3_direct_f32",
                (x_in.ptr, p1["w"].data_ptr(), p1["b"].data_ptr(), a1.ptr, B, H, Wd, 3, 32, 2, pt, pl,
                 h1, w1, 3, 32, 1), "conv", 2.0 * B * h1 * w1 * 27 * 32,
                4.0 * (B * H * Wd * 3 + B * h1 * w1 * 32), label="entry_flow_conv1_1 stem", out=a1)
            add(p2["fn"], (a1.ptr, p2["w"].data_ptr(), p2["b"].data_ptr(), a2.ptr, B, h1, w1, 32, 64, 1, 1, 1, h1,
                           w1, 32, 64, 1), "conv", 2.0 * B * h1 * w1 * 288 * 64,
                4.0 * (B * h1 * w1 * 96), label="entry_flow_conv1_2 conv3x3", out=a2)
            release(a1)
        x = block(a2, "entry_flow_block1", "conv", 2, 1, False)
        x, skip = block(x, "entry_flow_block2", "conv", 2, 1, False, return_skip=True)
        x = block(x, "entry_flow_block3", "conv", self.entry_block3_stride, 1, False)
        # ---- middle flow (model.py:172-179) ----
        for i in range(16):
            x = block(x, f"middle_flow_unit_{i + 1}", "sum", 1, self.middle_block_rate, False)
        # ---- exit flow (model.py:181-190) ----
        x = block(x, "exit_flow_block1", "conv", 1, self.exit_block_rates[0], False)
        x = block(x, "exit_flow_block2", None, 1, self.exit_block_rates[1], True)
        b, fh, fw, fc = x.shape
        if self.decoder == "dcnn":
            # ---- Decoder_only_DCNN (model.py:261-280): the encoder output goes straight to the 48-channel projection ----
            release(skip)
            feat = pw(x, "feature_projection0", relu=True)
            release(x)
            return self._finish_plan(feat, 48, None, steps, outs, pool, x_in, new, release, add, pw, sepconv, B)
        # ---- ASPP (model.py:192-233) ----
        cat = new((b, fh, fw, 1280))
        pooled = new((b, 1, 1, fc), pad=False)
        add("asr_gap_f32", (x.ptr, pooled.ptr, b, fh * fw, fc, x.ld), "misc", b * fh * fw * fc, 4.0 * b * fh * fw * fc,
            label="gap", out=pooled)
        pp = pw(pooled, "image_pooling", relu=True)
        add("asr_resize_bilinear_f32", (pp.ptr, cat.ptr, b, 1, 1, 256, fh, fw, pp.ld, cat.ld), "misc", 0,
            4.0 * b * fh * fw * 256, out=cat)
        release(pooled)
        release(pp)
        pw(x, "aspp0", out=cat, out_off=256, relu=True)
        rates = self.atrous_rates
        # the library's own geometry check (LDS per residue class AND its column limit), so the plan never meets
        # ASR_ERR_UNSUPPORTED at run time: planes it cannot stage take the three-launch path below
        fused_ok = bool(_lib.load().asr_aspp_dwconv3_supported(fh, fw, int(rates[0]), int(rates[1]), int(rates[2])))
        if fused_ok and "fused_aspp" not in self.disabled:
            # the three dilated depthwise convs read the same input: one fused launch stages each residue class of the
            # plane (modulo gcd(rates), on which the taps close) in LDS once -- input read from HBM 1x instead of 3x, on
            # planes of any size (csrc/dwconv.hip: aspp_dw3_phase_kernel)
            ts = [new((b, fh, fw, fc)) for _ in rates]
            p3 = self.p["aspp_dw3"]
            split = (all(self.p[f"aspp{i + 1}_pointwise"].get("fn", "").endswith("f16x3") for i in range(3)) and fc % 32 == 0
                     and b * fh * fw >= 256 and "presplit" not in self.disabled)
            add("asr_aspp_dwconv3_nhwc_split_f16" if split else "asr_aspp_dwconv3_nhwc_f32",
                (x.ptr, p3["w"].data_ptr(), p3["b"].data_ptr(), ts[0].ptr, ts[1].ptr, ts[2].ptr, b, fh, fw, fc,
                 rates[0], rates[1], rates[2], x.ld, fc // 32 if split else ts[0].ld, 0, 1),
                "dw", 3 * 18.0 * b * fh * fw * fc, 4.0 * (1 + 3) * b * fh * fw * fc,       # bytes MOVED: input once, three outputs
                label=f"aspp_dw3 {fh}x{fw}x{fc} r{rates[0]}/{rates[1]}/{rates[2]} fused" + (" split" if split else ""), out=ts)
            for i, t in enumerate(ts):
                if split:
                    pp = self.p[f"aspp{i + 1}_pointwise"]
                    m = b * fh * fw
                    add("asr_pwconv_mfma_f16x3_presplit",
                        (t.ptr, pp["w"].data_ptr(), pp["b"].data_ptr(), None, cat.ptr + 4 * (512 + 256 * i), m, pp["k"], pp["n"],
                         fc // 32, cat.ld, 0, 1),
                        "pw16", 2.0 * m * pp["k"] * pp["n"], 4.0 * (m * pp["k"] + m * pp["n"] + pp["k"] * pp["n"]),
                        label=f"aspp{i + 1}_pointwise M={m} K={pp['k']} N={pp['n']} presplit", out=cat)
                else:
                    pw(t, f"aspp{i + 1}_pointwise", out=cat, out_off=512 + 256 * i, relu=True)
                release(t)
        else:
            for i, rate in enumerate(rates):
                t = dw(x, f"aspp{i + 1}_depthwise", 1, rate, pre_relu=False, post_relu=True)
                pw(t, f"aspp{i + 1}_pointwise", out=cat, out_off=512 + 256 * i, relu=True)
                release(t)
        release(x)
        x = pw(cat, "concat_projection", relu=True)
        release(cat)
        if self.decoder == "aspp":     # Decoder_only_ASPP (model.py:282-294): no skip connection
            release(skip)
            skip = None
        return self._finish_plan(x, 256, skip, steps, outs, pool, x_in, new, release, add, pw, sepconv, B)

    def _finish_plan(self, feat, fch, skip, steps, outs, pool, x_in, new, release, add, pw, sepconv, B):
        """Decoder tail shared by the three decoders: resize -> [concat the projected skip] -> decoder_conv0/1 ->
        [logits] (model.py:235-306)."""
        b, fh, fw, _ = feat.shape
        if skip is not None:
            sb, sh, sw, _ = skip.shape
            cat2 = new((sb, sh, sw, 304))
        else:
            sh, sw = self.first_upsample_size          # Resizing(*first_upsample_size) (model.py:271-272, 285-286)
            cat2 = new((b, sh, sw, fch))
        add("asr_resize_bilinear_f32", (feat.ptr, cat2.ptr, b, fh, fw, fch, sh, sw, feat.ld, cat2.ld), "misc", 0,
            4.0 * b * sh * sw * fch, out=cat2)
        release(feat)
        if skip is not None:
            pw(skip, "feature_projection0", out=cat2, out_off=256, relu=True)
            release(skip)
        x = sepconv(cat2, "decoder_conv0", 1, 1, True)
        release(cat2)
        y = sepconv(x, "decoder_conv1", 1, 1, True)
        release(x)
        if self.class_prediction:
            logits = pw(y, self.logits_name, pad_out=False)
            release(y)
            channels = self.classes
        else:
            logits, channels = y, y.shape[-1]         # model.py:104-106: the decoder's features are the output
        return dict(steps=steps, outs=outs, pool=pool, x_in=x_in, logits=logits, pool_bytes=pool.total_bytes,
                    out_shape=(B, sh, sw, channels))

    def plan(self, B, H, Wd, lane=0):
        """lane: independent activation pools for forward passes that run concurrently on different HIP streams."""
        key = (B, H, Wd, lane)
        if key not in self._plans:
            mult = 16 if (self.backbone == "xception" and self.OS == 16) else 8
            if H % mult or Wd % mult:
                raise ValueError(f"input size must be a multiple of {mult} for the {self.backbone} backbone (got {H}x{Wd})")
            self._plans[key] = self._build_plan(B, H, Wd)
        return self._plans[key]

    # -- execution --------------------------------------------------------------------------------
    def forward_capture(self, x_dev):
        """Debug: run the plan and return [(label, output tensor copy [rows, ld])] for every labelled step."""
        B, H, Wd, _ = x_dev.shape
        plan = self.plan(B, H, Wd)
        plan["x_in"].t.copy_(x_dev.reshape(-1))
        lib = _lib.load()
        s = _lib.stream_ptr()
        cap = []
        for (name, args, _k, _f, _b, label), out in zip(plan["steps"], plan["outs"]):
            _lib.check(getattr(lib, name)(*args, s), name)
            if out is not None:
                torch.cuda.synchronize()
                first = out[0] if isinstance(out, (list, tuple)) else out
                cap.append((label, first.t.view(B, -1).clone()))
        return cap

    # -- range guard of the split-f16 GEMMs -----------------------------------------------------------
    SPLIT_GEMMS = ("asr_pwconv_mfma_f16x3", "asr_pwconv_mfma_f16x3_presplit", "asr_conv3x3_mfma_f16x3")
    SPLIT_PRODUCERS = ("asr_dwconv3x3_nhwc_split_f16", "asr_aspp_dwconv3_nhwc_split_f16")

    def calibrate_range(self, x_dev, verbose=True):
        """Activation side of the range guard.  Runs forward passes of the probe batch ``x_dev`` [B,H,W,3] (B small) on a plan
        with the in-kernel fusions opened up (repeated until no layer moves: at most 4 passes), reads the largest magnitude of every split-f16 GEMM's A operand, and moves
        each layer whose operand is within SPLIT_HEADROOM of 2^15 (f16 overflow at 65504) or entirely below 2^-10 (fewer
        than 13 significant bits left) to the exact-f32 MFMA kernel for the lifetime of the engine.  Returns
        {layer: reason} of the layers moved by this call; the cumulative set is ``routed_f32``.  The kernels saturate
        rather than overflow, so a later input that exceeds the probe's range gives finite, less exact results; call
        this again with such an input to re-route.  No-op for precision='f32'."""
        if self.precision != "f16x3":
            return {}
        moved_all = {}
        for _pass in range(4):            # a saturated layer hides the true range of the layers behind it: repeat until stable
            moved = self._calibrate_pass(x_dev, verbose)
            if not moved:
                break
            moved_all.update(moved)
        return moved_all

    def range_report(self, x_dev):
        """On-demand check of the split-f16 range guard for a GIVEN input batch x_dev [B,H,W,3] (nothing is re-routed, the
        hot path is untouched): runs it through the opened-up plan and returns {layer: max |A operand|} of every layer still
        on a split-f16 kernel whose operand on THIS input is within SPLIT_HEADROOM of 2^15 or entirely below 2^-10 -- i.e.
        the layers that calibrate_range would move.  An empty dict means the routing decided on the probe batch also holds
        for this input; otherwise call calibrate_range(x_dev) (the kernels saturate at +-65504: an out-of-range input
        gives finite, inexact logits, never inf / NaN, and never a signal by itself)."""
        if self.precision != "f16x3":
            return {}
        return {k: v for k, v in self._calibrate_pass(x_dev, verbose=False, dry_run=True).items()}

    def _calibrate_pass(self, x_dev, verbose, dry_run=False):
        B, H, Wd, _ = x_dev.shape
        saved = self.disabled
        self.disabled = saved | {"fused_stem", "fused_sepconv"}      # their internal operands become visible tensors
        try:
            plan = self._build_plan(B, H, Wd, zero_fill=True)
        finally:
            self.disabled = saved
        plan["x_in"].t.copy_(x_dev.reshape(-1))
        lib = _lib.load()
        s = _lib.stream_ptr()
        maxima = {plan["x_in"].ptr: float(x_dev.abs().max())}
        moved = {}
        for (name, args, _k, _f, _b, label), out in zip(plan["steps"], plan["outs"]):
            layer = label.split(" ")[0] if label else ""
            if name in self.SPLIT_GEMMS and layer in self.p and layer not in self.routed_f32:
                m = maxima.get(args[0])
                if m is not None and (m != m or (m > 0.0 and (m * SPLIT_HEADROOM >= SPLIT_MAX or m < SPLIT_ACT_MIN))):
                    moved[layer] = m if dry_run else (f"max |activation| = {m:.3g} on the probe batch, outside "
                                                      f"[{SPLIT_ACT_MIN:.3g}, {SPLIT_MAX / SPLIT_HEADROOM:.3g})")
            _lib.check(getattr(lib, name)(*args, s), name)
            for buf in (out if isinstance(out, (list, tuple)) else ([out] if out is not None else [])):
                t = buf.t.view(torch.float16) if name in self.SPLIT_PRODUCERS else buf.t
                maxima[buf.ptr] = float(t.abs().max())               # (a saturated split half reads 65504: still caught)
        if moved and not dry_run:
            self.routed_f32.update(moved)
            self._w = self._host_weights
            for layer in moved:
                self._put_conv(layer, *self._conv_meta[layer])
            del self._w
            torch.cuda.synchronize(self.device)
            self._plans.clear()
            if verbose:
                import sys
                for layer, why in moved.items():
                    print(f"asr_amd: {layer} runs on the exact-f32 MFMA kernel ({why})", file=sys.stderr)
        return moved

    def input_view(self, B, H, Wd, lane=0):
        """The plan's own input buffer as a [B,H,W,3] tensor: a producer (the augmentation kernel) that writes here saves
        forward() its copy of the batch.  Valid until the next forward of the same (shape, lane)."""
        return self.plan(B, H, Wd, lane)["x_in"].t.view(B, H, Wd, 3)

    def forward(self, x_dev, profile=None, lane=0):
        """x_dev: [B,H,W,3] float32 device tensor -> logits [B,H/4,W/4,classes] (a view of plan
        memory: consume or clone it before the next forward of the same shape).
        profile: optional dict kind -> [ms, flops, bytes, launches] filled with HIP-event timings."""
        B, H, Wd, c = x_dev.shape
        assert c == 3
        plan = self.plan(B, H, Wd, lane)
        xin = plan["x_in"].t
        in_place = (x_dev.data_ptr() == xin.data_ptr() and x_dev.is_contiguous() and x_dev.numel() == xin.numel()
                    and x_dev.dtype == xin.dtype)
        if not in_place:                                 # (input_view: already in place)
            xin.copy_(x_dev.reshape(-1))
        lib = _lib.load()
        s = _lib.stream_ptr()
        if profile is None:
            for name, args, _kind, _fl, _by, _lb in plan["steps"]:
                rc = getattr(lib, name)(*args, s)
                if rc != 0:
                    _lib.check(rc, name)
        else:
            evs = []
            for name, args, kind, fl, by, label in plan["steps"]:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = getattr(lib, name)(*args, s)
                e1.record()
                if rc != 0:
                    _lib.check(rc, name)
                evs.append((kind, e0, e1, fl, by, label))
            torch.cuda.synchronize()
            detail = profile.setdefault("_detail", [])
            for kind, e0, e1, fl, by, label in evs:
                acc = profile.setdefault(kind, [0.0, 0.0, 0.0, 0])
                ms = e0.elapsed_time(e1)
                detail.append((kind, label, ms, fl, by))
                acc[0] += ms
                acc[1] += fl
                acc[2] += by
                acc[3] += 1
        return plan["logits"].t.view(plan["out_shape"])

    def flops_per_copy(self, H, Wd):
        plan = self.plan(1, H, Wd)
        out = {}
        for _n, _a, kind, fl, by, _lb in plan["steps"]:
            acc = out.setdefault(kind, [0.0, 0.0])
            acc[0] += fl
            acc[1] += by
        return out
