"""GPU parity of the DeepLabV3+ layer kernels (through the C ABI) against plain torch-CPU float32 /
float64 references of the same op.  Tolerances: f32 MFMA == k-ordered fmaf chain, so GEMM-like
outputs are compared with rtol 1e-4 / atol 1e-4 against a float64 reference (K <= 2048, O(1)
operands); memory-bound layers with atol 1e-5."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf_ops

pytestmark = pytest.mark.gpu


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("m,k,n,relu,res", [
    (1000, 728, 728, False, True),     # middle-flow shape, M tail, N not a multiple of 128
    (256, 32, 21, False, False),       # logits-like, N < 32
    (300, 304, 48, True, False),       # decoder concat K, N = 48
    (384, 2048, 256, True, False),     # ASPP
    (130, 64, 128, False, False),      # K = 2 tiles, tiny M tail
    (512, 1280, 256, True, False),
])
def test_pwconv_matches_reference(dev, m, k, n, relu, res):
    from asr_amd import ops
    rng = np.random.default_rng(m + k + n)
    x = _rand(rng, m, k)
    w = _rand(rng, k, n, scale=1.0 / np.sqrt(k))
    b = _rand(rng, n)
    r = _rand(rng, m, n) if res else None
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        ref = np.maximum(ref, 0)
    if res:
        ref = ref + r
    wp = ops.pack_pw_weights(ops.to_device(w))
    got = ops.pwconv(ops.to_device(x), wp, ops.to_device(b), k, n, relu=relu,
                     residual=ops.to_device(r) if res else None).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("m,k,n,relu,res", [
    (1000, 728, 728, False, True),
    (384, 2048, 256, True, False),
    (130, 64, 128, False, False),
    (4, 2048, 256, True, False),
    (512, 304, 256, True, False),
])
def test_pwconv_split_f16_is_f32_grade(dev, m, k, n, relu, res):
    """hi/lo f16 split on v_mfma_f32_32x32x16_f16: error must stay at f32 summation-noise level
    (<= 4e-6 of sum |x||w|), including operands spanning several orders of magnitude."""
    from asr_amd import ops
    rng = np.random.default_rng(m + k + n + 1)
    x = _rand(rng, m, k) * np.exp(rng.uniform(-6, 3, (m, k))).astype(np.float32)   # wide dynamic range
    w = _rand(rng, k, n, scale=1.0 / np.sqrt(k))
    b = _rand(rng, n)
    r = _rand(rng, m, n) if res else None
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        ref = np.maximum(ref, 0)
    if res:
        ref = ref + r
    bound = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64) + np.abs(b)
    wp = ops.pack_pw_weights_f16x3(ops.to_device(w))
    got = ops.pwconv(ops.to_device(x), wp, ops.to_device(b), k, n, relu=relu,
                     residual=ops.to_device(r) if res else None, f16x3=True).cpu().numpy()
    err = np.abs(got - ref) / bound
    assert err.max() <= 4e-6, err.max()
    # the plain f32 MFMA kernel on the same data, for scale
    got32 = ops.pwconv(ops.to_device(x), ops.pack_pw_weights(ops.to_device(w)), ops.to_device(b), k, n, relu=relu,
                       residual=ops.to_device(r) if res else None).cpu().numpy()
    assert (np.abs(got32 - ref) / bound).max() <= 4e-6


def test_pwconv_asymmetric_identity(dev):
    """A = I with an asymmetric B catches a transposed C-write or a wrong k permutation exactly."""
    from asr_amd import ops
    k = n = 160
    x = np.eye(k, dtype=np.float32)
    w = (np.arange(k * n, dtype=np.float32).reshape(k, n) % 251) - 100.0
    got = ops.pwconv(ops.to_device(x), ops.pack_pw_weights(ops.to_device(w)), None, k, n).cpu().numpy()
    assert np.array_equal(got, w)


def test_pwconv_strided_rows_and_concat_output(dev):
    """stride-2 shortcut gather (model.py:529-541) writing into a channel slice of a wider buffer."""
    from asr_amd import ops, _lib
    rng = np.random.default_rng(11)
    b, h, w_, k, n = 2, 12, 10, 64, 40
    x = _rand(rng, b, h, w_, k)
    wt = _rand(rng, k, n, scale=0.1)
    bias = _rand(rng, n)
    ho, wo = 6, 5
    ref = x[:, ::2, ::2].reshape(-1, k).astype(np.float64) @ wt.astype(np.float64) + bias
    total_c = 72
    out = torch.full((b * ho * wo, total_c), -7.0, device=dev)
    xd, wp, bd = ops.to_device(x), ops.pack_pw_weights(ops.to_device(wt)), ops.to_device(bias)
    _lib.call("asr_pwconv_mfma_f32", xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), None, out.data_ptr() + 4 * 16,
              b * ho * wo, k, n, k, total_c, 0, 0, 2, h, w_, _lib.stream_ptr())
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, 16:16 + n], ref, rtol=1e-4, atol=1e-4)
    assert np.all(got[:, :16] == -7.0) and np.all(got[:, 16 + n:] == -7.0)


@pytest.mark.parametrize("f16x3,cin,cout", [(False, 32, 64), (True, 32, 64), (True, 64, 160)])
def test_conv3x3_mfma_matches_conv2d(dev, f16x3, cin, cout):
    """entry_flow_conv1_2 as an implicit GEMM on the f32 and on the split-f16 matrix path (both f32-grade: the bound
    is the summation-order noise of a 288- / 576-term f32 dot product)."""
    from asr_amd import ops
    rng = np.random.default_rng(12)
    b, h, w_ = 2, 20, 28
    x = _rand(rng, b, h, w_, cin)
    k = _rand(rng, 3, 3, cin, cout, scale=0.08)
    bias = _rand(rng, cout)
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(k).permute(3, 2, 0, 1).double(),
                   torch.from_numpy(bias).double(), padding=1).relu().permute(0, 2, 3, 1).numpy()
    wk = ops.to_device(k.reshape(9 * cin, cout))
    wp = ops.pack_pw_weights_f16x3(wk) if f16x3 else ops.pack_pw_weights(wk)
    got = ops.conv3x3_mfma(ops.to_device(x), wp, ops.to_device(bias), cout, relu=True, f16x3=f16x3).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    mag = F.conv2d(torch.from_numpy(np.abs(x)).permute(0, 3, 1, 2).double(), torch.from_numpy(np.abs(k)).permute(3, 2, 0, 1).double(),
                   padding=1).permute(0, 2, 3, 1).numpy()
    assert np.max(np.abs(got - ref) / (mag + 1e-30)) <= 4e-6


@pytest.mark.parametrize("f16x3", [False, True])
def test_conv3x3_direct_same_padding_stride2(dev, f16x3):
    """entry_flow_conv1_1: 'same' + stride 2 on an even input pads bottom/right only; the VALU kernel and the
    split-f16 MFMA implicit GEMM (ragged width: 24 of a 32-pixel wave group)."""
    from asr_amd import ops
    rng = np.random.default_rng(13)
    b, h, w_, cin, cout = 2, 32, 48, 3, 32
    x = _rand(rng, b, h, w_, cin)
    k = _rand(rng, 3, 3, cin, cout, scale=0.3)
    bias = _rand(rng, cout)
    xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (0, 1, 0, 1))
    ref = F.conv2d(xt, torch.from_numpy(k).permute(3, 2, 0, 1), torch.from_numpy(bias), stride=2).relu()
    ref = ref.permute(0, 2, 3, 1).numpy()
    got = ops.conv3x3_direct(ops.to_device(x), ops.to_device(k), ops.to_device(bias), 2, 0, 0, (h // 2, w_ // 2),
                             relu=True, f16x3=f16x3).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    # stride 1, symmetric padding, a width that spans several wave groups with a ragged tail
    x1 = _rand(rng, 1, 9, 70, cin)
    ref1 = F.conv2d(torch.from_numpy(x1).permute(0, 3, 1, 2), torch.from_numpy(k).permute(3, 2, 0, 1), torch.from_numpy(bias),
                    padding=1).permute(0, 2, 3, 1).numpy()
    got1 = ops.conv3x3_direct(ops.to_device(x1), ops.to_device(k), ops.to_device(bias), 1, 1, 1, (9, 70), f16x3=f16x3).cpu().numpy()
    np.testing.assert_allclose(got1, ref1, rtol=1e-5, atol=1e-5)
    # impulse at the last pixel only reaches the last output pixel (asymmetric padding check)
    imp = np.zeros((1, 8, 8, 3), np.float32)
    imp[0, 7, 7, 0] = 1.0
    g = ops.conv3x3_direct(ops.to_device(imp), ops.to_device(k), ops.to_device(np.zeros(cout, np.float32)), 2, 0, 0,
                           (4, 4), f16x3=f16x3).cpu().numpy()
    assert np.count_nonzero(np.abs(g).sum(-1)) == 1 and np.allclose(g[0, 3, 3], k[1, 1, 0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("b,h,w_", [(2, 64, 96), (1, 70, 90), (3, 32, 32)])
def test_entry_stem_fused_matches_the_two_kernels(dev, b, h, w_):
    """conv1_1 + conv1_2 in one kernel (intermediate in LDS) against the stem kernel followed by the implicit-GEMM conv,
    and against torch f32: ragged 16 x 16 tiles, map borders (conv1_2's zero padding around the conv1_1 map)."""
    from asr_amd import ops
    rng = np.random.default_rng(101)
    x = _rand(rng, b, h, w_, 3)
    k1 = _rand(rng, 3, 3, 3, 32, scale=0.3)
    b1 = _rand(rng, 32, scale=0.2)
    k2 = _rand(rng, 3, 3, 32, 64, scale=(2.0 / 288) ** 0.5)
    b2 = _rand(rng, 64, scale=0.2)
    xd = ops.to_device(x)
    w2p = ops.pack_pw_weights_f16x3(ops.to_device(k2.reshape(288, 64)))
    got = ops.entry_stem_fused(xd, ops.to_device(k1), ops.to_device(b1), w2p, ops.to_device(b2)).cpu().numpy()
    mid = ops.conv3x3_direct(xd, ops.to_device(k1), ops.to_device(b1), 2, 0, 0, (h // 2, w_ // 2), relu=True, f16x3=True)
    two = ops.conv3x3_mfma(mid, w2p, ops.to_device(b2), 64, relu=True, f16x3=True).cpu().numpy()
    xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (0, 1, 0, 1))
    m = F.conv2d(xt, torch.from_numpy(k1).permute(3, 2, 0, 1), torch.from_numpy(b1), stride=2).relu()
    ref = F.conv2d(m, torch.from_numpy(k2).permute(3, 2, 0, 1), torch.from_numpy(b2), padding=1).relu().permute(0, 2, 3, 1).numpy()
    assert got.shape == ref.shape == (b, h // 2, w_ // 2, 64)
    np.testing.assert_allclose(got, two, rtol=0, atol=2e-5)
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-5)


@pytest.mark.parametrize("c,n,b,h,w_,depth_act", [(128, 128, 2, 24, 40, False), (64, 128, 2, 17, 30, False),
                                                   (128, 128, 1, 8, 14, True), (64, 128, 3, 33, 15, True)])
def test_sepconv_fused_is_bit_identical_to_the_two_kernels(dev, c, n, b, h, w_, depth_act):
    """depthwise -> LDS (split f16) -> MFMA GEMM in one kernel against asr_dwconv3x3_nhwc_f32 + asr_pwconv_mfma_f16x3:
    same depthwise arithmetic, same MFMA sequence per accumulator -> bitwise equal; ragged 8 x 14 tiles and image borders."""
    from asr_amd import ops
    rng = np.random.default_rng(303)
    x = ops.to_device(_rand(rng, b, h, w_, c))
    wd = ops.to_device(_rand(rng, 3, 3, c, scale=0.3))
    bd = ops.to_device(_rand(rng, c, scale=0.1))
    wk = ops.to_device(_rand(rng, c, n, scale=(1.0 / c) ** 0.5))
    bk = ops.to_device(_rand(rng, n, scale=0.1))
    w16 = ops.pack_pw_weights_f16x3(wk)
    ref_dw = ops.dwconv3x3(x, wd, bd, stride=1, rate=1, out_hw=(h, w_), pre_relu=not depth_act, post_relu=depth_act)
    m = b * h * w_
    ref = ops.pwconv(ref_dw.reshape(m, c), w16, bk, c, n, relu=depth_act, f16x3=True).reshape(b, h, w_, n)
    got = ops.sepconv_fused(x, wd, bd, w16, bk, n, pre_relu=not depth_act, dw_relu=depth_act, out_relu=depth_act)
    assert got.shape == ref.shape
    assert torch.equal(got, ref), float((got - ref).abs().max())


def _dw_ref(x, k, bias, stride, rate, pad, pre, post):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    if pre:
        xt = xt.relu()
    xt = F.pad(xt, (pad[2], pad[3], pad[0], pad[1]))
    c = x.shape[-1]
    y = F.conv2d(xt, torch.from_numpy(k).permute(2, 0, 1)[:, None], torch.from_numpy(bias), stride=stride,
                 dilation=rate, groups=c)
    if post:
        y = y.relu()
    return y.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("h,w_,c,stride,rate,pre,post,direct", [
    # last field = kernel mode: 0 auto (streaming where possible), 1 direct, 2 streaming
    (32, 32, 728, 1, 1, True, False, 0),         # middle flow, C not a multiple of 64 (auto = streaming)
    (40, 24, 132, 1, 1, True, False, 0),         # ragged strips and columns, C % 64 != 0
    (32, 32, 256, 1, 1, False, True, 1),         # direct kernel, same result
    (64, 64, 128, 2, 1, True, False, 1),         # stride-2 block end (explicit pad 1,1), direct
    (64, 64, 128, 2, 1, True, False, 0),         # stride-2, streaming (auto)
    (70, 50, 256, 2, 1, True, True, 2),          # stride-2 streaming, odd sizes, several strips
    (32, 32, 2048, 1, 6, False, True, 0),        # ASPP rates: direct fallback
    (32, 32, 512, 1, 12, False, True, 0),
    (32, 32, 512, 1, 18, False, True, 0),
    (9, 7, 8, 1, 1, False, False, 0),            # tiny
    (32, 32, 728, 1, 1, True, False, 2),         # streaming (register window) kernel
    (70, 40, 128, 1, 1, True, True, 2),          # streaming: several row strips, ragged columns
    (32, 32, 1536, 1, 2, False, True, 2),        # streaming, rate 2
])
def test_dwconv_matches_conv2d(dev, h, w_, c, stride, rate, pre, post, direct):
    from asr_amd import ops
    rng = np.random.default_rng(h * 7 + c + rate)
    b = 2
    x = _rand(rng, b, h, w_, c)
    k = _rand(rng, 3, 3, c, scale=0.3)
    bias = _rand(rng, c)
    if stride == 1:
        pad = (rate, rate, rate, rate)
        out_hw = (h, w_)
    else:
        pad = (1, 1, 1, 1)                       # ZeroPadding2D((1,1)), model.py:480-486
        out_hw = ((h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1)
    ref = _dw_ref(x, k, bias, stride, rate, pad, pre, post)
    got = ops.dwconv3x3(ops.to_device(x), ops.to_device(k), ops.to_device(bias), stride=stride, rate=rate,
                        pad_top=pad[0], pad_left=pad[2], out_hw=out_hw, pre_relu=pre, post_relu=post,
                        force_direct=int(direct)).cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("h,w_,c,rates", [
    (32, 32, 256, (6, 12, 18)),      # the OS-16 map of a 512 x 512 input: one column-phase group per row phase
    (16, 24, 72, (6, 12, 18)),       # ragged: phases of 2-3 rows, c not a multiple of 32
    (64, 64, 96, (6, 12, 18)),       # BASELINE configs[4] (1024 x 1024 input): 11-row phases, two column-phase groups
    (64, 64, 64, (12, 24, 36)),      # OS 8 (model.py:42-47)
    (33, 45, 40, (2, 4, 6)),         # odd sizes: the last row / column phases are one line shorter
    (20, 28, 32, (1, 2, 3)),         # period 1: the whole plane is one phase
    (8, 8, 32, (6, 12, 18)),         # rates beyond the plane: only the centre tap is ever inside
])
def test_aspp_fused_three_rates(dev, h, w_, c, rates):
    """aspp1-3 depthwise (BN folded, ReLU after, model.py:212-221) from LDS-resident phases of the plane (residue classes
    modulo gcd(rates)), any plane size."""
    from asr_amd import ops
    rng = np.random.default_rng(15)
    x = _rand(rng, 2, h, w_, c)
    k3 = _rand(rng, 3, 3, 3, c, scale=0.3)
    b3 = _rand(rng, 3, c)
    outs = ops.aspp_dwconv3(ops.to_device(x), ops.to_device(k3), ops.to_device(b3), rates=rates)
    for i, r in enumerate(rates):
        ref = _dw_ref(x, k3[i], b3[i], 1, r, (r, r, r, r), False, True)
        np.testing.assert_allclose(outs[i].cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    pre = ops.aspp_dwconv3(ops.to_device(x), ops.to_device(k3), ops.to_device(b3), rates=rates, pre_relu=True, post_relu=False)
    ref = _dw_ref(x, k3[1], b3[1], 1, rates[1], (rates[1],) * 4, True, False)
    np.testing.assert_allclose(pre[1].cpu().numpy(), ref, rtol=1e-5, atol=2e-5)


def test_gap_and_resize(dev):
    from asr_amd import ops
    rng = np.random.default_rng(14)
    x = _rand(rng, 3, 32, 32, 2048)
    got = ops.gap(ops.to_device(x)).cpu().numpy()
    np.testing.assert_allclose(got, x.astype(np.float64).mean(axis=(1, 2)), rtol=1e-5, atol=1e-6)
    y = _rand(rng, 2, 8, 8, 256)
    ref = tf_ops.resize_bilinear(torch.from_numpy(y), (32, 32)).numpy()
    got = ops.resize_bilinear(ops.to_device(y), (32, 32)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)   # 4x in width: the three-column kernel
    for hw in ((24, 24), (16, 32), (20, 12)):                   # generic kernel (3x, 2.5x / 1.5x), 4x in width with 2x in height
        ref = tf_ops.resize_bilinear(torch.from_numpy(y), hw).numpy()
        got = ops.resize_bilinear(ops.to_device(y), hw).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6)   # (non-integer ratios: the lerp weights round differently)
    wide = torch.zeros((2, 32, 32, 320), device=dev)            # into a channel slice of a concat buffer (ldy > c)
    ops.resize_bilinear(ops.to_device(y), (32, 32), out=wide, ldy=320)
    assert torch.equal(wide[..., :256], ops.resize_bilinear(ops.to_device(y), (32, 32))) and float(wide[..., 256:].abs().max()) == 0.0
    one = _rand(rng, 2, 1, 1, 256)               # image-pooling broadcast (model.py:204-205)
    got = ops.resize_bilinear(ops.to_device(one), (16, 16)).cpu().numpy()
    assert np.array_equal(got, np.broadcast_to(one, (2, 16, 16, 256)))


@pytest.mark.parametrize("c,n,hw,stride,rate", [(728, 728, 32, 1, 1), (256, 256, 32, 2, 1), (96, 256, 16, 1, 2), (1536, 2048, 16, 1, 2)])
def test_presplit_sepconv_matches_the_f32_handoff(dev, c, n, hw, stride, rate):
    """dw -> split-f16 chunks -> LDS-DMA GEMM (256 x 256 tile) against dw -> f32 -> split-f16 GEMM: the same hi / lo halves
    enter the same three products.  The 256 x 256 kernel's v_mfma_f32_16x16x32_f16 sums each 32-deep K-step in one MFMA (the
    in-kernel-split kernel uses two v_mfma_f32_32x32x16_f16), so the two differ by f32 summation-order noise only: both must
    sit within 4e-6 * sum |x||w| of the float64 product (the bound of test_pwconv_split_f16_is_f32_grade).  Tails: M, K and N
    not multiples of the tile (N = 728: the last N-tile's padded column tiles are skipped)."""
    from asr_amd import ops
    rng = np.random.default_rng(61)
    b = 3
    x = ops.to_device(_rand(rng, b, hw, hw, c))
    wd = ops.to_device(_rand(rng, 3, 3, c, scale=0.3))
    bd = ops.to_device(_rand(rng, c, scale=0.1))
    wk_h = _rand(rng, c, n, scale=(1.0 / c) ** 0.5)
    wk = ops.to_device(wk_h)
    bk_h = _rand(rng, n, scale=0.1)
    bk = ops.to_device(bk_h)
    w16 = ops.pack_pw_weights_f16x3(wk)
    out_hw = (hw, hw) if stride == 1 else ((hw + 2 * rate - (2 * rate + 1)) // stride + 1,) * 2
    ref_dw = ops.dwconv3x3(x, wd, bd, stride=stride, rate=rate, out_hw=out_hw, pre_relu=True, post_relu=False)
    m = ref_dw.shape[0] * ref_dw.shape[1] * ref_dw.shape[2]
    res_h = _rand(rng, m, n)
    res = ops.to_device(res_h)
    ref = ops.pwconv(ref_dw.reshape(m, c), w16, bk, c, n, residual=res, relu=True, f16x3=True)
    xs, (bb, ho, wo), chunks = ops.dwconv3x3_split(x, wd, bd, stride=stride, rate=rate, pre_relu=True, post_relu=0)
    assert bb * ho * wo == m and chunks == (c + 31) // 32
    got = ops.pwconv_presplit(xs, w16, bk, c, n, chunks, residual=res, relu=1)
    a64 = ref_dw.reshape(m, c).cpu().numpy().astype(np.float64)
    exact = np.maximum(a64 @ wk_h.astype(np.float64) + bk_h, 0) + res_h
    bound = np.abs(a64) @ np.abs(wk_h).astype(np.float64) + np.abs(bk_h) + np.abs(res_h)
    for out in (got, ref):
        assert (np.abs(out.cpu().numpy() - exact) / bound).max() <= 4e-6
    assert float((got - ref).abs().max()) <= 4e-6 * float(bound.max())
    # the split buffer itself: hi + lo reproduces the f32 depthwise output to 2^-22, padding channels are zero
    halves = xs.view(torch.float16).reshape(m, chunks, 2, 32).float()
    rec = (halves[:, :, 0, :] + halves[:, :, 1, :]).reshape(m, chunks * 32)
    np.testing.assert_allclose(rec[:, :c].cpu().numpy(), ref_dw.reshape(m, c).cpu().numpy(), rtol=3e-7, atol=1e-7)
    assert float(rec[:, c:].abs().max()) == 0.0 if chunks * 32 > c else True


@pytest.mark.parametrize("k,n,relu", [(728, 728, 0), (256, 256, 1), (128, 256, 2), (1024, 1536, 1)])
def test_presplit_gemm_persistent_walk_is_bit_identical_to_one_tile_per_workgroup(dev, k, n, relu):
    """Rows are independent: one launch over all rows must equal, bit for bit, launches over row blocks of <= 256 tiles --
    ragged M, the padded last N-tile of 728, every activation mode -- and sit within the f32-grade bound of the float64
    product.  The whole launch runs the persistent walk of the ring GEMM (pw_gemm_f16x3_pre_ring_persist_kernel: more tiles
    than CUs, no residual -- the product path since round 4), the row blocks run pw_gemm_f16x3_pre_ring_kernel, one tile
    per workgroup: DESIGN.md 4.1.)"""
    from asr_amd import ops
    rng = np.random.default_rng(77)
    m = 70000 + 37                                             # 274 row tiles, the last one ragged
    chunks = (k + 31) // 32
    a_h = _rand(rng, m, k)
    hi = a_h.astype(np.float16)
    lo = (a_h - hi.astype(np.float32)).astype(np.float16)
    lines = np.zeros((m, chunks, 2, 32), np.float16)
    pad = chunks * 32 - k
    lines[:, :, 0, :] = np.pad(hi, ((0, 0), (0, pad))).reshape(m, chunks, 32)
    lines[:, :, 1, :] = np.pad(lo, ((0, 0), (0, pad))).reshape(m, chunks, 32)
    xs = torch.from_numpy(lines.reshape(m, chunks * 64)).to(dev).view(torch.float32).reshape(m, chunks, 32)
    wk_h = _rand(rng, k, n, scale=(1.0 / k) ** 0.5)
    bk_h = _rand(rng, n, scale=0.1)
    w16 = ops.pack_pw_weights_f16x3(ops.to_device(wk_h))
    bk = ops.to_device(bk_h)
    tiles_n = -(-n // 256)
    assert -(-m // 256) * tiles_n > 256                        # the whole launch: more tiles than CUs
    whole = ops.pwconv_presplit(xs, w16, bk, k, n, chunks, relu=relu)
    rows = (256 // tiles_n) * 256                              # row blocks of <= 256 tiles: one tile per workgroup
    parts = [ops.pwconv_presplit(xs[r0:r0 + rows].contiguous(), w16, bk, k, n, chunks, relu=relu) for r0 in range(0, m, rows)]
    assert torch.equal(whole.view(torch.int32), torch.cat(parts).view(torch.int32))
    a64 = (hi.astype(np.float64) + lo.astype(np.float64))
    exact = a64 @ wk_h.astype(np.float64) + bk_h
    if relu:
        exact = np.maximum(exact, 0)
    if relu == 2:
        exact = np.minimum(exact, 6)
    bound = np.abs(a64) @ np.abs(wk_h).astype(np.float64) + np.abs(bk_h)
    assert (np.abs(whole.cpu().numpy()[:, :n] - exact) / bound).max() <= 4e-6


@pytest.mark.parametrize("hw", [32, 64])
def test_aspp_split_operands_match_the_f32_outputs(dev, hw):
    """asr_aspp_dwconv3_nhwc_split_f16 at the product shapes (32 x 32 x 2048 for 512 x 512 inputs, 64 x 64 x 2048 for the
    1024 x 1024 inputs of BASELINE configs[4]; rates 6 / 12 / 18, model.py:212-221): the split-f16 chunks it hands to the
    three pointwise GEMMs reproduce the f32 outputs of asr_aspp_dwconv3_nhwc_f32 to 2^-22, and the GEMM on them stays
    f32-grade against the float64 product."""
    from asr_amd import ops
    rng = np.random.default_rng(77)
    b, c, n = 2, 2048, 256
    x = ops.to_device(_rand(rng, b, hw, hw, c))
    w3 = ops.to_device(_rand(rng, 3, 3, 3, c, scale=0.3))
    b3 = ops.to_device(_rand(rng, 3, c, scale=0.1))
    ref = ops.aspp_dwconv3(x, w3, b3, rates=(6, 12, 18), pre_relu=False, post_relu=True)
    got = ops.aspp_dwconv3_split(x, w3, b3, rates=(6, 12, 18), pre_relu=False, post_relu=True)
    wk_h = _rand(rng, c, n, scale=(1.0 / c) ** 0.5)
    w16 = ops.pack_pw_weights_f16x3(ops.to_device(wk_h))
    bk_h = _rand(rng, n, scale=0.1)
    for r, g in zip(ref, got):
        m = b * hw * hw
        halves = g.view(torch.float16).reshape(m, c // 32, 2, 32).float()
        rec = (halves[:, :, 0, :] + halves[:, :, 1, :]).reshape(m, c)
        np.testing.assert_allclose(rec.cpu().numpy(), r.reshape(m, c).cpu().numpy(), rtol=3e-7, atol=1e-7)
        out = ops.pwconv_presplit(g, w16, ops.to_device(bk_h), c, n, c // 32, relu=1).cpu().numpy()
        a64 = r.reshape(m, c).cpu().numpy().astype(np.float64)
        exact = np.maximum(a64 @ wk_h.astype(np.float64) + bk_h, 0)
        bound = np.abs(a64) @ np.abs(wk_h).astype(np.float64) + np.abs(bk_h)
        assert (np.abs(out - exact) / bound).max() <= 4e-6


def test_split_f16_operands_saturate_and_document_their_range(dev):
    """Range behaviour of the split-f16 operands (hi = f16(v), lo = f16(v - hi)), both producers (the in-kernel split of
    asr_pwconv_mfma_f16x3 and the depthwise hand-off of asr_dwconv3x3_nhwc_split_f16):
      * |v| up to ~5e4: f32-grade (the 4e-6 * sum |x||w| bound);
      * |v| beyond f16's 65504 (here up to 3e5): the halves saturate -- every output stays FINITE, rows that hold no
        out-of-range value keep the f32-grade bound, the others are inexact (the engine's calibrate_range moves such a
        layer to the exact-f32 kernel, test_gpu_model.py::test_range_guard_routes_layers_to_the_exact_f32_kernels);
      * tiny operands (|v| ~ 1e-8): the absolute error is bounded by f16's subnormal spacing, 2^-24 * sum |w|."""
    from asr_amd import ops
    rng = np.random.default_rng(5)
    m, k, n = 512, 256, 256
    wk_h = _rand(rng, k, n, scale=(1.0 / k) ** 0.5)
    wk = ops.to_device(wk_h)
    w16 = ops.pack_pw_weights_f16x3(wk)
    w64 = wk_h.astype(np.float64)

    def both_paths(x_h):
        """x_h [m, k] -> outputs of the in-kernel split GEMM and of the depthwise (identity tap) hand-off + LDS-DMA GEMM"""
        xd = ops.to_device(x_h)
        a = ops.pwconv(xd, w16, None, k, n, f16x3=True).cpu().numpy()
        wd = torch.zeros(3, 3, k, device=xd.device)
        wd[1, 1] = 1.0
        xs, _, chunks = ops.dwconv3x3_split(xd.reshape(2, 16, 16, k), wd, torch.zeros(k, device=xd.device))
        b = ops.pwconv_presplit(xs, w16, None, k, n, chunks).cpu().numpy()
        return a, b

    # in range: row scales over four orders of magnitude, the largest values just below f16's 65504
    x = (np.clip(_rand(rng, m, k), -4.5, 4.5) * np.exp(rng.uniform(np.log(1.0), np.log(1.2e4), (m, 1)))).astype(np.float32)
    assert 3e4 < np.abs(x).max() < 6e4
    bound = np.abs(x).astype(np.float64) @ np.abs(w64)
    for out in both_paths(x):
        assert (np.abs(out - x.astype(np.float64) @ w64) / bound).max() <= 4e-6
    # out of range in rows 0..63 only
    x2 = x.copy()
    x2[:64, ::7] = (rng.choice([-1.0, 1.0], (64, x2[:, ::7].shape[1])) * rng.uniform(7e4, 3e5, (64, x2[:, ::7].shape[1]))).astype(np.float32)
    bound2 = np.abs(x2).astype(np.float64) @ np.abs(w64)
    for out in both_paths(x2):
        assert np.isfinite(out).all()
        assert (np.abs(out[64:] - x2[64:].astype(np.float64) @ w64) / bound2[64:]).max() <= 4e-6
        assert (np.abs(out[:64] - x2[:64].astype(np.float64) @ w64) / bound2[:64]).max() <= 1.0     # saturated, not garbage
    # tiny operands
    x3 = (_rand(rng, m, k) * 1e-8).astype(np.float32)
    abs_bound = 2.0 ** -24 * np.abs(w64).sum(axis=0)
    for out in both_paths(x3):
        assert (np.abs(out - x3.astype(np.float64) @ w64) <= abs_bound[None, :] + 1e-12).all()
