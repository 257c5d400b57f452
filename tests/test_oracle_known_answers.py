"""CPU tests that pin the oracle (SURVEY 8c known answers).  The reference ships no tests or golden
vectors for this path and TF/TFA are not installable offline, so these answers are derived from the
documented op semantics, not from reference outputs ("parity unpinned")."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import augment as o_aug
from oracle import sr as o_sr
from oracle import tf_ops


def test_numpy_legacy_rng_stream_seed_1234():
    """MT19937 legacy stream, angles first then shifts, entry 0 zeroed."""
    np.random.seed(1234)
    a, s = o_aug.draw_angles_shifts(8, 0.15, 80)
    np.testing.assert_allclose(a[1:4], [0.03663263, -0.01868168, 0.08560757], rtol=1e-6)
    np.testing.assert_allclose(s[1], [-22.749237, 0.15922008], rtol=1e-6)
    assert a[0] == 0 and np.all(s[0] == 0) and a.dtype == np.float32 and s.dtype == np.float32
    np.random.seed(1234)
    a, s = o_aug.draw_angles_shifts(100, 0.15, 80)
    np.testing.assert_allclose(a[1:4], [0.03663263, -0.01868168, 0.08560757], rtol=1e-6)
    np.testing.assert_allclose(s[1], [47.49875, 9.2417326], rtol=1e-6)


def test_identity_copy_is_bit_exact_and_integer_shift_moves_pixels():
    x = torch.rand(2, 16, 20, 3)
    assert torch.equal(tf_ops.translate(tf_ops.rotate(x, [0.0, 0.0]), [[0, 0], [0, 0]]), x)
    y = tf_ops.translate(x, [[3, -2], [0, 0]])[0]
    assert torch.equal(y[:-2, 3:], x[0, 2:, :-3])           # dx=+3 right, dy=-2 up
    assert torch.all(y[-2:] == 0) and torch.all(y[:, :3] == 0)   # zero fill


def test_rotation_by_quarter_turn_permutes_pixels_about_centre():
    x = torch.rand(1, 9, 9, 1)
    y = tf_ops.rotate(x, [np.pi / 2])[0, :, :, 0]
    # out[y,x] = in[x', y'] with x' = -y + 8, y' = x (transform [c,-s,xo,s,c,yo], c~0, s=1)
    exp = torch.flip(x[0, :, :, 0].T, dims=[1])
    alt = torch.flip(x[0, :, :, 0].T, dims=[0])
    d = min((y - exp).abs().max().item(), (y - alt).abs().max().item())
    assert d < 1e-5


def test_rotate_equals_grid_sample_align_corners():
    img = torch.rand(3, 20, 24, 2)
    ang = np.array([0.3, -0.2, 0.1], np.float32)
    got = tf_ops.rotate(img, ang)
    tr = torch.as_tensor(tf_ops.angles_to_projective_transforms(ang, 20, 24))
    ys, xs = torch.meshgrid(torch.arange(20.), torch.arange(24.), indexing="ij")
    ix = tr[:, 0, None, None] * xs + tr[:, 1, None, None] * ys + tr[:, 2, None, None]
    iy = tr[:, 3, None, None] * xs + tr[:, 4, None, None] * ys + tr[:, 5, None, None]
    grid = torch.stack([ix / 23 * 2 - 1, iy / 19 * 2 - 1], -1)
    ref = F.grid_sample(img.permute(0, 3, 1, 2), grid, mode="bilinear", padding_mode="zeros",
                        align_corners=True).permute(0, 2, 3, 1)
    assert (got - ref).abs().max() < 1e-5


def test_downsample_operator_is_central_2x2_mean_and_matches_interpolate():
    x = torch.rand(1, 32, 32, 1)
    d = tf_ops.resize_bilinear(x, (8, 8))
    box = x.reshape(1, 8, 4, 8, 4, 1)[:, :, 1:3, :, 1:3].mean(dim=(2, 4))
    assert (d - box).abs().max() < 2e-7
    ref = F.interpolate(x.permute(0, 3, 1, 2), size=(8, 8), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert (d - ref).abs().max() < 2e-7
    assert torch.all(tf_ops.resize_bilinear(torch.full((1, 16, 16, 1), 0.37), (4, 4)) == np.float32(0.37))
    ramp = torch.arange(16, dtype=torch.float32).reshape(1, 1, 16, 1).expand(1, 16, 16, 1)
    np.testing.assert_allclose(tf_ops.resize_bilinear(ramp, (4, 4))[0, 0, :, 0], [1.5, 5.5, 9.5, 13.5])
    # 2x: 2x2 box mean; upsample: edge clamp
    d2 = tf_ops.resize_bilinear(x, (16, 16))
    assert (d2 - x.reshape(1, 16, 2, 16, 2, 1).mean(dim=(2, 4))).abs().max() < 2e-7
    up = tf_ops.resize_bilinear(d, (32, 32))
    ref = F.interpolate(d.permute(0, 3, 1, 2), size=(32, 32), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert (up - ref).abs().max() < 3e-7


def test_resize_gradient_is_the_exact_adjoint():
    x = torch.rand(2, 32, 32, 1)
    g = torch.rand(2, 8, 8, 1)
    lhs = (tf_ops.resize_bilinear_grad(g, (32, 32)) * x).sum()
    rhs = (g * tf_ops.resize_bilinear(x, (8, 8))).sum()
    assert abs(lhs - rhs) < 1e-4 * abs(rhs)


def test_warp_gradient_is_inverse_warp_not_adjoint():
    g = torch.rand(1, 16, 16, 1)
    ident = tf_ops.angles_to_projective_transforms([0.0], 16, 16)
    assert torch.equal(tf_ops.projective_transform_grad(g, ident, (16, 16)), g)
    # translate by +3: TF's gradient is the translate by -3 of the upstream gradient
    tr = tf_ops.translations_to_projective_transforms([[3.0, 0.0]])
    got = tf_ops.projective_transform_grad(g, tr, (16, 16))
    assert torch.equal(got, tf_ops.translate(g, [[-3.0, 0.0]]))
    inv = tf_ops.invert_transforms(tf_ops.angles_to_projective_transforms([0.37], 16, 16))
    back = tf_ops.angles_to_projective_transforms([-0.37], 16, 16)
    np.testing.assert_allclose(inv, back, atol=2e-6)


def test_adam_first_step_is_minus_lr_sign_and_counter_persists():
    adam = o_sr.KerasAdam(1e-3, amsgrad=True)
    var = torch.zeros(1, 4, 4, 1)
    g = torch.tensor([-2.0, 3.0, 0.5, -0.25]).reshape(1, 4, 1, 1).expand(1, 4, 4, 1).clone()
    slots = adam.new_slots(var)
    adam.apply(var, g, slots)
    np.testing.assert_allclose(var.numpy(), -1e-3 * np.sign(g.numpy()), rtol=1e-4)
    assert adam.iterations == 1
    # a fresh variable keeps the global step: bias correction uses t = 2
    a2 = adam.alpha()
    assert abs(a2 - 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)) < 1e-8
    assert abs(o_sr.exponential_decay(1e-3, 60, 0.3, 60) - 3e-4) < 1e-8


def test_sr_gradient_matches_finite_differences():
    """Data term excluded (TF's warp gradient is not the true derivative): priors only + the TF rule."""
    sr = o_sr.Superresolution(0.0, 0.3, 0.7, 0.05, num_aug=1, feature_size=(4, 4), output_size=(8, 8))
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((1, 8, 8, 1)).astype(np.float32))
    y = torch.zeros(1, 4, 4, 1)
    a, s = np.zeros(1, np.float32), np.zeros((1, 2), np.float32)
    _, g = sr.loss_and_grad(x, y, a, s)
    eps = 1e-3
    for (i, j) in [(0, 0), (3, 4), (7, 7), (7, 2)]:
        xp, xm = x.clone(), x.clone()
        xp[0, i, j, 0] += eps
        xm[0, i, j, 0] -= eps
        fd = (sr.loss_function(xp, y, a, s) - sr.loss_function(xm, y, a, s)) / (2 * eps)
        assert abs(fd - g[0, i, j, 0].item()) < 2e-2, (i, j, fd, g[0, i, j, 0].item())
    # identity warp: TF-rule data gradient == exact adjoint == 2 * D^T (D x - y)
    sr2 = o_sr.Superresolution(1.0, 0.0, 0.0, 0.0, num_aug=1, feature_size=(4, 4), output_size=(8, 8))
    _, g2 = sr2.loss_and_grad(x, y, a, s)
    ref = tf_ops.resize_bilinear_grad(2.0 * tf_ops.resize_bilinear(x, (4, 4)), (8, 8))
    assert (g2 - ref).abs().max() < 1e-6


def test_threshold_minmax_and_iou_with_void_pixels():
    img = np.array([[0.0, 0.1, 0.5], [1.0, 0.21, 0.19]], np.float32)
    assert np.array_equal(o_sr.threshold_image(img, 8, th_factor=0.2), [[0, 0, 8], [8, 8, 0]])
    assert np.array_equal(o_sr.threshold_image(img, 8, th_mask=np.full_like(img, 0.5)), [[0, 0, 8], [8, 0, 0]])
    np.testing.assert_allclose(o_sr.min_max_normalization(np.array([2.0, 4.0, 6.0]), 0.0, 1.0), [0, 0.5, 1])
    assert np.all(o_sr.min_max_normalization(np.array([3.0, 3.0]), 0.0, 1.0) == 0)
    t = np.array([8, 8, 0, 255, 255, 3])
    p = np.array([8, 0, 0, 8, 0, 0])
    assert o_aug.single_class_IOU(t, p, 8, False) == 1 / 3           # inter 1, union 3 (void NOT excluded)
    # with bg: truth -> [8,8,0,0,0,0]; class 8: 1/3 ; class 0: inter 3, union 5
    assert abs(o_aug.single_class_IOU(t, p, 8, True) - (1 / 3 + 3 / 5) / 2) < 1e-12
    assert np.isnan(o_aug.single_class_IOU(np.zeros(4), np.zeros(4), 8, False))


def test_opm_modes():
    rng = np.random.default_rng(1)
    pred = rng.standard_normal((2, 5, 5, 21)).astype(np.float32)
    pred[0, 0, 0] = 0
    pred[0, 0, 0, [3, 8]] = 2.0                                         # tie: first maximum (3) wins
    cm, mm = o_aug.opm(pred, 8, "argmax")
    assert cm[0].shape == (5, 5, 1) and cm[0][0, 0, 0] == 0 and mm == []
    assert set(np.unique(np.stack(cm))) <= {0.0, 8.0}
    cm, mm = o_aug.opm(pred, 8, "slice_max")
    assert np.array_equal(cm[1][..., 0], pred[1, ..., 8])
    assert np.array_equal(mm[1][..., 0], np.delete(pred[1], 8, axis=-1).max(-1))
    cm, _ = o_aug.opm(pred, 8, "slice")
    exp = (pred[1, ..., 8] - pred[1].min()) / (pred[1].max() - pred[1].min())
    np.testing.assert_allclose(cm[1][..., 0], exp, atol=1e-7)


def test_same_padding_asymmetry_of_stride2_stem():
    """Keras 'same' with stride 2 on an even input pads bottom/right only."""
    from oracle.model import OracleDeeplabV3Plus, _same_pad
    assert _same_pad(512, 3, 2) == (0, 1) and _same_pad(512, 3, 1) == (1, 1) and _same_pad(32, 37, 1) == (18, 18)
    k = np.zeros((3, 3, 1, 1), np.float32)
    k[0, 0] = 1.0                                                      # picks the top-left tap
    m = OracleDeeplabV3Plus({"c/kernel": k})
    x = torch.arange(64, dtype=torch.float32).reshape(1, 1, 8, 8)
    y = m.conv(x, "c", stride=2)
    assert y.shape == (1, 1, 4, 4) and torch.equal(y[0, 0], x[0, 0, ::2, ::2])   # no top/left padding


def test_bn_folding_equals_unfolded_sepconv():
    """Host logic of the product (weights.fold_*) against the oracle's unfused layers."""
    from asr_amd import weights as W
    from oracle.model import OracleDeeplabV3Plus
    rng = np.random.default_rng(2)
    c, co = 8, 12
    w = {"p_depthwise/depthwise_kernel": rng.standard_normal((3, 3, c, 1)).astype(np.float32),
         "p_pointwise/kernel": rng.standard_normal((1, 1, c, co)).astype(np.float32)}
    for n_, ch in (("p_depthwise_BN", c), ("p_pointwise_BN", co)):
        w[n_ + "/gamma"] = rng.uniform(0.5, 1.5, ch).astype(np.float32)
        w[n_ + "/beta"] = rng.standard_normal(ch).astype(np.float32)
        w[n_ + "/moving_mean"] = rng.standard_normal(ch).astype(np.float32)
        w[n_ + "/moving_variance"] = rng.uniform(0.5, 1.5, ch).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((1, c, 6, 6)).astype(np.float32))
    ref = OracleDeeplabV3Plus(w).sepconv_bn(x, "p", depth_activation=True, eps=1e-3)
    kd, bd = W.fold_dw_bn(w, "p_depthwise", "p_depthwise_BN", 1e-3)
    kp, bp = W.fold_conv_bn(w, "p_pointwise", "p_pointwise_BN", 1e-3)
    t = F.conv2d(x, torch.from_numpy(kd).permute(2, 0, 1)[:, None], torch.from_numpy(bd), padding=1, groups=c).relu()
    got = (torch.einsum("bchw,cn->bnhw", t, torch.from_numpy(kp)) + torch.from_numpy(bp).reshape(1, -1, 1, 1)).relu()
    assert (got - ref).abs().max() < 1e-4


def test_oracle_model_runs_and_param_inventory():
    from asr_amd import weights as W
    from oracle.model import OracleDeeplabV3Plus
    w = W.make_synthetic_weights(1234)
    out, st = OracleDeeplabV3Plus(w).forward(np.random.default_rng(3).random((1, 64, 64, 3), dtype=np.float32),
                                             return_stages=True)
    assert out.shape == (1, 16, 16, 21) and st["entry"].shape == (1, 4, 4, 728) and st["aspp"].shape == (1, 4, 4, 256)
    assert np.isfinite(out).all() and 0.05 < out.std() < 5
    n_conv = sum(1 for k, *_ in W.layer_inventory() if k in ("conv", "dw"))
    assert n_conv == 2 + 21 * 6 + 4 + 9 + 5 + 1 == 147                  # 147 conv layers


def test_other_keras_optimizers_first_steps():
    """optimizer.py:21-35 -- hand-computed steps of the TF 2.7 dense kernels (SGD / momentum / Nesterov,
    AdagradV2, Adadelta, AdaMax) on var = 1, grad = 0.5."""
    def run(kind, steps, **kw):
        opt = o_sr.Optimizer(kind, 0.1, **kw).optimizer
        var, grad = torch.ones(1, 2, 2, 1), torch.full((1, 2, 2, 1), 0.5)
        slots = opt.new_slots(var)
        for _ in range(steps):
            opt.apply(var, grad, slots)
        assert opt.iterations == steps
        return float(var[0, 0, 0, 0])
    assert abs(run("sgd", 1) - 0.95) < 1e-7
    assert abs(run("sgd", 2, momentum=0.9) - 0.855) < 1e-6           # accum: -0.05, -0.095
    assert abs(run("sgd", 1, momentum=0.9, nesterov=True) - 0.905) < 1e-6
    assert abs(run("adagrad", 1) - (1 - 0.05 / (np.sqrt(0.35) + 1e-7))) < 1e-6    # accum 0.1 + 0.25
    upd = np.sqrt(1e-7) / np.sqrt(0.05 * 0.25 + 1e-7) * 0.5
    assert abs(run("adadelta", 1) - (1 - 0.1 * upd)) < 1e-7
    assert abs(run("adamax", 1) - 0.9) < 1e-6                         # lr/(1-b1) * (0.05 / 0.5)
    # any other name falls through to Adam (optimizer.py:36-41)
    assert isinstance(o_sr.Optimizer("rmsprop").optimizer, o_sr.KerasAdam)


def test_bilateral_tv_value_and_gradient():
    """superresolution.py:8-23: 15 (h, v) pairs weighted 0.6^(|h|+|v|); gradient vs autograd of the same sum."""
    assert len(o_sr.btv_pairs()) == 15 and o_sr.btv_pairs()[0] == (-2, 0) and o_sr.btv_pairs()[-1] == (2, 2)
    const = torch.ones(1, 8, 8, 1)
    # a constant image only differs from its zero-filled translates on the uncovered border strips
    exp = sum(float(o_sr.btv_weight(0.6, h, v)) * (64 - (8 - abs(h)) * (8 - v)) for h, v in o_sr.btv_pairs())
    assert abs(o_sr.bilateral_tv(const) - exp) < 1e-4
    torch.manual_seed(0)
    x = torch.rand(1, 7, 6, 1)
    xd = x.double().requires_grad_(True)
    tot = 0
    for h, v in o_sr.btv_pairs():
        sh = torch.zeros_like(xd)
        sh[0, max(v, 0):, max(h, 0):7 if h >= 0 else 6 + h, 0] = xd[0, :7 - v, max(-h, 0):6 - max(h, 0), 0]
        tot = tot + float(o_sr.btv_weight(0.6, h, v)) * (xd - sh).abs().sum()
    tot.backward()
    g = o_sr.bilateral_tv_grad(x, 0.3)
    np.testing.assert_allclose(g.numpy(), 0.3 * xd.grad.numpy(), rtol=0, atol=2e-6)


def test_copy_dropout_mask_is_drawn_once():
    """superresolution.py:47-53 runs np.random.shuffle inside a @tf.function: once per object (trace time)."""
    sr = o_sr.Superresolution(1, 0, 0, 0, num_aug=10, copy_dropout=0.3)
    np.random.seed(5)
    exp = np.full(10, True)
    exp[:3] = False
    np.random.shuffle(exp)
    np.random.seed(5)
    m1 = sr.drop_mask(3).copy()
    state = np.random.get_state()[1].copy()
    m2 = sr.drop_mask(3)
    assert np.array_equal(m1, exp) and np.array_equal(m1, m2) and m1.sum() == 7
    assert np.array_equal(state, np.random.get_state()[1])       # no further draws


def test_nearest_warp_and_mean_iou_known_answers():
    """check_robustness.py:45-51 / utils.py:151-177: NEAREST reads I(round(y'), round(x')); Mean_IOU averages over the
    labels present in the ground truth, void removed."""
    x = torch.arange(30, dtype=torch.float32).reshape(1, 6, 5, 1)
    y = tf_ops.translate(x, [[1, 2]], interpolation="nearest")[0, :, :, 0]
    assert torch.equal(y[2:, 1:], x[0, :-2, :-1, 0]) and torch.all(y[:2] == 0) and torch.all(y[:, 0] == 0)
    half = tf_ops.translate(x, [[0.5, 0.0]], interpolation="nearest")[0, :, :, 0]        # in_x = x - 0.5: std::round, half away
    assert torch.equal(half[:, 1:], x[0, :, 1:, 0]) and torch.all(half[:, 0] == 0)       # 0.5 -> 1, 1.5 -> 2, -0.5 -> -1 (outside)
    r = tf_ops.rotate(x, [0.4], interpolation="nearest")
    assert set(np.unique(r.numpy())) <= set(np.unique(x.numpy())) | {0.0}               # labels are never blended
    assert o_aug.Mean_IOU(np.array([0, 0, 8, 8, 255]), np.array([0, 8, 8, 8, 0])) == 0.5  # (1/3 + 2/3) / 2
    assert np.isnan(o_aug.Mean_IOU(np.array([255, 255]), np.array([0, 0])))
