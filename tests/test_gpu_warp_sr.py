"""GPU parity: warp / augment / SR / realign / threshold / IoU kernels (through the C ABI) against
the CPU oracle on identical seeded inputs.  Float tolerances are stated per test; integer and
index outputs must be bit-exact."""
import numpy as np
import pytest
import torch

from oracle import augment as o_aug
from oracle import sr as o_sr
from oracle import tf_ops

pytestmark = pytest.mark.gpu


def _blob_masks(rng, n, h, w, value=1.0):
    """Piecewise-constant LR 'class masks' with structure (like argmax OPM output)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.zeros((n, h, w), np.float32)
    for i in range(n):
        cy, cx = h * (0.35 + 0.3 * rng.random()), w * (0.35 + 0.3 * rng.random())
        ry, rx = h * (0.15 + 0.15 * rng.random()), w * (0.15 + 0.2 * rng.random())
        out[i] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0) * value
    return out


def _angles_shifts(rng, n, angle_max, shift_max):
    a = rng.uniform(-angle_max, angle_max, n)
    s = rng.uniform(-shift_max, shift_max, (n, 2))
    a[0] = 0
    s[0] = 0
    return a.astype(np.float32), s.astype(np.float32)


def test_warp_affine_matches_oracle(dev):
    from asr_amd import ops, transforms as T
    rng = np.random.default_rng(0)
    img = rng.random((5, 40, 56, 3), dtype=np.float32)
    ang = np.array([0.0, 0.3, -0.7, 1.2, -0.05], np.float32)
    tf = T.rotation_transforms(ang, 40, 56)
    tf[3, 6:] = [1e-3, -2e-3]                      # one genuinely projective transform
    tf[4] = T.translation_transforms(np.array([[7.25, -3.5]], np.float32))[0]
    ref = tf_ops.projective_transform(torch.from_numpy(img), tf).numpy()
    got = ops.warp_affine(ops.to_device(img), ops.to_device(tf)).cpu().numpy()
    # same op order, no FMA contraction on either side: expected bit-identical; tolerance 1e-6
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    # shared source / output-shape variant
    ref2 = tf_ops.projective_transform(torch.from_numpy(np.broadcast_to(img[:1], img.shape).copy()), tf,
                                       output_shape=(24, 31)).numpy()
    got2 = ops.warp_affine(ops.to_device(img[0]), ops.to_device(tf), out_hw=(24, 31)).cpu().numpy()
    np.testing.assert_allclose(got2, ref2, rtol=0, atol=1e-6)


def test_integer_shift_and_identity_exact(dev):
    from asr_amd import ops, transforms as T
    rng = np.random.default_rng(1)
    img = rng.random((32, 48, 3), dtype=np.float32)
    rot = T.rotation_transforms(np.zeros(2, np.float32), 32, 48)
    tr = T.translation_transforms(np.array([[0, 0], [5, -3]], np.float32))
    out = ops.augment_copies(ops.to_device(img), ops.to_device(rot), ops.to_device(tr)).cpu().numpy()
    assert np.array_equal(out[0], img)                                  # copy 0 == input, bit for bit
    exp = np.zeros_like(img)
    exp[:-3, 5:] = img[3:, :-5]                                         # dx=+5 moves right, dy=-3 moves up
    assert np.array_equal(out[1], exp)


@pytest.mark.parametrize("n,h,w,c", [(6, 64, 64, 3), (3, 48, 80, 1)])
def test_augment_copies_matches_oracle(dev, n, h, w, c):
    from asr_amd import ops, transforms as T
    rng = np.random.default_rng(2)
    img = rng.random((h, w, c), dtype=np.float32)
    ang, sh = _angles_shifts(rng, n, 0.4, 12.0)
    tiled = torch.from_numpy(np.broadcast_to(img[None], (n, h, w, c)).copy())
    ref = tf_ops.translate(tf_ops.rotate(tiled, ang), sh).numpy()
    got = ops.augment_copies(ops.to_device(img), ops.to_device(T.rotation_transforms(ang, h, w)),
                             ops.to_device(T.translation_transforms(sh))).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)


def _sr_problem(seed, b, n, H, h, angle_max=0.15, shift_frac=0.15):
    rng = np.random.default_rng(seed)
    ys, angs, shs = [], [], []
    for _ in range(b):
        ys.append(_blob_masks(rng, n, h, h))
        a, s = _angles_shifts(rng, n, angle_max, shift_frac * H)
        angs.append(a)
        shs.append(s)
    return np.stack(ys), np.stack(angs), np.stack(shs)


def _dev_tfs(angs, shs, H):
    from asr_amd import ops, transforms as T
    b, n = angs.shape
    rot = np.stack([T.rotation_transforms(angs[i], H, H) for i in range(b)])
    tr = np.stack([T.translation_transforms(shs[i]) for i in range(b)])
    irot = np.stack([T.inverse_transforms(rot[i]) for i in range(b)])
    itr = np.stack([T.inverse_transforms(tr[i]) for i in range(b)])
    return [ops.to_device(t) for t in (rot, tr, irot, itr)]


@pytest.mark.parametrize("H,h,n", [(64, 16, 5), (128, 32, 4), (64, 32, 3)])
def test_sr_forward_and_gradient_match_oracle(dev, H, h, n):
    from asr_amd import ops
    b = 2
    y, angs, shs = _sr_problem(3, b, n, H, h)
    lam = (1.0, 0.3, 0.7, 0.05)
    rot, tr, irot, itr = _dev_tfs(angs, shs, H)
    yd = ops.to_device(y)
    x0 = ops.sr_init_target(yd, (H, H))
    # perturb so TV / residuals are non-trivial
    rng = np.random.default_rng(4)
    x_np = x0.cpu().numpy() + 0.05 * rng.standard_normal((b, H, H)).astype(np.float32)
    xd = ops.to_device(x_np)
    resid = ops.sr_forward_residual(xd, yd, rot, tr)
    _, grad = ops.sr_backward_adam(xd, resid, irot, itr, lam, adam=None)
    terms = ops.sr_loss_terms(xd, resid).cpu().numpy()
    for i in range(b):
        sr = o_sr.Superresolution(*lam, num_aug=n, feature_size=(h, h), output_size=(H, H))
        tgt = torch.from_numpy(x_np[i][None, :, :, None])
        smp = torch.from_numpy(y[i][..., None])
        # x0 itself (bilinear upsample of copy 0)
        ref_x0 = tf_ops.resize_bilinear(smp[0:1], (H, H)).numpy()[0, :, :, 0]
        np.testing.assert_allclose(x0[i].cpu().numpy(), ref_x0, rtol=0, atol=1e-6)
        r_ref, dy, dx, df, tv, l2, l1 = sr.loss_terms(tgt, smp, angs[i], shs[i])
        np.testing.assert_allclose(resid[i].cpu().numpy(), r_ref.numpy()[..., 0], rtol=0, atol=2e-6)
        loss_ref, g_ref = sr.loss_and_grad(tgt, smp, angs[i], shs[i])
        # gradient: sum over n copies of O(1) terms; tolerance 2e-5 absolute
        np.testing.assert_allclose(grad[i].cpu().numpy(), g_ref.numpy()[0, :, :, 0], rtol=0, atol=2e-5)
        np.testing.assert_allclose(terms[i], [float(df), float(tv), float(l2), float(l1)], rtol=2e-5)


def test_sr_solve_trajectory_matches_oracle(dev):
    """10 AMSGrad iterations with the shipped hyper-parameters (test_SR.py:33-47).  The sign()
    in the TV gradient makes single pixels chaotic w.r.t. 1-ulp differences, so the trajectory is
    compared in the mean (1e-5), in the max (a few learning-rate steps) and on the thresholded mask."""
    from asr_amd import ops, transforms as T
    H, h, n, b, iters = 128, 32, 6, 2, 10
    y, angs, shs = _sr_problem(5, b, n, H, h)
    lam = (1.0, 0.3, 0.7, 0.0)
    rot, tr, irot, itr = _dev_tfs(angs, shs, H)
    yd = ops.to_device(y)
    xd = ops.sr_init_target(yd, (H, H))
    b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
    # image i starts at global Adam step i*iters (persistent counter of the reference)
    alphas = np.zeros((iters, b), np.float32)
    for i in range(b):
        for it in range(iters):
            lr = T.exponential_decay_lr(1e-3, 60, 0.3, it)
            alphas[it, i] = T.adam_alpha(lr, b1, b2, i * iters + it + 1)
    xd, terms = ops.sr_solve(xd, yd, rot, tr, irot, itr, ops.to_device(alphas), lam,
                             np.float32(1) - b1, np.float32(1) - b2, eps, True)
    got = xd.cpu().numpy()
    opt = o_sr.Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = o_sr.Superresolution(*lam, num_iter=iters, num_aug=n, optimizer=opt, feature_size=(h, h),
                              output_size=(H, H))
    for i in range(b):
        ref, loss = sr.augmented_superresolution(y[i][..., None], angs[i], shs[i])
        ref = ref[:, :, 0]
        d = np.abs(got[i] - ref)
        assert d.mean() < 1e-5, d.mean()
        assert d.max() < 5e-3, d.max()
        m_got = o_sr.threshold_image(got[i], 1, th_factor=0.2)
        m_ref = o_sr.threshold_image(ref, 1, th_factor=0.2)
        assert o_aug.single_class_IOU(m_ref, m_got, 1, False) >= 0.999
        t = terms[i].cpu().numpy()
        loss_got = lam[0] * t[0] + lam[1] * t[1] + lam[2] * t[2]
        assert abs(loss_got - loss) <= 1e-4 * abs(loss)


@pytest.mark.parametrize("mode", ["max", "mean"])
def test_realign_matches_oracle(dev, mode):
    from asr_amd import ops, transforms as T
    H, h, n, b = 128, 32, 5, 2
    y, angs, shs = _sr_problem(6, b, n, H, h)
    y = y + 0.1 * np.random.default_rng(7).random(y.shape, dtype=np.float32)
    rot_neg = np.stack([T.rotation_transforms(-angs[i], H, H) for i in range(b)])
    tr_neg = np.stack([T.translation_transforms(-shs[i]) for i in range(b)])
    got = ops.realign(ops.to_device(y), ops.to_device(tr_neg), ops.to_device(rot_neg), (H, H), mode).cpu().numpy()
    for i in range(b):
        sr = o_sr.Superresolution(1, 0, 0, 0, num_aug=n, feature_size=(h, h), output_size=(H, H))
        fn = sr.max_superresolution if mode == "max" else sr.mean_superresolution
        ref, _ = fn(y[i][..., None], angs[i], shs[i])
        np.testing.assert_allclose(got[i], ref[:, :, 0], rtol=0, atol=2e-6)
    # the fused pass returns exactly what the two separate calls return
    mx, mn = ops.realign(ops.to_device(y), ops.to_device(tr_neg), ops.to_device(rot_neg), (H, H), "both")
    assert np.array_equal((mx if mode == "max" else mn).cpu().numpy(), got)


def test_threshold_and_iou_bit_exact(dev):
    from asr_amd import ops
    rng = np.random.default_rng(8)
    img = rng.random((3, 96, 96), dtype=np.float32)
    th_mask = rng.random((3, 96, 96), dtype=np.float32)
    got = ops.threshold(ops.to_device(img), 8, th_factor=0.2, segments=3).cpu().numpy()
    got_m = ops.threshold(ops.to_device(img), 8, th_mask=ops.to_device(th_mask), segments=3).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], o_sr.threshold_image(img[i], 8, th_factor=0.2))
        assert np.array_equal(got_m[i], o_sr.threshold_image(img[i], 8, th_mask=th_mask[i]))
    truth = rng.choice(np.array([0, 8, 8, 3, 255], np.int32), size=(3, 96, 96))
    counts = ops.iou_counts(ops.to_device(truth, torch.int32), ops.to_device(got, torch.int32), 8, include_bg=True,
                            segments=3).cpu().numpy()
    counts_nb = ops.iou_counts(ops.to_device(truth, torch.int32), ops.to_device(got, torch.int32), 8, include_bg=False,
                               segments=3).cpu().numpy()
    for i in range(3):
        iou_c = counts_nb[i, 0] / counts_nb[i, 1]
        assert iou_c == o_aug.single_class_IOU(truth[i], got[i], 8, False)
        iou_bg = np.mean([counts[i, 0] / counts[i, 1], counts[i, 2] / counts[i, 3]])
        assert iou_bg == o_aug.single_class_IOU(truth[i], got[i], 8, True)


def test_opm_modes_match_oracle(dev):
    from asr_amd import ops
    rng = np.random.default_rng(9)
    logits = rng.standard_normal((4, 24, 24, 21)).astype(np.float32)
    logits[0, :4, :4] = 1.5                         # ties: first maximum must win
    ld = ops.to_device(logits)
    am = ops.argmax(ld).cpu().numpy()
    assert np.array_equal(am, np.argmax(logits, axis=-1))
    cm, _ = o_aug.opm(logits, 8, "argmax")
    assert np.array_equal(ops.opm_argmax(ld, 8).cpu().numpy(), np.stack(cm)[..., 0])
    cm, mm = o_aug.opm(logits, 8, "slice_max")
    g_c, g_m = ops.opm_slice_max(ld, 8)
    assert np.array_equal(g_c.cpu().numpy(), np.stack(cm)[..., 0])
    assert np.array_equal(g_m.cpu().numpy(), np.stack(mm)[..., 0])
    cm, _ = o_aug.opm(logits, 8, "slice")
    np.testing.assert_allclose(ops.opm_slice(ld, 8).cpu().numpy(), np.stack(cm)[..., 0], rtol=0, atol=1e-7)


@pytest.mark.parametrize("H,h", [(128, 32), (64, 32), (128, 16), (96, 16)])
def test_sr_solve_two_kernel_backward_is_bit_identical_to_fused(dev, H, h):
    """asr_sr_solve_* evaluates the translate stage of the gradient once per (copy, HR position) into a plane and gathers
    the rotation taps from it; asr_sr_backward_* (no workspace) nests the two stages in one kernel.  Same arithmetic ->
    the solver's x after k iterations equals k explicit forward + fused-backward steps bit for bit (f = 4, 2, 8, 6)."""
    from asr_amd import ops, transforms as T
    n, b, iters = 7, 2, 4
    y, angs, shs = _sr_problem(11, b, n, H, h)
    lam = (1.0, 0.3, 0.7, 0.0)
    rot, tr, irot, itr = _dev_tfs(angs, shs, H)
    yd = ops.to_device(y)
    b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
    alphas = np.zeros((iters, b), np.float32)
    for it in range(iters):
        alphas[it, :] = T.adam_alpha(np.float32(1e-3), b1, b2, it + 1)
    x_solve, _ = ops.sr_solve(ops.sr_init_target(yd, (H, H)), yd, rot, tr, irot, itr, ops.to_device(alphas), lam,
                              np.float32(1) - b1, np.float32(1) - b2, eps, True, want_loss=False)
    xd = ops.sr_init_target(yd, (H, H))
    m = torch.zeros_like(xd); v = torch.zeros_like(xd); vh = torch.zeros_like(xd)
    for it in range(iters):
        resid = ops.sr_forward_residual(xd, yd, rot, tr)
        xd, _ = ops.sr_backward_adam(xd, resid, irot, itr, lam,
                                     adam=dict(m=m, v=v, vhat=vh, alphas=ops.to_device(alphas[it]),
                                               one_minus_beta1=np.float32(1) - b1, one_minus_beta2=np.float32(1) - b2,
                                               epsilon=eps, amsgrad=True))
    assert torch.equal(x_solve, xd)


@pytest.mark.parametrize("n,H,h", [(70, 128, 32), (37, 64, 32)])
def test_sr_solve_does_not_depend_on_the_plane_chunking(dev, n, H, h):
    """The solver keeps the per-copy gradient planes of at most asr_sr_config.plane_chunk copies alive at once (default: an
    even split into chunks of <= 32) and carries the data-term sum across the chunks in copy order: the same float32
    additions as one pass over all copies, so x, m, v and vhat are bit-identical for every chunking -- all copies at once
    (the default below 1 GiB of planes), 24 / 19 (ragged tails of 22 / 18), 8 (multiples of the gather's unroll), 5 (the <8
    and <4 tails) and 1.  (The float64 loss terms are sums by atomics: equal to rounding, not bitwise.)"""
    from asr_amd import _lib, ops, transforms as T
    b, iters = 2, 3
    y, angs, shs = _sr_problem(21, b, n, H, h)
    lam = (1.0, 0.3, 0.7, 0.05)
    rot, tr, irot, itr = _dev_tfs(angs, shs, H)
    yd = ops.to_device(y)
    b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
    alphas = np.zeros((iters, b), np.float32)
    for it in range(iters):
        alphas[it, :] = T.adam_alpha(np.float32(1e-3), b1, b2, it + 1)
    ad = ops.to_device(alphas)

    def run(chunk):
        cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, eps, plane_chunk=chunk)
        st = {}
        x, terms = ops.sr_solve(ops.sr_init_target(yd, (H, H)), yd, rot, tr, irot, itr, ad, lam, cfg=cfg, state=st)
        return x, terms, st["m"], st["v"], st["vhat"]

    ref = run(n)
    legacy, _ = ops.sr_solve(ops.sr_init_target(yd, (H, H)), yd, rot, tr, irot, itr, ad, lam, np.float32(1) - b1,
                             np.float32(1) - b2, eps, True, want_loss=False)           # asr_sr_solve_f32: default chunking
    assert torch.equal(legacy, ref[0])
    for chunk in (0, 24 if n == 70 else 19, 8, 5, 1):
        got = run(chunk)
        for a, r in zip(got[:1] + got[2:], ref[:1] + ref[2:]):
            assert torch.equal(a, r), chunk
        np.testing.assert_allclose(got[1].cpu().numpy(), ref[1].cpu().numpy(), rtol=1e-12)


def test_mask_pipeline_entry_points(dev):
    """The per-image glue of HotPath as library kernels: global min-max normalisation of a mask stack
    (superres_utils.py:56-62,183-206), the fused standard-output mask (generate_standard_output.py:52-65 = Resizing +
    argmax + class filter) and the IoU counts of several masks against ONE label map (SR_single_class.py:109-120)."""
    from asr_amd import ops
    rng = np.random.default_rng(12)
    # min-max normalisation: argmax masks {0, 8} -> {0, 1}; a float stack; a constant stack (max == min -> den = 1)
    stack = rng.choice(np.array([0.0, 8.0], np.float32), size=(5, 32, 32))
    got = ops.minmax_normalize(ops.to_device(stack)).cpu().numpy()
    assert np.array_equal(got, np.stack([o_sr.min_max_normalization(m, 0.0, 1.0, stack.min(), stack.max()) for m in stack]).astype(np.float32))
    fl = rng.standard_normal((3, 16, 16)).astype(np.float32)
    got = ops.minmax_normalize(ops.to_device(fl), segments=3, new_min=0.0, new_max=255.0).cpu().numpy()
    for i in range(3):
        np.testing.assert_allclose(got[i], o_sr.min_max_normalization(fl[i], 0.0, 255.0).astype(np.float32), rtol=0, atol=2e-5)
    const = np.full((2, 8, 8), 3.0, np.float32)
    assert np.array_equal(ops.minmax_normalize(ops.to_device(const)).cpu().numpy(), np.zeros_like(const))
    # standard mask == resize (half-pixel bilinear) -> argmax (first maximum) -> class filter
    logits = rng.standard_normal((24, 20, 21)).astype(np.float32)
    logits[:6, :6] = 0.25                              # ties across classes: class 0 must win there
    ld = ops.to_device(logits)
    got = ops.standard_mask(ld, (96, 80), 8).cpu().numpy()
    padded = torch.full((1, 24, 20, 24), -3.0e38, dtype=torch.float32, device=ld.device)
    padded[0, :, :, :21] = ld
    am = ops.argmax(ops.resize_bilinear(padded, (96, 80)))[0].cpu().numpy()
    assert np.array_equal(got, np.where(am == 8, 8, 0))
    up = torch.nn.functional.interpolate(torch.from_numpy(logits).permute(2, 0, 1)[None], size=(96, 80), mode="bilinear",
                                         align_corners=False)[0].permute(1, 2, 0).numpy()
    ref = np.where(np.argmax(up, axis=-1) == 8, 8, 0)
    assert (got == ref).mean() >= 0.999 and (got == 8).any()
    # IoU counts of four masks against one label map == the replicated-truth form
    truth = rng.choice(np.array([0, 8, 8, 3, 255], np.int32), size=(96, 80))
    preds = rng.choice(np.array([0, 8], np.int32), size=(4, 96, 80))
    td, pd = ops.to_device(truth, torch.int32), ops.to_device(preds, torch.int32)
    for bg in (False, True):
        a = ops.iou_counts_shared_truth(td, pd, 8, include_bg=bg).cpu().numpy()
        b = ops.iou_counts(td[None].expand(4, -1, -1).contiguous(), pd, 8, include_bg=bg, segments=4).cpu().numpy()
        assert np.array_equal(a, b)


@pytest.mark.parametrize("classes", [21, 5, 40])
def test_class_activation_matches_torch(dev, classes):
    """Activation(last_activation) over the class axis (model.py:124-125): LDS-staged rows for <= 32 classes, the direct
    walk beyond; 1000 pixels = three full tiles of 256 rows and a ragged one."""
    from asr_amd import ops
    rng = np.random.default_rng(31)
    logits = (3.0 * rng.standard_normal((2, 20, 25, classes))).astype(np.float32)
    ld = ops.to_device(logits)
    sm = ops.class_activation(ld, "softmax").cpu().numpy()
    np.testing.assert_allclose(sm, torch.softmax(torch.from_numpy(logits), dim=-1).numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(sm.sum(-1), 1.0, atol=1e-6)
    sg = ops.class_activation(ld, "sigmoid").cpu().numpy()
    np.testing.assert_allclose(sg, torch.sigmoid(torch.from_numpy(logits)).numpy(), rtol=2e-6, atol=1e-7)
    assert torch.equal(ld, ops.to_device(logits))                         # the input is not touched
