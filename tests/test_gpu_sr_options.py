"""the sweep-only solver options -- the other Keras optimisers (optimizer.py:21-35), the
bilateral-TV prior (superresolution.py:8-23) and copy_dropout (superresolution.py:47-53) -- HIP solver vs oracle."""
import numpy as np
import pytest
import torch

from oracle import sr as o_sr
from test_gpu_warp_sr import _dev_tfs, _sr_problem

pytestmark = pytest.mark.gpu

LAM = (1.0, 0.3, 0.7, 0.05)


def _pair(kind, lam=LAM, iters=8, n=6, H=64, h=16, use_BTV=False, copy_dropout=0.0, **kw):
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    common = dict(lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    opt = Optimizer(kind, kw.pop("lr", 1e-3), **common, **kw)
    o_opt = o_sr.Optimizer(kind, opt.learning_rate, **common, **kw)
    args = dict(num_iter=iters, num_aug=n, feature_size=(h, h), output_size=(H, H), use_BTV=use_BTV,
                copy_dropout=copy_dropout)
    return Superresolution(*lam, optimizer=opt, **args), o_sr.Superresolution(*lam, optimizer=o_opt, **args)


@pytest.mark.parametrize("kind,kw", [
    ("sgd", dict(lr=2e-3)),
    ("sgd", dict(lr=2e-3, momentum=0.9)),
    ("sgd", dict(lr=2e-3, momentum=0.9, nesterov=True)),
    ("adagrad", dict(lr=1e-2)),
    ("adadelta", dict(lr=1.0)),
    ("adamax", dict(lr=1e-3)),
    ("adam", dict(lr=1e-3)),
])
def test_optimizers_match_oracle(dev, kind, kw):
    """Two images solved one after the other with ONE optimizer object each side (the persistent step counter matters
    for Adamax / Adam).  Same operation order as the oracle -> expected bit-identical; asserted to 1e-6 / 1e-4."""
    H, h, n, iters = 64, 16, 6, 8
    y, angs, shs = _sr_problem(11, 2, n, H, h)
    sr, ref_sr = _pair(kind, iters=iters, n=n, H=H, h=h, **dict(kw))
    for i in range(2):
        got, loss = sr.augmented_superresolution(y[i][..., None], angs[i], shs[i])
        ref, ref_loss = ref_sr.augmented_superresolution(y[i][..., None], angs[i], shs[i])
        d = np.abs(got - ref)
        assert d.mean() < 1e-6 and d.max() < 1e-4, (kind, kw, i, d.mean(), d.max())
        assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    assert sr.optimizer.optimizer.iterations == ref_sr.optimizer.optimizer.iterations == 2 * iters


def test_bilateral_tv_gradient_and_solve_match_oracle(dev):
    from asr_amd import ops
    H, h, n, b = 64, 16, 5, 2
    y, angs, shs = _sr_problem(12, b, n, H, h)
    rot, tr, irot, itr = _dev_tfs(angs, shs, H)
    yd = ops.to_device(y)
    x0 = ops.sr_init_target(yd, (H, H))
    x_np = x0.cpu().numpy() + 0.05 * np.random.default_rng(4).standard_normal((b, H, H)).astype(np.float32)
    xd = ops.to_device(x_np)
    cfg = ops.sr_config(use_btv=True)
    resid = ops.sr_forward_residual(xd, yd, rot, tr)
    _, grad = ops.sr_backward(xd, resid, irot, itr, LAM, cfg, state=None)
    terms = ops.sr_loss_terms(xd, resid, cfg).cpu().numpy()
    for i in range(b):
        ref_sr = o_sr.Superresolution(*LAM, num_aug=n, feature_size=(h, h), output_size=(H, H), use_BTV=True)
        tgt = torch.from_numpy(x_np[i][None, :, :, None])
        smp = torch.from_numpy(y[i][..., None])
        _, g_ref = ref_sr.loss_and_grad(tgt, smp, angs[i], shs[i])
        np.testing.assert_allclose(grad[i].cpu().numpy(), g_ref.numpy()[0, :, :, 0], rtol=0, atol=2e-5)
        assert abs(terms[i][1] - o_sr.bilateral_tv(tgt)) <= 1e-5 * terms[i][1]
    sr, ref_sr = _pair("adam", iters=8, n=n, H=H, h=h, use_BTV=True, amsgrad=True)
    got, loss = sr.augmented_superresolution(y[0][..., None], angs[0], shs[0])
    ref, ref_loss = ref_sr.augmented_superresolution(y[0][..., None], angs[0], shs[0])
    d = np.abs(got - ref)
    assert d.mean() < 1e-5 and d.max() < 5e-3, (d.mean(), d.max())
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    assert abs(sr.loss_function(ref[None], y[0][..., None], angs[0], shs[0]) -
               ref_sr.loss_function(ref[None], y[0][..., None], angs[0], shs[0])) <= 1e-4 * abs(ref_loss)


def test_copy_dropout_matches_oracle_and_freezes_its_mask(dev):
    H, h, n, iters = 64, 16, 10, 6
    y, angs, shs = _sr_problem(13, 2, n, H, h)
    sr, ref_sr = _pair("adam", iters=iters, n=n, H=H, h=h, copy_dropout=0.3, amsgrad=True)
    np.random.seed(21)
    outs = [sr.augmented_superresolution(y[i][..., None], angs[i], shs[i])[0] for i in range(2)]
    mask = sr._drop_mask(3).copy()
    np.random.seed(21)
    refs = [ref_sr.augmented_superresolution(y[i][..., None], angs[i], shs[i])[0] for i in range(2)]
    assert np.array_equal(mask, ref_sr.drop_mask(3)) and mask.sum() == 7      # drawn once, reused for image 2
    for got, ref in zip(outs, refs):
        d = np.abs(got - ref)
        assert d.mean() < 1e-5 and d.max() < 5e-3, (d.mean(), d.max())
    # dropping copies changes the data term: not the result of the full stack
    full, _ = _pair("adam", iters=iters, n=n, H=H, h=h, amsgrad=True)[0].augmented_superresolution(y[0][..., None], angs[0], shs[0])
    assert np.abs(full - outs[0]).max() > 1e-4


def test_module_level_bilateral_tv_matches_oracle(dev):
    """superresolution.py:8-23 as a free function (odd sizes, other alpha / window than the solver's defaults)."""
    from asr_amd.superresolution_scripts.superresolution import bilateral_tv
    from asr_amd.superresolution_scripts import superres_utils as su
    rng = np.random.default_rng(77)
    img = rng.random((1, 37, 53, 1), dtype=np.float32)
    for alpha, s in ((0.6, 2), (0.8, 3), (0.5, 1)):
        ref = o_sr.bilateral_tv(torch.from_numpy(img), alpha=alpha, shift_factor=s)
        got = bilateral_tv(img, alpha=alpha, shift_factor=s)
        assert abs(got - ref) <= 1e-6 * ref, (alpha, s, got, ref)
    a, b2 = bilateral_tv(img[0, :, :, 0]), bilateral_tv(img)      # float64 atomics: the summation order is not fixed
    assert abs(a - b2) <= 1e-12 * abs(b2)
    assert su.check_hdf5_validity is su.check_validity


def test_verbose_prints_the_reference_loss_lines_and_changes_no_update(dev, capsys):
    """superresolution.py:130-131: with verbose the loss of iteration i is printed for i % 10 == 0 and for the last
    iteration, as "{i+1}/{num_iter} -- loss = ...".  The solver is cut into several calls for that; the result must be
    the single-call result bit for bit, and the printed values the oracle's."""
    import re
    H, h, n, iters = 64, 16, 6, 23
    y, angs, shs = _sr_problem(12, 1, n, H, h)
    quiet, ref_sr = _pair("adam", iters=iters, n=n, H=H, h=h, amsgrad=True)
    loud, _ = _pair("adam", iters=iters, n=n, H=H, h=h, amsgrad=True)
    loud.verbose = True
    ref_sr.verbose = True
    a, loss_a = quiet.augmented_superresolution(y[0][..., None], angs[0], shs[0])
    capsys.readouterr()
    b, loss_b = loud.augmented_superresolution(y[0][..., None], angs[0], shs[0])
    got_lines = capsys.readouterr().out.strip().splitlines()
    ref_sr.augmented_superresolution(y[0][..., None], angs[0], shs[0])
    ref_lines = capsys.readouterr().out.strip().splitlines()
    assert np.array_equal(a, b) and loss_a == loss_b
    pat = re.compile(r"^(\d+)/(\d+) -- loss = (\S+)$")
    got = [pat.match(l).groups() for l in got_lines]
    ref = [pat.match(l).groups() for l in ref_lines]
    assert [g[:2] for g in got] == [r[:2] for r in ref] == [(str(i + 1), str(iters)) for i in (0, 10, 20, 22)]
    for g, r in zip(got, ref):
        assert abs(float(g[2]) - float(r[2])) <= 1e-4 * abs(float(r[2]))
    assert loud.optimizer.optimizer.iterations == quiet.optimizer.optimizer.iterations == iters
