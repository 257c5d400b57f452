"""Diagnostic (not a test): stage-by-stage bitwise comparison of the SR kernels with the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sr as o_sr, tf_ops  # noqa: E402
from asr_amd import ops, transforms as T  # noqa: E402
from test_gpu_warp_sr import _sr_problem, _dev_tfs  # noqa: E402


def rep(name, got, ref):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    print(f"{name:12s} max={d.max():.3e} mean={d.mean():.3e} mismatches={(got != ref).sum()}/{got.size}", flush=True)


def main():
    H, h, n, b = 128, 32, 6, 1
    y, angs, shs = _sr_problem(5, 2, n, H, h)
    y, angs, shs = y[:1], angs[:1], shs[:1]
    lam = (1.0, 0.3, 0.7, 0.0)
    rot, tr, irot, itr = _dev_tfs(angs, shs, H)
    yd = ops.to_device(y)
    xd = ops.sr_init_target(yd, (H, H))
    sr = o_sr.Superresolution(*lam, num_aug=n, feature_size=(h, h), output_size=(H, H))
    smp = torch.from_numpy(y[0][..., None])
    x_ref = tf_ops.resize_bilinear(smp[0:1], (H, H)).clone()
    rep("x0", xd[0].cpu().numpy(), x_ref.numpy()[0, :, :, 0])
    adam = o_sr.KerasAdam(1e-3, amsgrad=True)
    slots = adam.new_slots(x_ref)
    m = torch.zeros_like(xd); v = torch.zeros_like(xd); vh = torch.zeros_like(xd)
    b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
    for it in range(6):
        lr = T.exponential_decay_lr(1e-3, 60, 0.3, it)
        adam.learning_rate = lr
        resid = ops.sr_forward_residual(xd, yd, rot, tr)
        r_ref, dy, dx, *_ = sr.loss_terms(x_ref, smp, angs[0], shs[0])
        rep(f"it{it} resid", resid[0].cpu().numpy(), r_ref.numpy()[..., 0])
        _, g_ref = sr.loss_and_grad(x_ref, smp, angs[0], shs[0])
        alphas = ops.to_device(np.array([T.adam_alpha(lr, b1, b2, it + 1)], np.float32))
        x_new, grad = ops.sr_backward_adam(xd, resid, irot, itr, lam,
                                           adam=dict(m=m, v=v, vhat=vh, alphas=alphas, one_minus_beta1=np.float32(1) - b1,
                                                     one_minus_beta2=np.float32(1) - b2, epsilon=eps, amsgrad=True),
                                           want_grad=True)
        rep(f"it{it} grad", grad[0].cpu().numpy(), g_ref.numpy()[0, :, :, 0])
        adam.apply(x_ref, g_ref, slots)
        rep(f"it{it} m", m[0].cpu().numpy(), slots["m"].numpy()[0, :, :, 0])
        rep(f"it{it} v", v[0].cpu().numpy(), slots["v"].numpy()[0, :, :, 0])
        rep(f"it{it} x", x_new[0].cpu().numpy(), x_ref.numpy()[0, :, :, 0])
        xd = x_new


if __name__ == "__main__":
    main()
