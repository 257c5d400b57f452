"""CPU restatement of the reference's SR solvers in the TF-materialised formulation
(tile -> rotate -> translate -> resize, TF-rule backward, Keras Adam).  TEST INFRASTRUCTURE ONLY.

Follows superresolution_scripts/superresolution.py:8-161, superresolution_scripts/optimizer.py:4-52
and superresolution_scripts/superres_utils.py:56-62,118-139,213-273 of the reference.  The
arithmetic they delegate to TF (GradientTape, tf.optimizers.Adam, ExponentialDecay) is restated
from the pinned tensorflow==2.7.0 kernels.  PARITY UNPINNED (see package docstring).
"""
from __future__ import annotations

import numpy as np
import torch

from . import tf_ops

F32 = torch.float32


class KerasAdam:
    """tf.optimizers.Adam(learning_rate, beta_1, beta_2, epsilon, amsgrad) as used at
    optimizer.py:37-41.  Keras optimizer_v2/adam.py + training_ops ApplyAdam[WithAmsgrad]:
        t = iterations + 1 (GLOBAL, persists across variables -- SURVEY 3.3 quirk)
        alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
        m += (g - m) * (1 - beta1) ; v += (g*g - v) * (1 - beta2)
        [vhat = max(vhat, v)] ; var -= (m * alpha) / (sqrt(v or vhat) + epsilon)
    m / v / vhat slots are per variable (fresh for every new tf.Variable,
    superresolution.py:114)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False):
        self.learning_rate = np.float32(learning_rate)
        self.beta_1 = np.float32(beta_1)
        self.beta_2 = np.float32(beta_2)
        self.epsilon = np.float32(epsilon)
        self.amsgrad = bool(amsgrad)
        self.iterations = 0

    def new_slots(self, var):
        z = torch.zeros_like(var)
        return {"m": z.clone(), "v": z.clone(), "vhat": z.clone()}

    def alpha(self):
        t = np.float32(self.iterations + 1)
        b1p = np.power(self.beta_1, t, dtype=np.float32)
        b2p = np.power(self.beta_2, t, dtype=np.float32)
        one = np.float32(1.0)
        return np.float32(self.learning_rate * np.sqrt(one - b2p, dtype=np.float32) / (one - b1p))

    def apply(self, var, grad, slots):
        alpha = float(self.alpha())
        one_m_b1 = float(np.float32(1.0) - self.beta_1)
        one_m_b2 = float(np.float32(1.0) - self.beta_2)
        m, v = slots["m"], slots["v"]
        m += (grad - m) * one_m_b1
        v += (grad * grad - v) * one_m_b2
        # sqrt through numpy: torch's vectorised CPU sqrt is not correctly rounded (measured: ~0.6 %
        # of float32 inputs differ by 1 ulp from the IEEE result), Eigen's / numpy's is.
        if self.amsgrad:
            slots["vhat"] = torch.maximum(slots["vhat"], v)
            root = torch.from_numpy(np.sqrt(slots["vhat"].numpy()))
        else:
            root = torch.from_numpy(np.sqrt(v.numpy()))
        denom = root + float(self.epsilon)
        var -= (m * alpha) / denom
        self.iterations += 1

def _np(t):
    return t.numpy()            # shares memory with the torch tensor


class KerasSGD:
    """tf.optimizers.SGD(learning_rate, momentum, nesterov), optimizer.py:33-35.  training_ops
    ApplyGradientDescent (momentum == 0): var -= grad * lr; ApplyKerasMomentum:
        accum = accum * momentum - grad * lr ; var += nesterov ? accum * momentum - grad * lr : accum"""

    def __init__(self, learning_rate=1e-2, momentum=0.0, nesterov=False):
        self.learning_rate = np.float32(learning_rate)
        self.momentum = np.float32(momentum)
        self.nesterov = bool(nesterov)
        self.iterations = 0

    def new_slots(self, var):
        return {"m": torch.zeros_like(var)}

    def apply(self, var, grad, slots):
        lr, mom = self.learning_rate, self.momentum
        g, x = _np(grad), _np(var)
        if mom == 0:
            x -= g * lr
        else:
            acc = _np(slots["m"])
            acc[...] = acc * mom - g * lr
            if self.nesterov:
                x += acc * mom - g * lr
            else:
                x += acc
        self.iterations += 1


class KerasAdagrad:
    """tf.optimizers.Adagrad(learning_rate, initial_accumulator_value, epsilon), optimizer.py:24-27.
    ApplyAdagradV2: accum += grad^2 ; var -= grad * lr / (sqrt(accum) + epsilon)."""

    def __init__(self, learning_rate=1e-3, initial_accumulator_value=0.1, epsilon=1e-7):
        self.learning_rate = np.float32(learning_rate)
        self.initial_accumulator_value = np.float32(initial_accumulator_value)
        self.epsilon = np.float32(epsilon)
        self.iterations = 0

    def new_slots(self, var):
        return {"v": torch.full_like(var, float(self.initial_accumulator_value))}

    def apply(self, var, grad, slots):
        g, x, acc = _np(grad), _np(var), _np(slots["v"])
        acc += g * g
        x -= (g * self.learning_rate) / (np.sqrt(acc) + self.epsilon)
        self.iterations += 1


class KerasAdadelta:
    """tf.optimizers.Adadelta(learning_rate) with the Keras defaults rho=0.95, epsilon=1e-7,
    optimizer.py:21-23.  ApplyAdadelta:
        accum = accum * rho + grad^2 * (1 - rho)
        update = sqrt(accum_update + eps) * rsqrt(accum + eps) * grad
        var -= update * lr ; accum_update = accum_update * rho + update^2 * (1 - rho)"""

    def __init__(self, learning_rate=1e-3, rho=0.95, epsilon=1e-7):
        self.learning_rate = np.float32(learning_rate)
        self.rho = np.float32(rho)
        self.epsilon = np.float32(epsilon)
        self.iterations = 0

    def new_slots(self, var):
        return {"v": torch.zeros_like(var), "m": torch.zeros_like(var)}

    def apply(self, var, grad, slots):
        g, x, acc, au = _np(grad), _np(var), _np(slots["v"]), _np(slots["m"])
        one_m_rho = np.float32(1.0) - self.rho
        acc[...] = acc * self.rho + (g * g) * one_m_rho
        upd = np.sqrt(au + self.epsilon) * (np.float32(1.0) / np.sqrt(acc + self.epsilon)) * g
        x -= upd * self.learning_rate
        au[...] = au * self.rho + (upd * upd) * one_m_rho
        self.iterations += 1


class KerasAdamax:
    """tf.keras.optimizers.Adamax(learning_rate, epsilon, beta_1, beta_2), optimizer.py:28-32.
    ApplyAdaMax with beta1_power = beta1^(iterations + 1):
        m += (grad - m) * (1 - beta1) ; v = max(beta2 * v, |grad|)
        var -= lr / (1 - beta1_power) * (m / (v + epsilon))"""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate = np.float32(learning_rate)
        self.beta_1 = np.float32(beta_1)
        self.beta_2 = np.float32(beta_2)
        self.epsilon = np.float32(epsilon)
        self.iterations = 0

    def new_slots(self, var):
        return {"m": torch.zeros_like(var), "v": torch.zeros_like(var)}

    def apply(self, var, grad, slots):
        g, x, m, v = _np(grad), _np(var), _np(slots["m"]), _np(slots["v"])
        one = np.float32(1.0)
        b1p = np.power(self.beta_1, np.float32(self.iterations + 1), dtype=np.float32)
        m += (g - m) * (one - self.beta_1)
        v[...] = np.maximum(self.beta_2 * v, np.abs(g))
        x -= np.float32(self.learning_rate / (one - b1p)) * (m / (v + self.epsilon))
        self.iterations += 1


def exponential_decay(initial_lr, decay_steps, decay_rate, step):
    """tf.keras.optimizers.schedules.ExponentialDecay (non-staircase), optimizer.py:43-52:
    lr0 * rate ** (step / decay_steps), float32."""
    p = np.float32(step) / np.float32(decay_steps)
    return np.float32(np.float32(initial_lr) * np.power(np.float32(decay_rate), p, dtype=np.float32))


class Optimizer:
    """optimizer.py:4-52: the five Keras optimisers + optional ExponentialDecay."""

    def __init__(self, optimizer="adam", learning_rate=1e-3, epsilon=1e-7, beta_1=.9, beta_2=.999,
                 amsgrad=False, initial_accumulator_value=.1, momentum=.0, nesterov=False,
                 lr_scheduler=False, decay_steps=.5, decay_rate=100):
        self.learning_rate = learning_rate
        self.decay_steps = decay_steps
        self.decay_rate = decay_rate
        if optimizer == "adadelta":
            self.optimizer = KerasAdadelta(learning_rate)
        elif optimizer == "adagrad":
            self.optimizer = KerasAdagrad(learning_rate, initial_accumulator_value, epsilon)
        elif optimizer == "adamax":
            self.optimizer = KerasAdamax(learning_rate, beta_1, beta_2, epsilon)
        elif optimizer == "sgd":
            self.optimizer = KerasSGD(learning_rate, momentum, nesterov)
        else:
            self.optimizer = KerasAdam(learning_rate, beta_1, beta_2, epsilon, amsgrad)
        self.lr_scheduler = bool(lr_scheduler)

    def lr_decay(self, iteration):
        self.optimizer.learning_rate = exponential_decay(self.learning_rate, self.decay_steps,
                                                         self.decay_rate, iteration)


def _stack(copies):
    if isinstance(copies, (list, tuple)):
        copies = np.stack([np.asarray(c, dtype=np.float32) for c in copies])
    return torch.as_tensor(np.asarray(copies, dtype=np.float32))


def btv_pairs(shift_factor=2):
    """superresolution.py:10-12: [[h, v] for h in -s..s for v in 0..s]."""
    return [(h, v) for h in range(-shift_factor, shift_factor + 1) for v in range(0, shift_factor + 1)]


def btv_weight(alpha, h, v):
    """tf.pow(alpha, |h| + |v|) in float32 (superresolution.py:18-19): Eigen calls std::pow(float, float), which glibc
    rounds correctly -- restated as the double-precision power rounded once to float32 (numpy's own float32 pow is a
    SIMD approximation that can be 1 ulp off)."""
    import math
    return np.float32(math.pow(float(np.float32(alpha)), abs(h) + abs(v)))


def shift_zero(img, dx, dy):
    """tfa.image.translate(img, [dx, dy]) for integer shifts: out(x, y) = img(x - dx, y - dy), zero outside
    (bilinear weights are exactly 1 / 0 at integer positions).  img [..., H, W, 1] or [..., H, W]."""
    t = img[0, :, :, 0] if img.dim() == 4 else img
    H, W = t.shape
    out = torch.zeros_like(t)
    ys0, ys1 = max(dy, 0), min(H + dy, H)
    xs0, xs1 = max(dx, 0), min(W + dx, W)
    if ys1 > ys0 and xs1 > xs0:
        out[ys0:ys1, xs0:xs1] = t[ys0 - dy:ys1 - dy, xs0 - dx:xs1 - dx]
    return out.reshape(img.shape)


def bilateral_tv(target, alpha=0.6, shift_factor=2):
    """superresolution.py:8-23: sum_p alpha^(|h|+|v|) * || x - translate(x, p) ||_1 (float64 accumulation here:
    the scalar is only reported)."""
    total = 0.0
    for h, v in btv_pairs(shift_factor):
        total += float(btv_weight(alpha, h, v)) * float(torch.sum(torch.abs(target - shift_zero(target, h, v)).double()))
    return total


def bilateral_tv_grad(target, lambda_tv, alpha=0.6, shift_factor=2):
    """d(lambda_tv * bilateral_tv)/d target under TF's gradients: Abs -> sign; Sub -> (+g, -g);
    ImageProjectiveTransformV3 -> sample the upstream at q + p (zero outside); Tile -> sum over the pairs.
    Order: first every minuend term, then every translate term, each in pair order (the HIP kernel's order)."""
    lam = np.float32(lambda_tv)
    ga = torch.zeros_like(target)
    gb = torch.zeros_like(target)
    signs = []
    for h, v in btv_pairs(shift_factor):
        c = float(np.float32(lam * btv_weight(alpha, h, v)))
        s_p = torch.sign(target - shift_zero(target, h, v)) * c
        signs.append(s_p)
        ga += s_p
    for (h, v), s_p in zip(btv_pairs(shift_factor), signs):
        gb -= shift_zero(s_p, -h, -v)
    return ga + gb


class Superresolution:
    """superresolution.py:26-161."""

    def __init__(self, lambda_df, lambda_tv, lambda_L2, lambda_L1, num_iter=200, num_aug=100,
                 optimizer: Optimizer = None, feature_size=(64, 64), output_size=(512, 512), use_BTV=False,
                 verbose=False, copy_dropout=0.0):
        self.lambda_df = float(lambda_df)
        self.lambda_tv = float(lambda_tv)
        self.lambda_L2 = float(lambda_L2)
        self.lambda_L1 = float(lambda_L1)
        self.num_iter = num_iter
        self.num_aug = num_aug
        self.optimizer = optimizer
        self.feature_size = tuple(feature_size)
        self.output_size = tuple(output_size)
        self.use_BTV = bool(use_BTV)
        self.verbose = verbose
        self.copy_dropout = copy_dropout
        self._drop_masks = {}

    def drop_mask(self, n_drop):
        """superresolution.py:47-50.  The shuffle runs inside a @tf.function, i.e. once, at trace time: the mask
        is frozen for every later iteration and image solved by this object."""
        if n_drop not in self._drop_masks:
            mask = np.full(self.num_aug, fill_value=True)
            mask[:n_drop] = False
            np.random.shuffle(mask)
            self._drop_masks[n_drop] = mask
        return self._drop_masks[n_drop]

    # -- superresolution.py:44-100 ---------------------------------------------------
    def forward_model(self, target, angles, shifts):
        n = len(angles)
        tiled = target.expand(n, *target.shape[1:])
        rot = tf_ops.rotate(tiled, angles)
        aug = tf_ops.translate(rot, shifts)
        return tf_ops.resize_bilinear(aug, self.feature_size)

    def loss_terms(self, target, samples, angles, shifts):
        d = self.forward_model(target, angles, shifts)
        resid = d - samples
        df = torch.sum(resid * resid)
        dy, dx = tf_ops.image_gradients(target)
        if self.use_BTV:
            tv = torch.tensor(bilateral_tv(target), dtype=F32)
        else:
            tv = torch.sum(torch.abs(dy) + torch.abs(dx))
        l2 = torch.sum(target * target)
        l1 = torch.sum(torch.abs(target))
        return resid, dy, dx, df, tv, l2, l1

    def loss_function(self, target, samples, angles, shifts, n_drop=0):
        target = torch.as_tensor(np.asarray(target, dtype=np.float32))
        samples = _stack(samples)
        if n_drop != 0:
            mask = self.drop_mask(n_drop)
            samples, angles, shifts = samples[torch.as_tensor(mask)], np.asarray(angles)[mask], np.asarray(shifts)[mask]
        _, _, _, df, tv, l2, l1 = self.loss_terms(target, samples, angles, shifts)
        loss = self.lambda_df * df + self.lambda_tv * tv
        loss = loss + self.lambda_L2 * l2
        if self.lambda_L1 > 0.0:
            loss = loss + self.lambda_L1 * l1
        return float(loss)

    def loss_and_grad(self, target, samples, angles, shifts):
        """What tf.GradientTape.gradient(loss, [target]) yields at superresolution.py:133, with
        TF's registered gradients: SquaredDifference -> 2*g*(x-y); ResizeBilinearGrad (exact
        adjoint); ImageProjectiveTransformV3 grad (inverse-warp of the upstream gradient, twice:
        translate then rotate); Tile grad (sum over copies); Abs grad (sign); Square grad (2x)."""
        h, w = self.output_size
        resid, dy, dx, df, tv, l2, l1 = self.loss_terms(target, samples, angles, shifts)
        loss = self.lambda_df * df + self.lambda_tv * tv
        loss = loss + self.lambda_L2 * l2
        if self.lambda_L1 > 0.0:
            loss = loss + self.lambda_L1 * l1
        g_d = (2.0 * self.lambda_df) * resid
        g_t = tf_ops.resize_bilinear_grad(g_d, (h, w))
        g_r = tf_ops.projective_transform_grad(g_t, tf_ops.translations_to_projective_transforms(shifts), (h, w))
        g_x = tf_ops.projective_transform_grad(g_r, tf_ops.angles_to_projective_transforms(angles, h, w), (h, w))
        g_df = torch.zeros_like(target)
        for i in range(g_x.shape[0]):          # fixed order n = 0..N-1 (the HIP kernel's order)
            g_df[0] += g_x[i]
        if self.use_BTV:
            g_tv = bilateral_tv_grad(target, self.lambda_tv)
        else:
            sy = torch.sign(dy) * self.lambda_tv
            sx = torch.sign(dx) * self.lambda_tv
            g_tv = -sy - sx
            g_tv[:, 1:] += sy[:, :-1]
            g_tv[:, :, 1:] += sx[:, :, :-1]
        grad = g_df + g_tv + (2.0 * self.lambda_L2) * target
        if self.lambda_L1 > 0.0:
            grad = grad + self.lambda_L1 * torch.sign(target)
        return loss, grad

    # -- superresolution.py:102-137 --------------------------------------------------
    def augmented_superresolution(self, augmented_copies, angles, shifts, return_trajectory=False):
        if self.optimizer is None:
            raise Exception("You must provide an instance of the Optimizer class to compute the augmented SR")
        samples = _stack(augmented_copies)
        angles = np.asarray(angles, dtype=np.float32)
        shifts = np.asarray(shifts, dtype=np.float32)
        target = tf_ops.resize_bilinear(samples[0:1], self.output_size).clone()
        n_drop = int(self.num_aug * self.copy_dropout)
        if n_drop != 0:
            mask = self.drop_mask(n_drop)
            samples, angles, shifts = samples[torch.as_tensor(mask)], angles[mask], shifts[mask]
        adam = self.optimizer.optimizer
        slots = adam.new_slots(target)
        loss = None
        traj = []
        for i in range(self.num_iter):
            if self.optimizer.lr_scheduler:
                self.optimizer.lr_decay(i)
            loss, grad = self.loss_and_grad(target, samples, angles, shifts)
            if self.verbose and (i % 10 == 0 or i == self.num_iter - 1):
                print(f"{i + 1}/{self.num_iter} -- loss = {float(loss)}")
            adam.apply(target, grad, slots)
            if return_trajectory:
                traj.append(target[0].numpy().copy())
        out = target[0].numpy().copy()
        if return_trajectory:
            return out, float(loss), traj
        return out, float(loss)

    # -- superresolution.py:139-161 --------------------------------------------------
    def _realign(self, augmented_copies, angles, shifts):
        samples = _stack(augmented_copies)
        up = tf_ops.resize_bilinear(samples, self.output_size)
        tr = tf_ops.translate(up, -np.asarray(shifts, dtype=np.float32))
        return tf_ops.rotate(tr, -np.asarray(angles, dtype=np.float32))

    def max_superresolution(self, augmented_copies, angles, shifts):
        return torch.amax(self._realign(augmented_copies, angles, shifts), dim=0).numpy(), None

    def mean_superresolution(self, augmented_copies, angles, shifts):
        r = self._realign(augmented_copies, angles, shifts)
        acc = torch.zeros_like(r[0])
        for i in range(r.shape[0]):            # fixed order, then one divide (tf.reduce_mean)
            acc += r[i]
        return (acc / np.float32(r.shape[0])).numpy(), None


def min_max_normalization(image, new_min=0.0, new_max=255.0, global_min=None, global_max=None):
    """superres_utils.py:56-62."""
    image = np.asarray(image)
    mn = image.min() if global_min is None else global_min
    mx = image.max() if global_max is None else global_max
    num = (image - mn) * (new_max - new_min)
    den = (mx - mn) if (mx - mn) != 0 else 1.0
    return new_min + (num / den)


def threshold_image(image, th_value, th_factor=.15, th_mask=None):
    """superres_utils.py:118-139: image >= th_mask, or image > th_factor * max(image) (f32);
    returns int32 {0, th_value}."""
    image = np.asarray(image, dtype=np.float32)
    if th_mask is not None:
        return np.where(image >= np.asarray(th_mask, dtype=np.float32), th_value, 0).astype(np.int32)
    max_value = np.float32(image.max()) * np.float32(th_factor)
    return np.where(image > max_value, th_value, 0).astype(np.int32)


def compute_SR(superresolution_obj, class_masks, angles, shifts, SR_type="aug", max_masks=(),
               class_id=8, th_factor=0.15):
    """superres_utils.py:213-273 without the file IO."""
    fn = {"aug": superresolution_obj.augmented_superresolution,
          "mean": superresolution_obj.mean_superresolution,
          "max": superresolution_obj.max_superresolution}[SR_type]
    target_class, _ = fn(class_masks, angles, shifts)
    if max_masks is not None and len(max_masks) == len(class_masks):
        target_max, _ = fn(max_masks, angles, shifts)
        return threshold_image(target_class, class_id, th_mask=target_max)
    return threshold_image(target_class, class_id, th_factor=th_factor)
