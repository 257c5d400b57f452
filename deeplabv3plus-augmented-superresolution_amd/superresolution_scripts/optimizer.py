"""``Optimizer`` with the reference's constructor and ``lr_decay`` (superresolution_scripts/
optimizer.py:4-52).  The Keras optimizer object it wrapped is replaced by a small state object
(``AdamState``, ``SGDState``, ``AdagradState``, ``AdadeltaState``, ``AdamaxState``): the
hyper-parameters plus the GLOBAL step counter that persists across solves (the reference reuses
one tf.optimizers object for every image, SR_single_class.py:66-70, so its ``iterations`` keeps
growing while the slot variables are fresh per new tf.Variable).  The arithmetic of the update runs
inside the HIP solver (asr_sr_solve_cfg_f32); this module only produces, per iteration, the float32
scalar the TF kernel would receive (``step_size``) and the kernel configuration (``config``).
"""
from __future__ import annotations

import numpy as np

from .. import _lib, ops, transforms as T

_f32 = np.float32


class _State:
    kind = _lib.OPT_ADAM
    slot_init = {}

    def __init__(self, learning_rate):
        self.learning_rate = _f32(learning_rate)
        self.iterations = 0          # global, never reset (Keras optimizer.iterations)

    def step_size(self):
        """The scalar of the NEXT apply_gradients call (t = iterations + 1) at the current learning rate."""
        return self.learning_rate

    def config(self, use_btv=False):
        raise NotImplementedError


class AdamState(_State):
    """tf.optimizers.Adam(learning_rate, beta_1, beta_2, epsilon, amsgrad), optimizer.py:37-41."""
    kind = _lib.OPT_ADAM

    def __init__(self, learning_rate, beta_1, beta_2, epsilon, amsgrad):
        super().__init__(learning_rate)
        self.beta_1 = _f32(beta_1)
        self.beta_2 = _f32(beta_2)
        self.epsilon = _f32(epsilon)
        self.amsgrad = bool(amsgrad)

    def step_size(self):
        return T.adam_alpha(self.learning_rate, self.beta_1, self.beta_2, self.iterations + 1)

    def config(self, use_btv=False):
        one = _f32(1.0)
        return ops.sr_config(self.kind, self.amsgrad, one - self.beta_1, one - self.beta_2, self.epsilon, use_btv)


class SGDState(_State):
    """tf.optimizers.SGD(learning_rate, momentum, nesterov), optimizer.py:33-35."""
    kind = _lib.OPT_SGD

    def __init__(self, learning_rate, momentum, nesterov):
        super().__init__(learning_rate)
        self.momentum = _f32(momentum)
        self.nesterov = bool(nesterov)

    def config(self, use_btv=False):
        return ops.sr_config(self.kind, self.nesterov, self.momentum, 0.0, 0.0, use_btv)


class AdagradState(_State):
    """tf.optimizers.Adagrad(learning_rate, initial_accumulator_value, epsilon), optimizer.py:24-27."""
    kind = _lib.OPT_ADAGRAD

    def __init__(self, learning_rate, initial_accumulator_value, epsilon):
        super().__init__(learning_rate)
        self.initial_accumulator_value = _f32(initial_accumulator_value)
        self.epsilon = _f32(epsilon)
        self.slot_init = {"v": float(self.initial_accumulator_value)}

    def config(self, use_btv=False):
        return ops.sr_config(self.kind, False, 0.0, 0.0, self.epsilon, use_btv)


class AdadeltaState(_State):
    """tf.optimizers.Adadelta(learning_rate) -- Keras defaults rho=0.95, epsilon=1e-7, optimizer.py:21-23."""
    kind = _lib.OPT_ADADELTA

    def __init__(self, learning_rate, rho=0.95, epsilon=1e-7):
        super().__init__(learning_rate)
        self.rho = _f32(rho)
        self.epsilon = _f32(epsilon)

    def config(self, use_btv=False):
        return ops.sr_config(self.kind, False, self.rho, _f32(1.0) - self.rho, self.epsilon, use_btv)


class AdamaxState(_State):
    """tf.keras.optimizers.Adamax(learning_rate, epsilon, beta_1, beta_2), optimizer.py:28-32."""
    kind = _lib.OPT_ADAMAX

    def __init__(self, learning_rate, beta_1, beta_2, epsilon):
        super().__init__(learning_rate)
        self.beta_1 = _f32(beta_1)
        self.beta_2 = _f32(beta_2)
        self.epsilon = _f32(epsilon)

    def step_size(self):
        """lr / (1 - beta1^t), the scalar ApplyAdaMax forms from (lr, beta1_power)."""
        b1p = np.power(self.beta_1, _f32(self.iterations + 1), dtype=np.float32)
        return _f32(self.learning_rate / (_f32(1.0) - b1p))

    def config(self, use_btv=False):
        return ops.sr_config(self.kind, False, _f32(1.0) - self.beta_1, self.beta_2, self.epsilon, use_btv)


class Optimizer:
    def __init__(self, optimizer="adam", learning_rate=1e-3,
                 epsilon=1e-7, beta_1=.9, beta_2=.999, amsgrad=False,
                 initial_accumulator_value=.1, momentum=.0, nesterov=False,
                 lr_scheduler=False, decay_steps=.5, decay_rate=100) -> None:
        self.learning_rate = learning_rate
        self.epsilon = epsilon
        self.beta_1 = beta_1
        self.beta_2 = beta_2
        self.amsgrad = amsgrad
        self.initial_accumulator_value = initial_accumulator_value
        self.momentum = momentum
        self.nesterov = nesterov
        self.decay_steps = decay_steps
        self.decay_rate = decay_rate
        if optimizer == "adadelta":
            self.optimizer = AdadeltaState(learning_rate)
        elif optimizer == "adagrad":
            self.optimizer = AdagradState(learning_rate, initial_accumulator_value, epsilon)
        elif optimizer == "adamax":
            self.optimizer = AdamaxState(learning_rate, beta_1, beta_2, epsilon)
        elif optimizer == "sgd":
            self.optimizer = SGDState(learning_rate, momentum, nesterov)
        else:   # like the reference, any other string falls through to Adam (optimizer.py:36-41)
            self.optimizer = AdamState(learning_rate, beta_1, beta_2, epsilon, amsgrad)
        self.lr_scheduler = bool(lr_scheduler)

    def lr_decay(self, iteration):
        self.optimizer.learning_rate = T.exponential_decay_lr(self.learning_rate, self.decay_steps,
                                                              self.decay_rate, iteration)

    def schedule_alphas(self, num_iter):
        """float32 [num_iter] per-step scalars for one solve, advancing the global step counter exactly
        as num_iter apply_gradients calls would (superresolution.py:120-135)."""
        out = np.empty(num_iter, np.float32)
        for i in range(num_iter):
            if self.lr_scheduler:
                self.lr_decay(i)
            out[i] = self.optimizer.step_size()
            self.optimizer.iterations += 1
        return out
