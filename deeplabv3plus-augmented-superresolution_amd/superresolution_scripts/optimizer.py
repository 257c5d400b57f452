"""``Optimizer`` with the reference's constructor and ``lr_decay`` (superresolution_scripts/
optimizer.py:4-52).  The Keras optimizer object it wrapped is replaced by ``AdamState``: the
hyper-parameters plus the GLOBAL step counter that persists across solves (the reference
reuses one tf.optimizers.Adam for every image, SR_single_class.py:66-70, so its ``iterations``
keeps growing while m / v slots are fresh per new variable).  The arithmetic of the update runs
inside asr_sr_backward_adam_f32; this module only produces the per-iteration float32 step size
alpha_t = lr_t * sqrt(1 - beta2^t) / (1 - beta1^t).
"""
from __future__ import annotations

import numpy as np

from .. import transforms as T


class AdamState:
    def __init__(self, learning_rate, beta_1, beta_2, epsilon, amsgrad):
        self.learning_rate = np.float32(learning_rate)
        self.beta_1 = np.float32(beta_1)
        self.beta_2 = np.float32(beta_2)
        self.epsilon = np.float32(epsilon)
        self.amsgrad = bool(amsgrad)
        self.iterations = 0          # global, never reset (Keras optimizer.iterations)

    def step_size(self):
        """alpha for the NEXT step (t = iterations + 1) at the current learning rate."""
        return T.adam_alpha(self.learning_rate, self.beta_1, self.beta_2, self.iterations + 1)


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
        if optimizer in ("adadelta", "adagrad", "adamax", "sgd"):
            # sweep-only choices (configs/sweep_configs/sweep_all.yaml:34-38): SURVEY 8f item 3
            raise NotImplementedError(f"optimizer '{optimizer}' is not on the accelerated path (Adam/AMSGrad only)")
        # like the reference, any other string falls through to Adam (optimizer.py:36-41)
        self.optimizer = AdamState(learning_rate, beta_1, beta_2, epsilon, amsgrad)
        self.lr_scheduler = bool(lr_scheduler)

    def lr_decay(self, iteration):
        self.optimizer.learning_rate = T.exponential_decay_lr(self.learning_rate, self.decay_steps,
                                                              self.decay_rate, iteration)

    def schedule_alphas(self, num_iter):
        """float32 [num_iter] step sizes for one solve, advancing the global step counter exactly
        as num_iter apply_gradients calls would (superresolution.py:120-135)."""
        out = np.empty(num_iter, np.float32)
        for i in range(num_iter):
            if self.lr_scheduler:
                self.lr_decay(i)
            out[i] = self.optimizer.step_size()
            self.optimizer.iterations += 1
        return out
