"""optax.chain(clip_by_global_norm(max_grad_norm), adam(lr, eps=1e-5)) on one flat parameter buffer (rec_magpo.py:581-589, :412-420):
the state (count, mu, nu) of the reference's OptStates entry and the ``update`` function get_learner_fn receives, as one fused HIP kernel
(csrc/optim.hip: deterministic global norm + one pass over the buffer)."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import lib


class ClipAdam:
    def __init__(self, net, sys):
        """``net``: SableGuider / GruActor (flat parameters ``net.P.flat``, flat gradients ``net.grads``); ``sys``: SystemConfig (actor_lr,
        max_grad_norm, decay_learning_rates, lr_num_updates, ppo_epochs, num_minibatches)."""
        self.net, self.sys = net, sys
        self.mu, self.nu = torch.zeros_like(net.P.flat), torch.zeros_like(net.P.flat)
        self.count = 0
        self.last_lr = float(sys.actor_lr)
        self.L = lib()

    def learning_rate(self) -> float:
        """make_learning_rate (mava/utils/training.py:20-64): constant, or linear decay evaluated at the optimiser step count before the step."""
        s = self.sys
        if not s.decay_learning_rates:
            return float(s.actor_lr)
        return float(s.actor_lr) * (1.0 - (self.count // (s.ppo_epochs * s.num_minibatches)) / s.lr_num_updates)

    def update(self, grad_scale: float, ws64: torch.Tensor, gnorm_out: torch.Tensor) -> float:
        """One optimiser step on ``net`` from ``net.grads * grad_scale`` (the 1 / groups of the gradient mean is folded in here); the
        transposed weight copies of the network are rebuilt afterwards.  Returns the learning rate used."""
        s, net = self.sys, self.net
        cnt = self.count + 1
        bc1 = float(np.float32(1) - np.float32(0.9) ** np.float32(cnt))
        bc2 = float(np.float32(1) - np.float32(0.999) ** np.float32(cnt))
        lr = self.learning_rate()
        self.L.call("magpo_clip_adam", net.P.flat, net.grads, self.mu, self.nu, net.P.numel, grad_scale, s.max_grad_norm, lr,
                    0.9, 0.999, 1e-5, bc1, bc2, ws64, gnorm_out, torch.cuda.current_stream().cuda_stream)
        self.count = cnt
        self.last_lr = lr
        net.refresh()
        return lr
