"""Throughput at the paper's batch shape (64 envs x 2 groups, tuned 8x15-100 hyper-parameters).  (Round 3: replaying the minibatch pass as a
HIP graph was built and measured here -- 905 vs 910 ms per update step: the ~3 400 kernels of a pass are bound by per-kernel dispatch on the
device, not by host launches -- and dropped.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
l = MagpoLearner(CoordSumConfig(8, 15, 100, 100), 64, SystemConfig(ppo_epochs=15, num_minibatches=8), "cuda", net_seed=0, n_block=2, num_groups=2)
l.setup(host_split(prng_key(1), 4)[0], n_groups=2)
for _ in range(2): l.update_step()
torch.cuda.synchronize(); t0 = time.time()
n = 3
for _ in range(n): l.update_step()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"{dt * 1e3:.0f} ms per update step, {2 * 64 * 128 / dt:.0f} env-steps/s")
