"""A few update steps at the tuned small-batch shape (64 envs, 8x15, 15 epochs x 8 minibatches) for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
l = MagpoLearner(CoordSumConfig(8, 15, 100, 100), 64, SystemConfig(ppo_epochs=15, num_minibatches=8), "cuda", net_seed=0, n_block=2)
l.setup(host_split(prng_key(1), 4)[0])
for _ in range(2):
    l.update_step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    l.update_step()
torch.cuda.synchronize()
print("s per update step", (time.time() - t0) / 3)
