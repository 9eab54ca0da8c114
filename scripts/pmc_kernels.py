"""One update step at the bench shapes (for rocprofv3 --pmc passes; see scripts/pmc_collect.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dl = MagpoLearner(CoordSumConfig(4, 20, 100, 60), N, SystemConfig(ppo_epochs=1, num_minibatches=2), 'cuda', net_seed=0)
dl.use_graph = False
dl.setup(host_split(prng_key(42), 4)[0])
dl.update_step()
torch.cuda.synchronize()
