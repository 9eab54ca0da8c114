"""Rollout only (eager), for rocprofv3 kernel traces of the acting path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dl = MagpoLearner(CoordSumConfig(4, 20, 100, 60), N, SystemConfig(), 'cuda', net_seed=0)
dl.use_graph = False
dl.setup(host_split(prng_key(42), 4)[0])
for it in range(3):
    t0 = time.time(); dl.rollout(); torch.cuda.synchronize(); t1 = time.time(); dl._carry_over()
    print(f"N={N} rollout {1e3*(t1-t0):.1f} ms", flush=True)
