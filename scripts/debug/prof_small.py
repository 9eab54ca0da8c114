"""cProfile of the host side at the tuned small-batch shape (64 envs): where does the Python time per minibatch pass go?"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
l = MagpoLearner(CoordSumConfig(8, 15, 100, 100), 64, SystemConfig(ppo_epochs=15, num_minibatches=8), "cuda", net_seed=0, n_block=2)
l.setup(host_split(prng_key(1), 4)[0])
for _ in range(2):
    l.update_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
l.update_step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
