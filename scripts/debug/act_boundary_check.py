"""Fused vs composed acting path over several rollouts WITHOUT episode ends (the carried decoder states are non-zero at every rollout
boundary): max differences of values / carried states per rollout.  Guards the deferred-update protocol of k_sable_act at rollout seams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
ls = []
for fused in (False, True):
    l = MagpoLearner(CoordSumConfig(3, 10, 100, 30), 8, SystemConfig(rollout_length=6, ppo_epochs=1, num_minibatches=1), "cuda", net_seed=9, wgrad_groups=4, n_block=2)
    l.fused_act, l.use_graph = fused, False
    l.setup(host_split(prng_key(11), 4)[0])
    ls.append(l)
for it in range(4):
    for l in ls:
        l.rollout(); l._carry_over()
    a, b = ls
    print(it, "actions equal", bool(torch.equal(a.traj["action"], b.traj["action"])), "dvalue %.2e" % float((a.traj["value"] - b.traj["value"]).abs().max()),
          "dstate %.2e (scale %.2e)" % (max(float((x - y).abs().max()) for x, y in zip(a.sable_hs, b.sable_hs)), max(float(x.abs().max()) for x in a.sable_hs)),
          "done any", bool(a.traj["done"].any()))
