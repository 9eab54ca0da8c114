"""Acting-kernel time per launch over num_envs x envs-per-wave (Tuning.act_envs_per_wave): which wave shape should magpo_sable_act pick?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
for A, K, mv, nb in ((4, 20, 60, 1), (8, 15, 100, 2)):
    for N in (64, 256, 1024, 2048, 4096, 8192, 16384):
        l = MagpoLearner(CoordSumConfig(A, K, 100, mv), N, SystemConfig(rollout_length=8, ppo_epochs=1, num_minibatches=1), "cuda", net_seed=0, n_block=nb)
        l.use_graph = False
        l.setup(host_split(prng_key(1), 4)[0])
        res = []
        for epw in (4, 8, 16):
            l.tuning.act_envs_per_wave = epw   # per-call argument of magpo_sable_act (dims[11])
            g = l.groups[0]; tr = g.traj
            l._rollout_keys(g)
            def call():
                l.guider.act_fused(tr["obs"][0], tr["step_count"][0], g.sable_hs, g.skeys_host[0], tr["action"][0], tr["log_prob"][0], tr["value"][0], done=tr["done"][0])
            for _ in range(3): call()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(20): call()
            torch.cuda.synchronize()
            res.append((time.time() - t0) / 20 * 1e6)
        print(f"A={A} nb={nb} N={N:6d}  us per launch: EPW4 {res[0]:8.1f}  EPW8 {res[1]:8.1f}  EPW16 {res[2]:8.1f}")
        del l
        torch.cuda.empty_cache()
