"""Acting kernel, 4 vs 16 envs per wave (Tuning.act_envs_per_wave) at small and medium batches (A = 4, one block)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
for N in [int(x) for x in sys.argv[1:]] or (256, 1024, 2048, 4096, 8192):
    l = MagpoLearner(CoordSumConfig(4, 20, 100, 60), N, SystemConfig(rollout_length=8, ppo_epochs=1, num_minibatches=1), "cuda", net_seed=0)
    l.use_graph = False
    l.setup(host_split(prng_key(1), 4)[0])
    g = l.groups[0]; tr = g.traj
    l._rollout_keys(g)
    res = []
    for epw in (4, 8, 16):
        l.tuning.act_envs_per_wave = epw   # per-call argument of magpo_sable_act (dims[11])
        def call():
            l.guider.act_fused(tr["obs"][0], tr["step_count"][0], g.sable_hs, g.skeys_host[0], tr["action"][0], tr["log_prob"][0], tr["value"][0], done=tr["done"][0])
        for _ in range(10): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # device time: a host-side pause (GC) does not count
        e0.record()
        for _ in range(30): call()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"N={N:6d}  us per launch: EPW4 {res[0]:8.1f}  EPW8 {res[1]:8.1f}  EPW16 {res[2]:8.1f}", flush=True)
    del l; torch.cuda.empty_cache()
