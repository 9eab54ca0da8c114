"""Acting kernel alone at a given shape: us per launch in rollout mode (pending decoder-state rows, no flush), by device events.
usage: python scripts/debug/act_time.py [N] [A] [n_block] [K] [launches]     (MAGPO_LIB=path selects an experiment build, MAGPO_ACT_EPW the wave shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import magpo_amd._lib as _lib
if os.environ.get("MAGPO_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MAGPO_LIB"])
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
A = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
l = MagpoLearner(CoordSumConfig(A, K, 100, 60), N, SystemConfig(rollout_length=4, ppo_epochs=1, num_minibatches=1), "cuda", net_seed=0, n_block=nb)
l.use_graph = False
l.setup(host_split(prng_key(1), 4)[0])
g = l.groups[0]; tr = g.traj
l._rollout_keys(g)
def call(pending=True):   # rollout mode: pending rows of the previous launch, deferred candidate pass (precand / defer), no flush
    l.guider.act_fused(tr["obs"][0], tr["step_count"][0], g.sable_hs, g.skeys_host[0], tr["action"][0], tr["log_prob"][0], tr["value"][0],
                       done=tr["done"][0], pending=pending, flush=False, precand=pending, defer=True)
call(False)
for _ in range(3): call()
torch.cuda.synchronize()
# Device events around a burst of eager launches also see the HOST when it falls behind (a Python GC pause of ~75 ms inside a 40-launch
# window reads as +1.9 ms per launch: the 2 333.8 us "outlier" of profiles/r03_act_kernel_launch_times.txt:6).  So: GC off, several
# windows, min and median reported; a window far above the median is host time, not kernel time.
import gc
gc.disable()
wins = []
for w in range(int(os.environ.get("WINDOWS", 7))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    wins.append(e0.elapsed_time(e1) / reps * 1e3)
wins.sort()
us, med = wins[0], wins[len(wins) // 2]
alg = 6.0 * 16384 * nb * N   # bytes: three 16 KiB states per block read + written once per env step
epw = _lib.lib().call("magpo_sable_act_envs_per_wave", N, A, l.tuning.act_envs_per_wave)
print(f"N={N} A={A} nb={nb} K={K} epw={epw}{'' if l.tuning.act_envs_per_wave else ' (auto)'} lib={os.path.basename(_lib.LIB_PATH)}: min {us:8.1f} us per launch "
      f"(median {med:.1f}, max {wins[-1]:.1f} over {len(wins)} windows of {reps}), {alg / us / 1e6:7.3f} TB/s algorithmic ({alg / us / 1e6 / 8:.3f} of 8 TB/s)")
