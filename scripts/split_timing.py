"""Wall-clock split of one update step into rollout and training (synchronised; diagnostic only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dl = MagpoLearner(CoordSumConfig(4, 20, 100, 60), N, SystemConfig(), 'cuda', net_seed=0)
dl.fused_act = os.environ.get('MAGPO_FUSED_ACT', '1') == '1'
dl.batched_actor_carry = os.environ.get('MAGPO_BATCHED_CARRY', '1') == '1'
if os.environ.get('MAGPO_FUSED_SEG') is not None: dl.guider.fused_segments = os.environ['MAGPO_FUSED_SEG'] == '1'
dl.setup(host_split(prng_key(42), 4)[0])
dl.update_step(); torch.cuda.synchronize()
for it in range(2):
    t0 = time.time(); dl.rollout(); torch.cuda.synchronize(); t1 = time.time()
    dl.update(); torch.cuda.synchronize(); t2 = time.time(); dl._carry_over()
    print(f"N={N} rollout {1e3*(t1-t0):.1f} ms  update {1e3*(t2-t1):.1f} ms  -> {N*128/(t2-t0):.0f} env-steps/s", flush=True)
