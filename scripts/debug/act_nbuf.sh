#!/bin/bash
# Acting kernel at 16 envs per wave with 2 / 3 / 4 state buffers in flight (registers vs prefetch depth); restores the normal build.
set -e
cat > /tmp/act_time.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
for A, K, mv, nb, N in ((4, 20, 60, 1, 16384), (4, 20, 60, 1, 4096), (8, 15, 100, 2, 16384)):
    l = MagpoLearner(CoordSumConfig(A, K, 100, mv), N, SystemConfig(rollout_length=8, ppo_epochs=1, num_minibatches=1), "cuda", net_seed=0, n_block=nb)
    l.use_graph = False
    l.setup(host_split(prng_key(1), 4)[0])
    os.environ["MAGPO_ACT_EPW"] = "16"
    g = l.groups[0]; tr = g.traj
    l._rollout_keys(g)
    def call():
        l.guider.act_fused(tr["obs"][0], tr["step_count"][0], g.sable_hs, g.skeys_host[0], tr["action"][0], tr["log_prob"][0], tr["value"][0], done=tr["done"][0])
    for _ in range(10): call()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(30): call()
    torch.cuda.synchronize()
    print(f"   A={A} nb={nb} N={N}: {(time.time() - t0) / 30 * 1e6:.1f} us per launch", flush=True)
    del l; torch.cuda.empty_cache()
PY
for nb in 2 3 4; do
  touch magpo_amd/csrc/act_fused.hip
  MAGPO_EXTRA_FLAGS="-DMAGPO_ACT_NBUF16=$nb" python -m magpo_amd.build > /dev/null
  echo "NBUF16=$nb"; python /tmp/act_time.py
done
touch magpo_amd/csrc/act_fused.hip
python -m magpo_amd.build > /dev/null
