"""In-kernel stage timing of k_sable_act (debug build: MAGPO_EXTRA_FLAGS=-DMAGPO_ACT_PROF python -m magpo_amd.build --force)."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import magpo_amd._lib as _libmod
if os.environ.get("MAGPO_LIB"):
    _libmod.LIB_PATH = os.path.abspath(os.environ["MAGPO_LIB"])
from magpo_amd.learner import CoordSumConfig, MagpoLearner, SystemConfig, host_split, prng_key
from magpo_amd._lib import lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dl = MagpoLearner(CoordSumConfig(4, 20, 100, 60), N, SystemConfig(), 'cuda', net_seed=0)
dl.use_graph = False
dl.setup(host_split(prng_key(42), 4)[0])
dl.rollout(); torch.cuda.synchronize()
fn = lib().raw("magpo_debug_act_prof"); out = np.zeros(32, dtype=np.uint64)
fn(ctypes.c_void_p(out.ctypes.data), 1)
t0 = time.time(); dl.rollout(); torch.cuda.synchronize(); t1 = time.time()
fn(ctypes.c_void_p(out.ctypes.data), 1)
calls = 129
EPW = lib().call("magpo_sable_act_envs_per_wave", N, 4, dl.tuning.act_envs_per_wave); nwg = len(range(0, (N + EPW - 1) // EPW, 64))
us = lambda k: float(out[k]) / 100 / nwg / calls
names = {0: "enc embed+qkvg", 1: "enc state pass", 2: "enc wo/norm/value/q2", 19: "pre-pass cross states", 4: "pre-pass candidates", 5: "dec embed+qkvg1+self",
         6: "dec wo1+kvg2+cross", 7: "dec wo2+head", 16: "sampling", 17: "dec qkvg1 (blocks>0)", 18: "dec state pass (blocks>0)", 3: "-"}
print(f"rollout {1e3*(t1-t0):.1f} ms, {EPW} envs per wave; per launch and wave, us:")
for k in (0, 1, 2, 19, 4, 5, 6, 7, 16, 17, 18):
    print(f"  {names[k]:28s} {us(k):7.1f}")
print(f"  {'total':28s} {sum(us(k) for k in names):7.1f}")
tot = float(sum(out[8:13])) or 1.0
print("state-pass sub-stages (share of cycles): prefetch-issue %.2f  stage-tokens %.2f  decay+update %.2f  store %.2f  output %.2f" % (
    out[8]/tot, out[9]/tot, out[10]/tot, out[11]/tot, out[12]/tot))
