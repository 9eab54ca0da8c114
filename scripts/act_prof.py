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
fn = lib().raw("magpo_debug_act_prof"); out = np.zeros(16, dtype=np.uint64)
fn(ctypes.c_void_p(out.ctypes.data), 1)
t0 = time.time(); dl.rollout(); torch.cuda.synchronize(); t1 = time.time()
fn(ctypes.c_void_p(out.ctypes.data), 1)
nwg = len(range(0, (N + 31) // 32, 64)); calls = 129
EPW = int(os.environ.get('EPW', '16')); nwg = len(range(0, (N + EPW - 1) // EPW, 64))
print(f"rollout {1e3*(t1-t0):.1f} ms; per-launch per-wave us: dense/rows {out[0]/100/nwg/calls:.1f} ret {out[1]/100/nwg/calls:.1f} sample {out[2]/100/nwg/calls:.1f}")
tot = float(sum(out[8:13])) or 1.0
print("ret sub-stages (share of cycles): prefetch-issue %.2f  stage-tokens %.2f  decay+update %.2f  store %.2f  output %.2f ; cycles/pair %.0f" % (
    out[8]/tot, out[9]/tot, out[10]/tot, out[11]/tot, out[12]/tot, tot/nwg/calls/(9*EPW)))
