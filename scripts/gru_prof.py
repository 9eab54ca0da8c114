import os
"""In-kernel phase timing of k_gru_scan_fwd (debug build: MAGPO_EXTRA_FLAGS=-DMAGPO_GRU_PROF python -m magpo_amd.build --force)."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magpo_amd._lib import lib
L = lib()
CT, SPLIT = int(os.environ.get("MAGPO_RET_CHUNK", 0)), int(os.environ.get("MAGPO_GRU_SPLIT_BF16", 0))   # per-call tuning arguments (the library keeps no state)
nseq, T, A, H = 8192, 128, 4, 128
R = nseq * T * A
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
xi = torch.randn(R, 3 * H, device=dev, generator=g) * 0.3
Wht = torch.randn(3 * H, H, device=dev, generator=g) * 0.05
bhn = torch.zeros(H, device=dev); h0 = torch.zeros(nseq * A, H, device=dev)
reset = (torch.rand(nseq, T, device=dev, generator=g) < 0.01).to(torch.uint8)
hs = torch.empty(R, H, device=dev); gates = torch.empty(R, 4 * H, device=dev); hprev = torch.empty(R, H, device=dev)
st = torch.cuda.current_stream().cuda_stream
call = lambda: L.call("magpo_gru_scan_fwd", xi, Wht, bhn, h0, None, reset, hs, gates, hprev, nseq, T, A, None, SPLIT, 0, st)
fn = L.raw("magpo_debug_gru_prof"); out = np.zeros(8, dtype=np.uint64)
call(); call(); torch.cuda.synchronize(); fn(ctypes.c_void_p(out.ctypes.data), 1)
t0 = time.time()
for it in range(5): call()
torch.cuda.synchronize(); t1 = time.time()
fn(ctypes.c_void_p(out.ctypes.data), 1)
tot = float(out.sum())
names = ["ld0", "mfma0", "elt0", "ld1", "mfma1", "elt1", "gap", "barrier"]
nwg = len(range(0, nseq * A // 64, 64)); per = tot / nwg / 5 / T
print(f"fwd {1e3*(t1-t0)/5:.2f} ms/launch; s_memtime ticks per step {per:.0f}: " + "  ".join(f"{n} {out[i]/tot:.2f}" for i, n in enumerate(names)))
