import os
"""In-kernel phase timing of k_ret_chunk_bwd (debug build: MAGPO_EXTRA_FLAGS=-DMAGPO_RET_PROF python -m magpo_amd.build --force)."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magpo_amd import _lib
if os.environ.get("MAGPO_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MAGPO_LIB"])
from magpo_amd._lib import lib
L = lib()
CT, SPLIT = int(os.environ.get("MAGPO_RET_CHUNK", 0)), int(os.environ.get("MAGPO_GRU_SPLIT_BF16", 0))   # per-call tuning arguments (the library keeps no state)
nseq, T, A = 8192, 128, 4
R = nseq * T * A
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, dr = (torch.randn(R, 64, device=dev, generator=g) * 0.3 for _ in range(4))
dq, dk, dv, r = (torch.empty(R, 64, device=dev) for _ in range(4))
dones = (torch.rand(nseq, T, device=dev, generator=g) < 0.01).to(torch.uint8)
nch = L.call("magpo_retention_num_chunks", T, A, CT)
states = torch.empty(nseq, nch, 64, 64, device=dev)
s0 = torch.zeros(nseq, 64, 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.call("magpo_retention_chunk_fwd", q, 64, k, 64, v, 64, r, 64, s0, None, dones, states, None, nseq, T, A, 1, 0.95, 64, None, CT, st)
fn = L.raw("magpo_debug_ret_prof"); out = np.zeros(8, dtype=np.uint64)
for it in range(2):
    L.call("magpo_retention_chunk_bwd", q, 64, k, 64, v, 64, dr, 64, dq, 64, dk, 64, dv, 64, dones, states, nseq, T, A, 1, 0.95, 64, None, CT, st)
torch.cuda.synchronize(); fn(ctypes.c_void_p(out.ctypes.data), 1)
t0 = time.time()
for it in range(5):
    L.call("magpo_retention_chunk_bwd", q, 64, k, 64, v, 64, dr, 64, dq, 64, dk, 64, dv, 64, dones, states, nseq, T, A, 1, 0.95, 64, None, CT, st)
torch.cuda.synchronize(); t1 = time.time()
fn(ctypes.c_void_p(out.ctypes.data), 1)
tot = float(out[:6].sum())
names = ["barrier+meta", "P/dP+barrier", "dQ", "dK", "dV", "G+barrier+stash"]
nwg = len(range(0, nseq, 64)); per = tot / nwg / 5 / nch
print(f"bwd {1e3*(t1-t0)/5:.2f} ms/launch; clock ticks per chunk {per:.0f} (MFMA per wave: 6656 cycles per 32-token chunk, 18432 per 64-token chunk): " + "  ".join(f"{n} {out[i]/tot:.2f}" for i, n in enumerate(names)))
