"""In-kernel phase timing of k_wgrad (debug build: MAGPO_EXTRA_FLAGS=-DMAGPO_WG_PROF python -m magpo_amd.build --force)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magpo_amd._lib import lib
L = lib(); dev = 'cuda'
R = 8192 * 128 * 4
st = torch.cuda.current_stream().cuda_stream
fn = L.raw("magpo_debug_wg_prof"); out = np.zeros(8, dtype=np.uint64)
for KIN, NOUT in ((128, 384), (64, 256)):
    X = torch.randn(R, KIN, device=dev); dY = torch.randn(R, NOUT, device=dev); dW = torch.empty(KIN, NOUT, device=dev); G = 512
    ws = torch.empty(L.call("magpo_wgrad_workspace_floats", KIN, NOUT, G), device=dev)
    L.call("magpo_wgrad", X, KIN, dY, NOUT, R, KIN, KIN, NOUT, dW, None, ws, G, 1.0, 0, 0, st); torch.cuda.synchronize()
    fn(ctypes.c_void_p(out.ctypes.data), 1)
    L.call("magpo_wgrad", X, KIN, dY, NOUT, R, KIN, KIN, NOUT, dW, None, ws, G, 1.0, 0, 0, st); torch.cuda.synchronize()
    fn(ctypes.c_void_p(out.ctypes.data), 1)
    tot = float(out[:4].sum()); ntile = (R + 63) // 64
    print(f"wgrad {KIN}x{NOUT}: cycles/tile/block {tot / 16 / (ntile / 512):.0f} (MFMA-only 4096): wait+barrier {out[0]/tot:.2f} stash+barrier {out[1]/tot:.2f} fetch-issue {out[2]/tot:.2f} mfma-loop {out[3]/tot:.2f}")
