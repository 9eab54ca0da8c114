import os
"""Per-kernel micro-benchmarks at the bench shapes (R = 8192 seqs x 128 steps x 4 agents rows), HIP-event timed.
usage: python scripts/kbench.py [wgrad] [linear] [ret] [gru] [rows] [loss]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magpo_amd import _lib
if os.environ.get("MAGPO_LIB"):   # experiment build (A/B runs)
    _lib.LIB_PATH = os.path.abspath(os.environ["MAGPO_LIB"])
from magpo_amd._lib import lib
L = lib(); dev = 'cuda'
CT, SPLIT = int(os.environ.get("MAGPO_RET_CHUNK", 0)), int(os.environ.get("MAGPO_GRU_SPLIT_BF16", 0))   # per-call tuning arguments (the library keeps no state)
which = set(sys.argv[1:]) or {"wgrad", "linear", "ret", "gru"}
nseq, T, A = 8192, 128, 4
R = nseq * T * A
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)

def timeit(name, fn, flops=0, bytes_=0, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} {ms:8.3f} ms  {flops/ms/1e9:7.1f} TFLOP/s  {bytes_/ms/1e6:7.0f} GB/s", flush=True)

if "wgrad" in which:
    for KIN, NOUT in ((64, 256), (64, 64), (64, 192), (128, 384), (128, 128)):
        X = torch.randn(R, KIN, device=dev, generator=g); dY = torch.randn(R, NOUT, device=dev, generator=g)
        dW = torch.empty(KIN, NOUT, device=dev); G = 512
        ws = torch.empty(L.call("magpo_wgrad_workspace_floats", KIN, NOUT, G), device=dev)
        timeit(f"wgrad {KIN}x{NOUT}", lambda: L.call("magpo_wgrad", X, KIN, dY, NOUT, R, KIN, KIN, NOUT, dW, None, ws, G, 1.0, 0, int(os.environ.get("MAGPO_WGRAD_VARIANT", 0)), st),
               2.0 * R * KIN * NOUT, 4.0 * R * (KIN + NOUT))
if "linear" in which:
    for KIN, NOUT in ((64, 256), (64, 64), (64, 192), (128, 384), (128, 128), (256, 64), (384, 128), (192, 64)):
        X = torch.randn(R, KIN, device=dev, generator=g); Y = torch.empty(R, NOUT, device=dev)
        npad = (NOUT + 31) // 32 * 32
        Wt = torch.randn(npad, KIN, device=dev, generator=g)
        timeit(f"linear {KIN}->{NOUT}", lambda: L.call("magpo_linear", X, KIN, Wt, None, Y, NOUT, None, R, KIN, NOUT, 0, int(os.environ.get("MAGPO_LINEAR_VARIANT", 0)), st),
               2.0 * R * KIN * NOUT, 4.0 * R * (KIN + NOUT))
if "ret" in which:
    q, k, v, dr = (torch.randn(R, 64, device=dev, generator=g) * 0.3 for _ in range(4))
    dq, dk, dv, r = (torch.empty(R, 64, device=dev) for _ in range(4))
    dones = (torch.rand(nseq, T, device=dev, generator=g) < 0.01).to(torch.uint8)
    nch = L.call("magpo_retention_num_chunks", T, A, CT)
    states = torch.empty(nseq, nch, 64, 64, device=dev); s0 = torch.zeros(nseq, 64, 64, device=dev)
    timeit("ret_chunk_fwd", lambda: L.call("magpo_retention_chunk_fwd", q, 64, k, 64, v, 64, r, 64, s0, None, dones, states, None, nseq, T, A, 1, 0.95, 64, None, CT, st),
           nseq * nch * 4 * 2.0 * 64 ** 3, 0)
    timeit("ret_chunk_bwd", lambda: L.call("magpo_retention_chunk_bwd", q, 64, k, 64, v, 64, dr, 64, dq, 64, dk, 64, dv, 64, dones, states, nseq, T, A, 1, 0.95, 64, None, CT, st),
           nseq * nch * 9 * 2.0 * 64 ** 3, 0)
if "gru" in which:
    H = 128
    xi = torch.randn(R, 3 * H, device=dev, generator=g) * 0.3
    Wht = torch.randn(3 * H, H, device=dev, generator=g) * 0.05; Wh = Wht.t().contiguous()
    bhn = torch.zeros(H, device=dev); h0 = torch.zeros(nseq * A, H, device=dev)
    reset = (torch.rand(nseq, T, device=dev, generator=g) < 0.01).to(torch.uint8)
    hs = torch.empty(R, H, device=dev); gates = torch.empty(R, 4 * H, device=dev); hprev = torch.empty(R, H, device=dev)
    timeit("gru_scan_fwd", lambda: L.call("magpo_gru_scan_fwd", xi, Wht, bhn, h0, None, reset, hs, gates, hprev, nseq, T, A, None, SPLIT, 0, st), 2.0 * R * H * 3 * H, 0)
    dhs = torch.randn(R, H, device=dev, generator=g) * 0.1
    dg = torch.empty(R, 4 * H, device=dev); slab = torch.empty((nseq * A + 63) // 64, H, device=dev)
    timeit("gru_scan_bwd", lambda: L.call("magpo_gru_scan_bwd", gates, hprev, reset, dhs, Wh, dg, slab, nseq, T, A, SPLIT, 0, st), 2.0 * R * H * 3 * H, 0)
if "rows" in which:
    F, K = 5, 20
    obs = torch.randn(R, F, device=dev, generator=g)
    W = torch.randn(F, 128, device=dev, generator=g); b = torch.zeros(128, device=dev); Y = torch.empty(R, 128, device=dev)
    timeit("small_linear 5->128", lambda: L.call("magpo_small_linear", obs, F, F, W, b, Y, 128, 128, R, 1, st), 0, 4.0 * R * (F + 128))
    pe = torch.empty(101, 64, device=dev); L.call("magpo_pe_table", pe, 101, 64, st)
    sobs = torch.ones(F, device=dev); Wo = torch.randn(F, 64, device=dev, generator=g); sln = torch.ones(64, device=dev)
    pos = torch.randint(0, 100, (nseq * T,), device=dev, generator=g, dtype=torch.int32).repeat_interleave(A)
    idx = torch.randint(0, K + 1, (R,), device=dev, generator=g, dtype=torch.int32); Wa = torch.randn(K + 1, 64, device=dev, generator=g)
    xn = torch.empty(R, 64, device=dev); kin = torch.empty(R, 64, device=dev)
    timeit("embed_fwd obs", lambda: L.call("magpo_embed_fwd", 0, obs, F, F, sobs, Wo, None, 0, sln, pe, pos, 1, 101, None, 64, xn, 64, kin, 64, R, st),
           0, 4.0 * R * (F + 1 + 128))
    timeit("embed_fwd act", lambda: L.call("magpo_embed_fwd", 1, None, 0, 0, None, Wa, idx, 1, sln, pe, pos, 1, 101, None, 64, xn, 64, kin, 64, R, st),
           0, 4.0 * R * (2 + 128))
    ws = torch.empty(8 * 1024, device=dev, dtype=torch.float64); out = torch.empty(2, device=dev); adv = torch.randn(R, device=dev, generator=g)
    timeit("adv_moments", lambda: L.call("magpo_adv_moments", adv, R, ws, out, st), 0, 4.0 * R)
    timeit("torch fill (write-only ref)", lambda: Y.fill_(1.0), 0, 4.0 * R * 128)
    Y2 = torch.empty_like(Y)
    timeit("torch copy (read+write ref)", lambda: Y2.copy_(Y), 0, 8.0 * R * 128)
if "loss" in which:
    K = 20
    gl = torch.randn(R, 64, device=dev, generator=g); al = torch.randn(R, 64, device=dev, generator=g)
    action = torch.randint(0, K, (R,), device=dev, generator=g, dtype=torch.int32)
    old = torch.randn(R, device=dev, generator=g) * 0.1 - 3.0; vold = torch.randn(R, device=dev, generator=g); val = vold + 0.1 * torch.randn(R, device=dev, generator=g)
    adv = torch.randn(R, device=dev, generator=g); tgt = torch.randn(R, device=dev, generator=g)
    stats = torch.tensor([0.0, 1.0], device=dev); ws = torch.empty(8 * 1024, device=dev, dtype=torch.float64); lo = torch.empty(9, device=dev)
    dg = torch.empty(R, 64, device=dev); da = torch.empty(R, 64, device=dev); dv = torch.empty(R, device=dev)
    timeit("loss_fwd_bwd (K=20, 64-wide rows)", lambda: L.call("magpo_loss_fwd_bwd", gl, 64, al, 64, None, action, old, vold, val, adv, tgt, stats, dg, 64, da, 64, dv, ws, lo,
           R, K, 0.2, 3.0, 0.01, 0.5, 1.0, st), 0, 4.0 * R * (4 * 64 + 8))
