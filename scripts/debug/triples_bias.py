"""Signed error of the GRU forward scan / dense kernel against fp64 in the fp32-MFMA and bf16-triple modes: is there a BIAS (mean signed error,
error correlated with the value) that |error| statistics would not show?  (Follow-up of the learning check in profiles/r03_sweep_return_at_10M.md.)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd._lib import lib
L = lib(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
H = 128
def tp(W):   # W [K, N] -> W^T padded rows
    K, N = W.shape; npad = (N + 31) // 32 * 32
    t = torch.zeros(npad, K, device=dev); L.call("magpo_transpose_pad", W.to(dev), t, K, N, npad, st); return t
# dense 128 -> 128 with relu (the actor's post torso), inputs like GRU states
R = 1 << 16
X = torch.tanh(torch.randn(R, H, generator=g)); W = torch.randn(H, H, generator=g) / math.sqrt(H); b = torch.randn(H, generator=g) * 0.1
ref = torch.relu(X.double() @ W.double() + b.double())
for variant in (0, 4):
    Y = torch.zeros(R, H, device=dev)
    L.call("magpo_linear", X.to(dev), H, tp(W), b.to(dev), Y, H, None, R, H, H, 1, variant, st)
    e = Y.cpu().double() - ref
    m = ref > 0
    print(f"dense variant {variant}: mean signed err {e[m].mean():+.3e}  mean |err| {e[m].abs().mean():.3e}  corr(err, y) {torch.corrcoef(torch.stack([e[m], ref[m]]))[0,1]:+.4f}  mean(err * sign(y - median)) {(e[m] * torch.sign(ref[m] - ref[m].median())).mean():+.3e}")
# GRU forward scan
nseq, T, A = 96, 128, 4; Rr = nseq * T * A
xi = torch.randn(Rr, 3 * H, generator=g) * 0.7; Wh = torch.randn(H, 3 * H, generator=g) * 0.09; bhn = torch.randn(H, generator=g) * 0.1
h0 = torch.randn(nseq * A, H, generator=g) * 0.3; done = torch.rand(nseq, T, generator=g) < 0.02
x = xi.double().reshape(nseq, T, A, 3 * H); h = h0.double().reshape(nseq, A, H); refs = []
for t in range(T):
    h = torch.where(done[:, t][:, None, None], torch.zeros_like(h), h)
    hh = h @ Wh.double()
    r = torch.sigmoid(x[:, t, :, :H] + hh[..., :H]); z = torch.sigmoid(x[:, t, :, H:2 * H] + hh[..., H:2 * H])
    n = torch.tanh(x[:, t, :, 2 * H:] + r * (hh[..., 2 * H:] + bhn.double())); h = (1 - z) * n + z * h
    refs.append(h)
ref = torch.stack(refs, 1).reshape(Rr, H)
for mode in (0, 2):
    hs = torch.empty(Rr, H, device=dev); gates = torch.empty(Rr, 4 * H, device=dev); hp = torch.empty(Rr, H, device=dev)
    L.call("magpo_gru_scan_fwd", xi.to(dev), tp(Wh), bhn.to(dev), h0.to(dev), None, done.to(torch.uint8).to(dev), hs, gates, hp, nseq, T, A, None, mode, 0, st)
    e = hs.cpu().double() - ref
    print(f"GRU scan mode {mode}: mean signed err {e.mean():+.3e}  mean |err| {e.abs().mean():.3e}  corr(err, h) {torch.corrcoef(torch.stack([e.reshape(-1), ref.reshape(-1)]))[0,1]:+.4f}  mean(err * sign(h)) {(e * torch.sign(ref)).mean():+.3e}")
