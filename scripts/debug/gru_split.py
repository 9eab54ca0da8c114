"""A/B of the GRU training scan: fp32 MFMA vs split-bf16 x3 (accuracy against an fp64 reference + time at the bench shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd._lib import lib
L = lib(); st = torch.cuda.current_stream().cuda_stream
H = 128
def run(nseq, T, A, reps):
    g = torch.Generator().manual_seed(0)
    R = nseq * T * A
    xi = (torch.randn(R, 3 * H, generator=g) * 0.7).cuda()
    Wh = torch.randn(H, 3 * H, generator=g) * 0.09
    Wht = Wh.t().contiguous().cuda()
    bhn = (torch.randn(H, generator=g) * 0.1).cuda()
    h0 = (torch.randn(nseq * A, H, generator=g) * 0.3).cuda()
    reset = (torch.rand(nseq, T, generator=g) < 0.02).to(torch.uint8).cuda()
    out = {}
    for mode in (0, 1):
        SPLIT = mode
        hs = torch.empty(R, H, device="cuda"); gates = torch.empty(R, 4 * H, device="cuda"); hp = torch.empty(R, H, device="cuda")
        L.call("magpo_gru_scan_fwd", xi, Wht, bhn, h0, None, reset, hs, gates, hp, nseq, T, A, None, SPLIT, 0, st)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(reps):
            L.call("magpo_gru_scan_fwd", xi, Wht, bhn, h0, None, reset, hs, gates, hp, nseq, T, A, None, SPLIT, 0, st)
        torch.cuda.synchronize()
        out[mode] = (hs.clone(), gates.clone(), (time.time() - t0) / reps * 1e3)
    return xi, Wh, bhn, h0, reset, out
# accuracy at a small shape against fp64
nseq, T, A = 16, 128, 4
xi, Wh, bhn, h0, reset, out = run(nseq, T, A, 1)
x = xi.cpu().double().view(nseq, T, A, 3 * H); W = Wh.double(); b = bhn.cpu().double()
h = h0.cpu().double().view(nseq, A, H); ref = []
for t in range(T):
    h = torch.where(reset.cpu()[:, t].bool()[:, None, None], torch.zeros_like(h), h)
    hh = h @ W
    r = torch.sigmoid(x[:, t, :, :H] + hh[..., :H]); z = torch.sigmoid(x[:, t, :, H:2 * H] + hh[..., H:2 * H])
    n = torch.tanh(x[:, t, :, 2 * H:] + r * (hh[..., 2 * H:] + b))
    h = (1 - z) * n + z * h
    ref.append(h)
ref = torch.stack(ref, 1).reshape(-1, H)
for mode in (0, 1):
    e = (out[mode][0].cpu().double() - ref).abs()
    print("mode", mode, "hs max err %.3e mean %.3e (scale %.2f)" % (e.max(), e.mean(), ref.abs().max()))
_, _, _, _, _, out = run(8192, 128, 4, 5)
print("bench shape: fp32 %.2f ms, split-bf16 %.2f ms" % (out[0][2], out[1][2]), "max diff", (out[0][0] - out[1][0]).abs().max().item())
