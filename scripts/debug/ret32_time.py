"""Launch time of the chunkwise retention kernels at the bench shape under both chunk sizes (64-token / 32-token tiles)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd._lib import lib
L = lib()
nseq, T, A = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = nseq * T * A
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
buf = torch.randn(R, 256, device=dev, generator=g) * 0.3
dr = torch.randn(R, 64, device=dev, generator=g) * 0.3
r = torch.empty(R, 64, device=dev); dbuf = torch.empty(R, 256, device=dev)
s0 = torch.randn(nseq, 64, 64, device=dev, generator=g) * 0.1
dones = (torch.rand(nseq, T, device=dev, generator=g) < 0.01).to(torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for ct in (64, 32):
    CT = ct
    nch = L.call("magpo_retention_num_chunks", T, A, CT)
    states = torch.empty(nseq, nch, 64, 64, device=dev)
    fwd = lambda: L.call("magpo_retention_chunk_fwd", buf, 256, buf[:, 64:], 256, buf[:, 128:], 256, r, 64, s0, None, dones, states, None, nseq, T, A, 1, 0.775, 64, None, CT, st)
    bwd = lambda: L.call("magpo_retention_chunk_bwd", buf, 256, buf[:, 64:], 256, buf[:, 128:], 256, dr, 64, dbuf, 256, dbuf[:, 64:], 256, dbuf[:, 128:], 256,
                         dones, states, nseq, T, A, 1, 0.775, 64, None, CT, st)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print(f"chunk {ct:2d} tokens ({nch:2d} chunks): {name} {1e3 * (time.time() - t0) / 10:.3f} ms per launch")
