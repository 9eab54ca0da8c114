"""Launch time of the chunkwise retention kernels at the bench shape (8192 sequences x 128 steps x 4 agents), with q | k | v read per token
row and through a row table (csrc/classtab.hip), under both chunk sizes.  MAGPO_LIB selects the library build (A/B runs);
RET_ONLY32=1 times the 32-token kernels only (for --pmc passes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magpo_amd import _lib
if os.environ.get("MAGPO_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MAGPO_LIB"])
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
NC = 30000                                                  # distinct input rows of the class table at the bench shape (order of magnitude)
tab = torch.randn(NC, 256, device=dev, generator=g) * 0.3
rows = torch.randint(0, NC, (R,), device=dev, generator=g).to(torch.int32)
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("RET_REPS", 10))
for ct in ((32,) if os.environ.get("RET_ONLY32") else (64, 32)):
    CT = ct
    nch = L.call("magpo_retention_num_chunks", T, A, CT)
    states = torch.empty(nseq, nch, 64, 64, device=dev)
    for what, src, ridx in (("rows", buf, None), ("table", tab, rows)):
        fwd = lambda: L.call("magpo_retention_chunk_fwd", src, 256, src[:, 64:], 256, src[:, 128:], 256, r, 64, s0, None, dones, states, None, nseq, T, A, 1, 0.775, 64, ridx, CT, st)
        bwd = lambda: L.call("magpo_retention_chunk_bwd", src, 256, src[:, 64:], 256, src[:, 128:], 256, dr, 64, dbuf, 256, dbuf[:, 64:], 256, dbuf[:, 128:], 256,
                             dones, states, nseq, T, A, 1, 0.775, 64, ridx, CT, st)
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            fn(); fn(); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            print(f"chunk {ct:2d} tokens ({nch:2d} chunks), q|k|v by {what:5s}: {name} {1e3 * (time.time() - t0) / reps:.3f} ms per launch", flush=True)
