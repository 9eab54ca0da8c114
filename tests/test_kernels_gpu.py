"""GPU parity of every primitive HIP kernel against the CPU oracle / torch fp64 (through the C ABI)."""
import math

import numpy as np
import pytest
import torch

from oracle import coordsum as ocs
from oracle import learner as olearn
from oracle import networks as onets
from oracle import prng as oprng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(x, dtype=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV).contiguous()


def close(a, b, rtol=2e-5, atol=2e-6, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} vs ref scale {ref:.3e}"


def transpose_pad(L, st, W):
    K, N = W.shape
    Np = (N + 31) // 32 * 32
    Wt = torch.empty(Np, K, device=DEV)
    L.call("magpo_transpose_pad", W, Wt, K, N, Np, st)
    return Wt


@pytest.mark.parametrize("KIN,NOUT,R,act", [(64, 64, 200, 0), (64, 256, 130, 2), (128, 384, 64, 0), (256, 64, 77, 0),
                                             (64, 20, 100, 0), (128, 128, 300, 1), (192, 64, 65, 0), (384, 128, 70, 0)])
def test_linear(L, stream, KIN, NOUT, R, act):
    g = torch.Generator().manual_seed(1)
    X = torch.randn(R, KIN, generator=g)
    W = torch.randn(KIN, NOUT, generator=g) / math.sqrt(KIN)
    b = torch.randn(NOUT, generator=g)
    Xd, Wd, bd = dev(X), dev(W), dev(b)
    Wt = transpose_pad(L, stream, Wd)
    close(Wt[:NOUT], W.T, 0, 0, "transpose")
    ld = (NOUT + 3) // 4 * 4
    Y = torch.zeros(R, ld, device=DEV)
    Yp = torch.zeros(R, ld, device=DEV)
    L.call("magpo_linear", Xd, KIN, Wt, bd, Y, ld, Yp, R, KIN, NOUT, act, 0, stream)
    ref = X.double() @ W.double() + b.double()
    close(Yp[:, :NOUT], ref, what="pre")
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    close(Y[:, :NOUT], ref, what="act")


@pytest.mark.parametrize("KIN,NOUT,R", [(64, 64, 1000), (64, 256, 333), (128, 384, 200), (64, 20, 500), (256, 64, 100)])
def test_wgrad(L, stream, KIN, NOUT, R):
    g = torch.Generator().manual_seed(2)
    X = torch.randn(R, KIN, generator=g)
    ld = (NOUT + 3) // 4 * 4
    dY = torch.zeros(R, ld)
    dY[:, :NOUT] = torch.randn(R, NOUT, generator=g)
    G = 7
    ws = torch.empty(L.call("magpo_wgrad_workspace_floats", KIN, NOUT, G), device=DEV)
    dW = torch.zeros(KIN, NOUT, device=DEV)
    db = torch.zeros(NOUT, device=DEV)
    L.call("magpo_wgrad", dev(X), KIN, dev(dY), ld, R, KIN, KIN, NOUT, dW, db, ws, G, 0.5, 0, 0, stream)
    close(dW, 0.5 * X.double().T @ dY[:, :NOUT].double(), what="dW")
    close(db, 0.5 * dY[:, :NOUT].double().sum(0), what="db")


@pytest.mark.parametrize("KIN,NOUT,R,act", [(64, 64, 203, 0), (64, 256, 130, 1), (64, 192, 97, 0), (64, 20, 100, 0), (128, 384, 1000, 0),
                                             (128, 128, 333, 1), (128, 96, 65, 0), (256, 64, 77, 0), (192, 64, 65, 2), (384, 128, 70, 0),
                                             (384, 100, 31, 0)])
def test_linear_shared_tile(L, stream, KIN, NOUT, R, act):
    """k_linear_lds (no pre-activation copy requested): full and ragged tiles / column groups, 2- and 4-wave blocks."""
    g = torch.Generator().manual_seed(11)
    X = torch.randn(R, KIN, generator=g)
    W = torch.randn(KIN, NOUT, generator=g) / math.sqrt(KIN)
    b = torch.randn(NOUT, generator=g)
    Wt = transpose_pad(L, stream, dev(W))
    ld = (NOUT + 3) // 4 * 4
    Y = torch.full((R + 1, ld), 7.0, device=DEV)   # guard row: nothing may be written past R
    L.call("magpo_linear", dev(X), KIN, Wt, dev(b), Y, ld, None, R, KIN, NOUT, act, 0, stream)
    ref = X.double() @ W.double() + b.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    close(Y[:R, :NOUT], ref, what="y")
    assert bool((Y[R] == 7.0).all()) and (ld == NOUT or bool((Y[:R, NOUT:] == 7.0).all()))


@pytest.mark.parametrize("KIN,NOUT,R,act", [(128, 384, 1000, 0), (128, 128, 333, 1), (128, 100, 65, 0), (192, 128, 65, 2), (192, 256, 4001, 0), (128, 20, 77, 0)])
def test_linear_bf16_triples(L, stream, KIN, NOUT, R, act):
    """variant bit 2: the shared-tile dense kernel on bf16 MFMA with both operands split into three bf16 pieces (KIN 128 / 192): same
    tolerances against fp64 as the fp32-MFMA kernel (full and ragged row tiles, ragged column groups, every epilogue; the kernel takes the
    shapes whose column groups fill four-wave blocks -- (128, 20) falls back to fp32 MFMA by design)."""
    g = torch.Generator().manual_seed(14)
    X = torch.randn(R, KIN, generator=g); W = torch.randn(KIN, NOUT, generator=g) / math.sqrt(KIN); b = torch.randn(NOUT, generator=g)
    Wt = transpose_pad(L, stream, dev(W))
    ld = (NOUT + 3) // 4 * 4
    Y = torch.zeros(R, ld, device=DEV)
    L.call("magpo_linear", dev(X), KIN, Wt, dev(b), Y, ld, None, R, KIN, NOUT, act, 4, stream)
    ref = X.double() @ W.double() + b.double()
    ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref, approximate="tanh") if act == 2 else ref)
    close(Y[:, :NOUT], ref, what="bf16 triples")


def test_linear_bf16_triples_keep_fp32_accuracy(L, stream):
    """Acceptance bar for taking a GEMM kernel off fp32 MFMA: the error against the fp64 product is no larger than the
    fp32-MFMA kernel's on the same data."""
    g = torch.Generator().manual_seed(15)
    for KIN, NOUT, R in ((128, 384, 8192), (192, 128, 8192), (128, 128, 8192)):
        X = torch.randn(R, KIN, generator=g) * 1.3; W = torch.randn(KIN, NOUT, generator=g) / math.sqrt(KIN)
        Wt = transpose_pad(L, stream, dev(W))
        ref = X.double() @ W.double()
        err = {}
        for variant in (0, 4):
            Y = torch.zeros(R, NOUT, device=DEV)
            L.call("magpo_linear", dev(X), KIN, Wt, None, Y, NOUT, None, R, KIN, NOUT, 0, variant, stream)
            d = (Y.cpu().double() - ref).abs()
            err[variant] = (float(d.max()), float(d.mean()), float((d * d).mean().sqrt()))
        print("linear %d -> %d, |y - fp64| max / mean / rms: fp32 MFMA %.2e / %.2e / %.2e, bf16 triples %.2e / %.2e / %.2e" % (KIN, NOUT, *err[0], *err[4]))
        # mean and rms over the 1 - 3 M outputs must not exceed the fp32-MFMA kernel's; the max is one element's rounding luck (seen 0.77 - 1.19 x
        # across shapes and seeds) and only has to stay in the same range
        assert err[4][1] <= err[0][1] and err[4][2] <= err[0][2] and err[4][0] <= 1.3 * err[0][0], err


def test_linear_relu_mask_epilogue(L, stream):
    """act 4: dX = (dY W^T) masked by the forward activation passed in the Ypre slot (ReLU backward fused into the GEMM)."""
    KIN, NOUT, R = 64, 128, 173
    g = torch.Generator().manual_seed(14)
    X = torch.randn(R, KIN, generator=g)
    W = torch.randn(KIN, NOUT, generator=g) / 8
    M = torch.relu(torch.randn(R, NOUT, generator=g))
    Wt = transpose_pad(L, stream, dev(W))
    Y = torch.zeros(R, NOUT, device=DEV)
    L.call("magpo_linear", dev(X), KIN, Wt, None, Y, NOUT, dev(M), R, KIN, NOUT, 4, 0, stream)
    close(Y, (X.double() @ W.double()) * (M > 0), what="masked dX")


@pytest.mark.parametrize("KIN,NOUT,R,G", [(128, 384, 64 * 256 + 37, 200), (128, 384, 64 * 300, 200), (128, 128, 64 * 600, 512),
                                          (64, 256, 64 * 300, 256), (64, 192, 64 * 700, 512)])
def test_wgrad_whole_matrix(L, stream, KIN, NOUT, R, G):
    """Whole-matrix weight gradient (enough rows for one slab per CU) incl. the bias column sums: k_wgrad_full with a ragged
    last tile, k_wgrad_full_x when every tile is full (more tiles than slabs, so the prefetch pipeline runs)."""
    g = torch.Generator().manual_seed(12)
    X = torch.randn(R, KIN, generator=g) * 0.1
    dY = torch.randn(R, NOUT, generator=g) * 0.1
    ws = torch.empty(L.call("magpo_wgrad_workspace_floats", KIN, NOUT, G), device=DEV)
    dW = torch.zeros(KIN, NOUT, device=DEV)
    db = torch.zeros(NOUT, device=DEV)
    L.call("magpo_wgrad", dev(X), KIN, dev(dY), NOUT, R, KIN, KIN, NOUT, dW, db, ws, G, 1.0, 0, 0, stream)
    close(dW, X.double().T @ dY.double(), what="dW")
    close(db, dY.double().sum(0), what="db")


def test_wgrad_bf16_triples(L, stream):
    """variant bit 6 (OPT-IN): the 128 x 384 whole-matrix weight gradient on bf16 MFMA with both operands split into three bf16 pieces: inside
    the usual tolerances, 3.6 -> 2.9 ms per launch -- but it does NOT meet the bar for replacing fp32 MFMA (error against fp64 no larger than
    the fp32-MFMA kernel's): over 65 536 rows the error is the fp32 ACCUMULATION's, and six partial products per k-block put 19 % more of it
    into the sums (mean 4.4e-6 vs 3.7e-6).  Checked here: correct, and within 1.3 x of the fp32-MFMA kernel's error; never a default."""
    g = torch.Generator().manual_seed(16)
    KIN, NOUT, R, G = 128, 384, 64 * 1024, 256
    X = torch.randn(R, KIN, generator=g) * 0.3
    dY = torch.randn(R, NOUT, generator=g) * 0.2
    ref, refb = X.double().T @ dY.double(), dY.double().sum(0)
    ws = torch.empty(L.call("magpo_wgrad_workspace_floats", KIN, NOUT, G), device=DEV)
    err = {}
    for variant in (0, 64):
        dW = torch.zeros(KIN, NOUT, device=DEV); db = torch.zeros(NOUT, device=DEV)
        L.call("magpo_wgrad", dev(X), KIN, dev(dY), NOUT, R, KIN, KIN, NOUT, dW, db, ws, G, 1.0, 0, variant, stream)
        d = (dW.cpu().double() - ref).abs()
        err[variant] = (float(d.max()), float(d.mean()), float((d * d).mean().sqrt()))
        close(dW, ref, what="dW"); close(db, refb, what="db")
    print("wgrad 128 x 384 over %d rows, |dW - fp64| max / mean / rms: fp32 MFMA %.2e / %.2e / %.2e, bf16 triples %.2e / %.2e / %.2e" % (R, *err[0], *err[64]))
    assert err[64][1] <= 1.3 * err[0][1] and err[64][2] <= 1.3 * err[0][2] and err[64][0] <= 1.3 * err[0][0], err


@pytest.mark.parametrize("rows", [64, 32])
def test_gru_carry_equals_stepwise_scan(L, stream, rows):
    _gru_carry_case(L, stream, rows)


def _gru_carry_case(L, stream, rows):
    """magpo_gru_carry (time-major rollout trajectory, last state only) == the sequence-major scan on the same data; and xi rows
    taken through a class table (xi_cls, csrc/classtab.hip) == the materialised rows, in both row layouts, bit for bit."""
    N, T, A, H = 37, 9, 3, 128
    g = torch.Generator().manual_seed(13)
    C = 11
    tab = torch.randn(C, 3 * H, generator=g) * 0.5
    cls_tm = torch.randint(0, C, (T, N, A), generator=g).int()         # rows (t, env, agent)
    xi_tm = tab[cls_tm.long()]
    Wht = torch.randn(3 * H, H, generator=g) * 0.08
    bhn = torch.randn(H, generator=g) * 0.1
    h0 = torch.randn(N * A, H, generator=g) * 0.3
    reset_tm = (torch.rand(T, N, generator=g) < 0.2).to(torch.uint8)
    h_last = torch.zeros(N * A, H, device=DEV)
    L.call("magpo_gru_carry", dev(xi_tm.reshape(-1, 3 * H)), dev(Wht), dev(bhn), dev(h0), dev(reset_tm), h_last, N, T, A, None, rows, stream)
    xi_sm = xi_tm.permute(1, 0, 2, 3).contiguous()                     # rows (env, t, agent)
    hs = torch.zeros(N * T * A, H, device=DEV)
    L.call("magpo_gru_scan_fwd", dev(xi_sm.reshape(-1, 3 * H)), dev(Wht), dev(bhn), dev(h0), None, dev(reset_tm.t().contiguous()), hs, None, None,
           N, T, A, None, 0, rows, stream)
    ref = hs.view(N, T, A, H)[:, T - 1].reshape(N * A, H)
    assert torch.equal(h_last, ref)
    h_last2 = torch.zeros(N * A, H, device=DEV)
    L.call("magpo_gru_carry", dev(tab), dev(Wht), dev(bhn), dev(h0), dev(reset_tm), h_last2, N, T, A, dev(cls_tm.reshape(-1)), rows, stream)
    assert torch.equal(h_last2, h_last)
    hs2 = torch.zeros(N * T * A, H, device=DEV); gates = [torch.zeros(N * T * A, 4 * H, device=DEV) for _ in range(2)]
    hp = [torch.zeros(N * T * A, H, device=DEV) for _ in range(2)]
    cls_sm = cls_tm.permute(1, 0, 2).contiguous().reshape(-1)
    L.call("magpo_gru_scan_fwd", dev(xi_sm.reshape(-1, 3 * H)), dev(Wht), dev(bhn), dev(h0), None, dev(reset_tm.t().contiguous()), hs, gates[0], hp[0],
           N, T, A, None, 0, rows, stream)
    L.call("magpo_gru_scan_fwd", dev(tab), dev(Wht), dev(bhn), dev(h0), None, dev(reset_tm.t().contiguous()), hs2, gates[1], hp[1],
           N, T, A, dev(cls_sm), 0, rows, stream)
    assert torch.equal(hs2, hs) and torch.equal(gates[0], gates[1]) and torch.equal(hp[0], hp[1])


@pytest.mark.parametrize("F,R,relu", [(5, 100, 1), (5, 5003, 1), (8, 4096, 0), (3, 4096 * 32 * 2 + 4133, 1)])
def test_small_linear(L, stream, F, R, relu):
    """Actor pre-torso Dense(F -> 128) (+ReLU): the one-row kernel (small R) and the tiled grid-stride kernel (R >= 4096)."""
    g = torch.Generator().manual_seed(F * 1000 + R)
    X = torch.randn(R, F, generator=g)
    W = torch.randn(F, 128, generator=g) * 0.4
    b = torch.randn(128, generator=g) * 0.1
    Y = torch.full((R + 1, 128), -7.0, device=DEV)
    L.call("magpo_small_linear", dev(X), F, F, dev(W), dev(b), Y, 128, 128, R, relu, stream)
    ref = X.double() @ W.double() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    close(Y[:R], ref, what="small_linear")
    assert (Y[R] == -7.0).all()            # nothing written past the last row


def _pe_table(L, st, E=64, npos=101):
    pe = torch.empty(npos, E, device=DEV)
    L.call("magpo_pe_table", pe, npos, E, st)
    return pe


def test_pe_table(L, stream):
    pe = _pe_table(L, stream)
    ref = onets.positional_encoding(torch.arange(101), 64, torch.float32)
    close(pe, ref, 0, 1e-5, "pe")  # fp32 exp/sin ulp x position <= 100


def _slabsum(slab):
    return slab.double().sum(0)


@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("mode", [0, 1])
def test_embed_fwd_bwd(L, stream, mode, E):
    """E = row width (embed_dim of the device network): 16 or 32 lanes per row (csrc/rowops.hip)."""
    g = torch.Generator().manual_seed(3)
    R, F, K = 333, 5, 20
    pe = _pe_table(L, stream, E)
    pos = torch.randint(0, 101, (R,), generator=g, dtype=torch.int32)
    s_ln = 1 + 0.1 * torch.randn(E, generator=g)
    obs = torch.randint(0, 60, (R, F), generator=g).float()
    s_obs = 1 + 0.1 * torch.randn(F, generator=g)
    W = torch.randn(F if mode == 0 else K + 1, E, generator=g) * 0.5
    idx = torch.randint(0, K + 1, (R,), generator=g, dtype=torch.int32)
    z = torch.empty(R, E, device=DEV); xn = torch.empty_like(z); kin = torch.empty_like(z)
    L.call("magpo_embed_fwd", mode, dev(obs), F, F, dev(s_obs), dev(W), dev(idx), 1, dev(s_ln), pe, dev(pos), 1, 101,
           z, E, xn, E, kin, E, R, E, stream)
    # reference (fp64 autograd)
    Wd = W.double().requires_grad_(True); sl = s_ln.double().requires_grad_(True); so = s_obs.double().requires_grad_(True)
    if mode == 0:
        zr = onets.rmsnorm(obs.double(), so) @ Wd
    else:
        zr = Wd[idx.long()]
    xnr = onets.rmsnorm(onets.gelu(zr), sl)
    kinr = xnr + pe.cpu().double()[pos.long()]
    close(z, zr, what="z"); close(xn, xnr, what="xn"); close(kin, kinr, what="kin")
    d0 = torch.randn(R, E, generator=g); d1 = torch.randn(R, E, generator=g)
    (xnr * d0.double() + kinr * d1.double()).sum().backward()
    grid = L.call("magpo_row_grid", R)
    dz = torch.empty(R, E, device=DEV)
    slab_sln = torch.zeros(grid, E, device=DEV); slab_sobs = torch.zeros(grid, 32, device=DEV)
    slab_w = torch.zeros(grid, 32 * E, device=DEV)
    rows = F if mode == 0 else K + 1
    L.call("magpo_embed_bwd", mode, z, E, dev(d0), E, dev(d1), E, None, 0, dev(s_ln), dz, E, slab_sln, slab_w, rows,
           dev(obs), F, F, dev(s_obs), dev(W), slab_sobs, dev(idx), 1, R, E, stream)
    close(_slabsum(slab_sln), sl.grad, 1e-4, 1e-5, "ds_ln")
    dW = torch.zeros(rows, E, device=DEV)
    L.call("magpo_reduce_slabs", slab_w, dW, grid, rows * E, 32 * E, 1.0, 0, stream)
    close(dW, Wd.grad, 1e-4, 1e-5, "dW")
    if mode == 0:
        close(_slabsum(slab_sobs)[:F], so.grad, 1e-4, 1e-5, "ds_obs")


@pytest.mark.parametrize("mode", [0, 1])
def test_embed_fwd_grid_stride(L, stream, mode):
    """More rows than one pass of the grid: the prefetched, double-buffered input tiles of the embedding kernel."""
    g = torch.Generator().manual_seed(31 + mode)
    R, F, K = 2048 * 64 * 2 + 777, 5, 20
    pe = _pe_table(L, stream)
    pos = torch.randint(0, 101, (R,), generator=g, dtype=torch.int32)
    s_ln = 1 + 0.1 * torch.randn(64, generator=g)
    obs = torch.randint(0, 60, (R, F), generator=g).float()
    s_obs = 1 + 0.1 * torch.randn(F, generator=g)
    W = torch.randn(F if mode == 0 else K + 1, 64, generator=g) * 0.5
    idx = torch.randint(0, K + 1, (R,), generator=g, dtype=torch.int32)
    xn = torch.empty(R + 1, 64, device=DEV).fill_(-7.0); kin = torch.empty(R + 1, 64, device=DEV).fill_(-7.0)
    L.call("magpo_embed_fwd", mode, dev(obs), F, F, dev(s_obs), dev(W), dev(idx), 1, dev(s_ln), pe, dev(pos), 1, 101,
           None, 64, xn, 64, kin, 64, R, 64, stream)
    zr = onets.rmsnorm(obs.double(), s_obs.double()) @ W.double() if mode == 0 else W.double()[idx.long()]
    xnr = onets.rmsnorm(onets.gelu(zr), s_ln.double())
    close(xn[:R], xnr, what="xn"); close(kin[:R], xnr + pe.cpu().double()[pos.long()], what="kin")
    assert (xn[R] == -7.0).all() and (kin[R] == -7.0).all()


@pytest.mark.parametrize("E,hs,nh", [(64, 64, 1), (128, 128, 1), (128, 64, 2), (128, 32, 4)])
def test_retpost_resnorm_headmid(L, stream, E, hs, nh):
    g = torch.Generator().manual_seed(4)
    R = 300
    pe = _pe_table(L, stream, E)
    pos = torch.randint(0, 101, (R,), generator=g, dtype=torch.int32)
    grid = L.call("magpo_row_grid", R)
    rn = lambda *s: torch.randn(*s, generator=g)
    # retpost
    r, gp, du = rn(R, E), rn(R, E), rn(R, E)
    ga, be = 1 + 0.1 * rn(hs), 0.1 * rn(hs)
    u = torch.empty(R, E, device=DEV)
    L.call("magpo_retpost_fwd", dev(r), E, dev(gp), E, dev(ga), dev(be), u, E, R, hs, hs // nh, E, stream)
    rd, gpd, gad, bed = (t.double().requires_grad_(True) for t in (r, gp, ga, be))
    ur = onets.swish(gpd) * onets.groupnorm_rows(rd.reshape(R * (E // hs), hs), gad, bed, nh).reshape(R, E)
    close(u, ur, what="u")
    (ur * du.double()).sum().backward()
    dr = torch.empty(R, E, device=DEV); dgp = torch.empty(R, E, device=DEV)
    sg = torch.zeros(grid, E, device=DEV); sb = torch.zeros(grid, E, device=DEV)
    L.call("magpo_retpost_bwd", dev(r), E, dev(gp), E, dev(ga), dev(be), dev(du), E, dr, E, dgp, E, sg, sb, R, hs, hs // nh, E, stream)
    close(dr, rd.grad, 1e-4, 1e-5, "dr"); close(dgp, gpd.grad, 1e-4, 1e-5, "dgp")
    close(_slabsum(sg).reshape(E // hs, hs).sum(0), gad.grad, 1e-4, 1e-5, "dgamma"); close(_slabsum(sb).reshape(E // hs, hs).sum(0), bed.grad, 1e-4, 1e-5, "dbeta")
    # resnorm (two norms + pe) and (one norm)
    for two in (True, False):
        a, y, d0, d1 = rn(R, E), rn(R, E), rn(R, E), rn(R, E)
        s1, s2 = 1 + 0.1 * rn(E), 1 + 0.1 * rn(E)
        out = torch.empty(R, E, device=DEV); outpe = torch.empty(R, E, device=DEV)
        L.call("magpo_resnorm_fwd", dev(a), E, dev(y), E, dev(s1), dev(s2) if two else None, pe, dev(pos), 1, 101,
               out, E, outpe, E, R, E, stream)
        ad, yd, s1d, s2d = (t.double().requires_grad_(True) for t in (a, y, s1, s2))
        o = onets.rmsnorm(ad + yd, s1d)
        if two:
            o = onets.rmsnorm(o, s2d)
        ope = o + pe.cpu().double()[pos.long()]
        close(out, o, what="resnorm out"); close(outpe, ope, what="resnorm outpe")
        (o * d0.double() + ope * d1.double()).sum().backward()
        dsum = torch.empty(R, E, device=DEV)
        sl1 = torch.zeros(grid, E, device=DEV); sl2 = torch.zeros(grid, E, device=DEV)
        L.call("magpo_resnorm_bwd", dev(a), E, dev(y), E, dev(s1), dev(s2) if two else None, dev(d0), E, dev(d1), E,
               None, 0, dsum, E, sl1, sl2, R, E, stream)
        close(dsum, ad.grad, 1e-4, 1e-5, "dsum"); close(_slabsum(sl1), s1d.grad, 1e-4, 1e-5, "ds1")
        if two:
            close(_slabsum(sl2), s2d.grad, 1e-4, 1e-5, "ds2")
    # headmid: logits mode and value mode
    hpre, dhn = rn(R, E), rn(R, E)
    s, w, b, dv = 1 + 0.1 * rn(E), rn(E), rn(1), rn(R)
    hn = torch.empty(R, E, device=DEV)
    L.call("magpo_headmid_fwd", dev(hpre), E, dev(s), hn, E, None, None, None, 0, R, E, stream)
    hd, sd, wd, bd = (t.double().requires_grad_(True) for t in (hpre, s, w, b))
    hnr = onets.rmsnorm(onets.gelu(hd), sd)
    close(hn, hnr, what="hn")
    (hnr * dhn.double()).sum().backward()
    dh = torch.empty(R, E, device=DEV); ss = torch.zeros(grid, E, device=DEV)
    L.call("magpo_headmid_bwd", dev(hpre), E, dev(s), dev(dhn), E, None, None, 0, dh, E, ss, None, None, R, E, stream)
    close(dh, hd.grad, 1e-4, 1e-5, "dhpre"); close(_slabsum(ss), sd.grad, 1e-4, 1e-5, "ds")
    hd.grad = None; sd.grad = None
    val = torch.empty(R, device=DEV)
    L.call("magpo_headmid_fwd", dev(hpre), E, dev(s), None, 0, dev(w), dev(b), val, 1, R, E, stream)
    vr = onets.rmsnorm(onets.gelu(hd), sd) @ wd + bd
    close(val, vr, what="value")
    (vr * dv.double()).sum().backward()
    sw = torch.zeros(grid, E, device=DEV); sbb = torch.zeros(grid, device=DEV)
    L.call("magpo_headmid_bwd", dev(hpre), E, dev(s), None, 0, dev(w), dev(dv), 1, dh, E, ss, sw, sbb, R, E, stream)
    close(dh, hd.grad, 1e-4, 1e-5, "dhpre(v)"); close(_slabsum(ss), sd.grad, 1e-4, 1e-5, "ds(v)")
    close(_slabsum(sw), wd.grad, 1e-4, 1e-5, "dw"); close(sbb.double().sum(), bd.grad[0], 1e-4, 1e-5, "db")


def _ret_reference(q, k, v, s0, dones_t, A, kappa, masked):
    """Oracle decay-matrix form (retention.py:66-100) in fp64."""
    B, C, _ = q.shape
    dones = dones_t[:, :, None].expand(-1, -1, A).reshape(B, C)
    D = onets.decay_matrix(dones, A, kappa, masked, torch.float64)
    xi = onets.xi_vector(dones, A, kappa, torch.float64)
    return ((q @ k.transpose(1, 2)) * D) @ v + (q @ s0) * xi


@pytest.mark.parametrize("A,T,masked,hs", [(4, 40, 1, 64), (4, 128, 0, 64), (2, 50, 1, 64), (3, 45, 1, 64), (8, 24, 0, 64), (5, 30, 1, 64),
                                           # BASELINE config 5 / the 5x20 scenario at the default rollout length: 16 and 10 chunks, i.e. more
                                           # than the 8 whose decay bookkeeping is built once per workgroup (the per-chunk `!pre` branch)
                                           (8, 128, 0, 64), (8, 128, 1, 64), (5, 128, 0, 64), (5, 128, 1, 64),
                                           # narrow heads (n_head 2 / 4) over >= 3 chunks with a non-zero carried state: the padded-head
                                           # inter-chunk state hand-off, forward and backward
                                           (4, 48, 1, 32), (4, 48, 0, 16), (3, 70, 1, 16), (8, 128, 1, 32), (5, 128, 0, 16)])
@pytest.mark.parametrize("ct", [64, 32])
def test_retention_chunk(L, stream, A, T, masked, hs, ct):
    """ct = tokens per chunk: the 64-token kernels (one workgroup per CU in the backward) or the 32-token ones (retention32.hpp)."""
    _retention_chunk_case(L, stream, A, T, masked, hs, ct)


def _retention_chunk_case(L, stream, A, T, masked, hs, ct):
    g = torch.Generator().manual_seed(5)
    B, kappa = 5, 0.775
    C = T * A
    q, k, v, dr = (torch.randn(B, C, hs, generator=g) * 0.5 for _ in range(4))
    s0 = torch.randn(B, hs, hs, generator=g) * 0.3
    dones = torch.rand(B, T, generator=g) < (0.08 if T < 100 else 0.02)
    dones[0, 0] = True
    dones[1, :] = False
    dones[2, :] = False; dones[2, T // 2] = True     # exactly one episode start in the middle
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _ret_reference(qd, kd, vd, s0.double(), dones, A, kappa, bool(masked))
    (ref * dr.double()).sum().backward()
    nch = L.call("magpo_retention_num_chunks", T, A, ct)
    assert nch >= 3 or hs == 64
    # q, k, v live in one [R, 256] buffer like the fused projection output; with narrow heads the neighbouring columns hold the
    # other heads' data (random here): the kernel must neither read them into the product nor write over them
    buf = dev(torch.randn(B * C, 256, generator=g))
    buf[:, 0:hs] = dev(q.reshape(-1, hs)); buf[:, 64:64 + hs] = dev(k.reshape(-1, hs)); buf[:, 128:128 + hs] = dev(v.reshape(-1, hs))
    drb = dev(torch.randn(B * C, 64, generator=g))
    drb[:, :hs] = dev(dr.reshape(-1, hs))
    r = torch.full((B * C, 64), 7.0, device=DEV)
    states = torch.zeros(B, nch, 64, 64, device=DEV)
    sfin = torch.zeros(B, 64, 64, device=DEV)
    perm = torch.tensor([3, 1, 4, 0, 2], dtype=torch.int32)
    s0_store = torch.zeros(B, 64, 64)     # device head states are zero-padded to 64 x 64
    s0_store[perm.long(), :hs, :hs] = s0
    dn = dev(dones.to(torch.uint8))
    L.call("magpo_retention_chunk_fwd", buf, 256, buf[:, 64:], 256, buf[:, 128:], 256, r, 64, dev(s0_store), dev(perm),
           dn, states, sfin, B, T, A, masked, kappa, hs, None, ct, stream)
    close(r[:, :hs].reshape(B, C, hs), ref, 1e-4, 1e-5, "ret fwd")
    if hs < 64:
        assert bool((r[:, hs:] == 7.0).all()), "columns of the neighbouring heads were written"
    dbuf = torch.full((B * C, 256), 9.0, device=DEV)
    L.call("magpo_retention_chunk_bwd", buf, 256, buf[:, 64:], 256, buf[:, 128:], 256, drb, 64,
           dbuf, 256, dbuf[:, 64:], 256, dbuf[:, 128:], 256, dn, states, B, T, A, masked, kappa, hs, None, ct, stream)
    close(dbuf[:, 0:hs].reshape(B, C, hs), qd.grad, 1e-4, 1e-5, "dq")
    close(dbuf[:, 64:64 + hs].reshape(B, C, hs), kd.grad, 1e-4, 1e-5, "dk")
    close(dbuf[:, 128:128 + hs].reshape(B, C, hs), vd.grad, 1e-4, 1e-5, "dv")
    if hs < 64:
        for o in (0, 64, 128):
            assert bool((dbuf[:, o + hs:o + 64] == 9.0).all()), "gradient columns of the neighbouring heads were written"
    # the state after the last token (next-chunk hand-off, retention.py:88-92) against the recurrent form
    S = s0.double()
    for t in range(T):
        for b in range(B):
            if dones[b, t]:
                S[b] = 0
        kt = k.double().reshape(B, T, A, hs)[:, t]; vt = v.double().reshape(B, T, A, hs)[:, t]
        S = kappa * S + kt.transpose(1, 2) @ vt
    close(sfin[:, :hs, :hs], S, 1e-4, 1e-5, "final state")


@pytest.mark.parametrize("A,T,masked,hs", [(4, 128, 0, 64), (4, 40, 1, 64), (8, 128, 1, 64), (5, 30, 1, 64), (3, 70, 1, 16), (4, 8, 1, 64)])
@pytest.mark.parametrize("ct", [64, 32])
def test_retention_chunk_row_table(L, stream, A, T, masked, hs, ct):
    _retention_row_table_case(L, stream, A, T, masked, hs, ct)


def _retention_row_table_case(L, stream, A, T, masked, hs, ct):
    """q | k | v read through a row table (block-0 projections on the distinct input rows, csrc/classtab.hip) == the same rows
    gathered per token first: outputs, saved chunk states and all three gradients bit-identical (forward and backward, one to many
    chunks, ragged last chunk, narrow head)."""
    g = torch.Generator().manual_seed(11)
    B, kappa, C = 6, 0.775, T * A
    R, NC = B * C, 97                                   # 97 distinct rows, every token row points at one of them
    tab = dev(torch.randn(NC, 256, generator=g) * 0.5)
    rows = dev(torch.randint(0, NC, (R,), generator=g).to(torch.int32))
    buf = tab[rows.long()].contiguous()
    drb = dev(torch.randn(R, 64, generator=g))
    dones = torch.rand(B, T, generator=g) < 0.05
    dn = dev(dones.to(torch.uint8))
    s0 = dev(torch.randn(B, 64, 64, generator=g) * 0.3)
    if hs < 64:
        s0[:, hs:, :] = 0; s0[:, :, hs:] = 0
    nch = L.call("magpo_retention_num_chunks", T, A, ct)
    out = []
    for src, ridx in ((buf, None), (tab, rows)):
        r = torch.full((R, 64), 7.0, device=DEV)
        states = torch.zeros(B, nch, 64, 64, device=DEV); sfin = torch.zeros(B, 64, 64, device=DEV)
        L.call("magpo_retention_chunk_fwd", src, 256, src[:, 64:], 256, src[:, 128:], 256, r, 64, s0, None, dn, states, sfin, B, T, A,
               masked, kappa, hs, ridx, ct, stream)
        dbuf = torch.full((R, 256), 9.0, device=DEV)
        L.call("magpo_retention_chunk_bwd", src, 256, src[:, 64:], 256, src[:, 128:], 256, drb, 64, dbuf, 256, dbuf[:, 64:], 256,
               dbuf[:, 128:], 256, dn, states, B, T, A, masked, kappa, hs, ridx, ct, stream)
        out.append((r, states, sfin, dbuf))
    for x, y, what in zip(out[0], out[1], ("ret", "chunk states", "final state", "dq | dk | dv")):
        assert torch.equal(x, y), what


def test_retention_recurrent(L, stream):
    g = torch.Generator().manual_seed(6)
    N, A = 37, 4
    S = torch.randn(N, 64, 64, generator=g)
    q, k, v = (torch.randn(N * A, 64, generator=g) for _ in range(3))
    Sd = dev(S)
    r = torch.zeros(N * A, 64, device=DEV)
    L.call("magpo_retention_recurrent", Sd, dev(q), 64, dev(k), 64, dev(v), 64, A, r, 64, N, A, 0, 0.775, 1, None, 0, None, None, 64, 64, stream)
    qq, kk, vv = (t.double().reshape(N, A, 64) for t in (q, k, v))
    Sn = 0.775 * S.double() + kk.transpose(1, 2) @ vv
    close(Sd, Sn, what="S"); close(r.reshape(N, A, 64), qq @ Sn, what="ret")
    # decoder iteration i = 2: tokens 0..2 applied on the fly, output for token 2 only, state left untouched
    S2 = dev(S)
    r2 = torch.zeros(N * A, 64, device=DEV)
    L.call("magpo_retention_recurrent", S2, dev(q), 64, dev(k), 64, dev(v), 64, A, r2, 64, N, 3, 2, 0.775, 0, None, 0, None, None, 64, 64, stream)
    Sn = 0.775 * S.double() + kk[:, :3].transpose(1, 2) @ vv[:, :3]
    close(S2, S, 0, 0, "state must not be written")
    close(r2.reshape(N, A, 64)[:, 2], (qq[:, 2:3] @ Sn)[:, 0], what="ret token 2")
    assert r2.reshape(N, A, 64)[:, :2].abs().max().item() == 0 and r2.reshape(N, A, 64)[:, 3].abs().max().item() == 0


def test_gru_scan_bf16_triples_keep_fp32_accuracy(L, stream):
    """split_bf16 = 2 (three bf16 pieces per operand = 24 mantissa bits, six products) against the fp64 oracle: no worse than the fp32-MFMA
    scan on the same data (the acceptance bar for replacing fp32 MFMA); the two-piece mode (16 bits) is ~10 x worse."""
    g = torch.Generator().manual_seed(17)
    nseq, T, A, H = 48, 40, 4, 128
    R = nseq * T * A
    xi = torch.randn(R, 3 * H, generator=g) * 0.7
    Wh = torch.randn(H, 3 * H, generator=g) * 0.09
    bhn = torch.randn(H, generator=g) * 0.1
    h0 = torch.randn(nseq * A, H, generator=g) * 0.3
    done = torch.rand(nseq, T, generator=g) < 0.05
    # fp64 reference on rows (seq, t, a)
    x = xi.double().reshape(nseq, T, A, 3 * H)
    h = h0.double().reshape(nseq, A, H)
    ref = []
    for t in range(T):
        h = torch.where(done[:, t][:, None, None], torch.zeros_like(h), h)
        hh = h @ Wh.double()
        r = torch.sigmoid(x[:, t, :, :H] + hh[..., :H]); z = torch.sigmoid(x[:, t, :, H:2 * H] + hh[..., H:2 * H])
        n = torch.tanh(x[:, t, :, 2 * H:] + r * (hh[..., 2 * H:] + bhn.double()))
        h = (1 - z) * n + z * h
        ref.append(h)
    ref = torch.stack(ref, 1).reshape(R, H)
    Wht = transpose_pad(L, stream, dev(Wh))
    err = {}
    for mode in (0, 1, 2):
        hs = torch.empty(R, H, device=DEV); gates = torch.empty(R, 4 * H, device=DEV); hp = torch.empty(R, H, device=DEV)
        L.call("magpo_gru_scan_fwd", dev(xi), Wht, dev(bhn), dev(h0), None, dev(done.to(torch.uint8)), hs, gates, hp, nseq, T, A, None, mode, 0, stream)
        err[mode] = float((hs.cpu().double() - ref).abs().max())
    print("GRU forward scan, max |h - fp64|: fp32 MFMA %.2e, bf16 pairs %.2e, bf16 triples %.2e" % (err[0], err[1], err[2]))
    assert err[2] <= 1.05 * err[0], err     # (measured: 2.97e-7 vs 3.15e-7, profiles/r03_bf16_triples_error_vs_fp64.txt)
    assert err[1] > err[2]


@pytest.mark.parametrize("split,rows", [(0, 64), (0, 32), (1, 64), (2, 64)])
def test_gru_scan(L, stream, split, rows):
    """split = 1: the training scans on split-bf16 x3 MFMA (products hi*hi + hi*lo + lo*hi, ~2^-16 relative) instead of fp32 MFMA:
    same tolerances against the fp64 oracle."""
    _gru_scan_case(L, stream, split, rows)   # both block sizes (by size alone this case would only ever run 32-row blocks)


def _gru_scan_case(L, stream, split, rows):
    g = torch.Generator().manual_seed(7)
    nseq, T, A, F, K, H = 21, 9, 4, 5, 20, 128
    p = onets.init_actor_params(3, F, H, K, dtype=torch.float64)
    for n in ("gru.ir.bias", "gru.iz.bias", "gru.in.bias", "gru.hn.bias"):
        p[n] = torch.randn(H, generator=g, dtype=torch.float64) * 0.1
    p = {n: t.requires_grad_(True) for n, t in p.items()}
    emb = torch.randn(T, nseq, A, H, generator=g).double().requires_grad_(True)
    h0 = torch.randn(nseq, A, H, generator=g).double()
    done = torch.rand(nseq, T, generator=g) < 0.2
    done[0, 0] = True
    # oracle scan on pre-computed embeddings
    h = h0
    hs = []
    for t in range(T):
        h = torch.where(done[:, t][:, None, None], torch.zeros_like(h), h)
        h = onets.gru_cell(p, h, emb[t])
        hs.append(h)
    hs = torch.stack(hs)  # (T, nseq, A, H)
    dhs = torch.randn(T, nseq, A, H, generator=g)
    (hs * dhs.double()).sum().backward()
    # device layout rows (seq, t, a)
    to_rows = lambda x: x.detach().permute(1, 0, 2, 3).reshape(nseq * T * A, -1).float()
    Wi = torch.cat([p["gru.ir.kernel"], p["gru.iz.kernel"], p["gru.in.kernel"]], 1).detach().float()
    bi = torch.cat([p["gru.ir.bias"], p["gru.iz.bias"], p["gru.in.bias"]]).detach().float()
    Wh = torch.cat([p["gru.hr.kernel"], p["gru.hz.kernel"], p["gru.hn.kernel"]], 1).detach().float()
    R = nseq * T * A
    xi = torch.empty(R, 3 * H, device=DEV)
    L.call("magpo_linear", dev(to_rows(emb)), H, transpose_pad(L, stream, dev(Wi)), dev(bi), xi, 3 * H, None, R, H, 3 * H, 0, 0, stream)
    hsd = torch.empty(R, H, device=DEV); gates = torch.empty(R, 4 * H, device=DEV); hprev = torch.empty(R, H, device=DEV)
    perm = torch.randperm(nseq * A, generator=g).int()
    h0_store = torch.zeros(nseq * A, H)
    h0_store[perm.long()] = h0.reshape(-1, H).float()
    rs = dev(done.to(torch.uint8))
    Wht = transpose_pad(L, stream, dev(Wh))
    L.call("magpo_gru_scan_fwd", xi, Wht, dev(p["gru.hn.bias"].detach().float()), dev(h0_store), dev(perm), rs, hsd, gates, hprev,
           nseq, T, A, None, split, rows, stream)
    close(hsd, to_rows(hs), 1e-4, 1e-5, "hs")
    dg = torch.empty(R, 4 * H, device=DEV)     # (dn_in | dr | dz | dn_hid), include/magpo.h
    nblk = (nseq * A + 63) // 64
    slab = torch.zeros(nblk, H, device=DEV)
    L.call("magpo_gru_scan_bwd", gates, hprev, rs, dev(to_rows(dhs)), dev(Wh), dg, slab, nseq, T, A, split, rows, stream)
    dxi = torch.cat([dg[:, H:3 * H], dg[:, :H]], 1).contiguous()     # input side in W_i's gate order (r | z | n)
    dhh = dg[:, H:]                                                  # hidden side (r | z | n), ld 4H
    # check through the parameter gradients
    G = 3
    ws = torch.empty(L.call("magpo_wgrad_workspace_floats", H, 3 * H, G), device=DEV)
    dWi = torch.zeros(H, 3 * H, device=DEV); dbi = torch.zeros(3 * H, device=DEV); dWh = torch.zeros(H, 3 * H, device=DEV)
    L.call("magpo_wgrad", dev(to_rows(emb)), H, dxi, 3 * H, R, H, H, 3 * H, dWi, dbi, ws, G, 1.0, 0, 0, stream)
    L.call("magpo_wgrad", hprev, H, dhh, 4 * H, R, H, H, 3 * H, dWh, None, ws, G, 1.0, 0, 0, stream)
    refWi = torch.cat([p["gru.ir.kernel"].grad, p["gru.iz.kernel"].grad, p["gru.in.kernel"].grad], 1)
    refWh = torch.cat([p["gru.hr.kernel"].grad, p["gru.hz.kernel"].grad, p["gru.hn.kernel"].grad], 1)
    refbi = torch.cat([p["gru.ir.bias"].grad, p["gru.iz.bias"].grad, p["gru.in.bias"].grad])
    close(dWi, refWi, 1e-4, 1e-5, "dWi"); close(dWh, refWh, 1e-4, 1e-5, "dWh"); close(dbi, refbi, 1e-4, 1e-5, "dbi")
    close(_slabsum(slab), p["gru.hn.bias"].grad, 1e-4, 1e-5, "dbhn")
    demb = torch.empty(R, H, device=DEV)
    L.call("magpo_linear", dxi, 3 * H, dev(Wi), None, demb, H, None, R, 3 * H, H, 0, 0, stream)
    close(demb, to_rows(emb.grad), 1e-4, 1e-5, "demb")


def test_sample_categorical(L, stream):
    g = torch.Generator().manual_seed(8)
    N, K, A = 1000, 20, 4
    logits = torch.randn(N, 32, generator=g)
    key = oprng.split(oprng.prng_key(7), 3)[2]
    act = torch.zeros(N, A, dtype=torch.int32, device=DEV); lp = torch.zeros(N, A, device=DEV)
    nxt = torch.zeros(N, A, dtype=torch.int32, device=DEV); lpall = torch.zeros(N, 32, device=DEV)
    L.call("magpo_sample_categorical", dev(logits), 32, None, 0, int(key[0]), int(key[1]), None, act[:, 1:], A, lp[:, 1:], A,
           nxt[:, 2:], A, lpall, 32, N, K, stream)
    lpd = lpall[:, :K].cpu()
    ref_lp = torch.log_softmax(logits[:, :K].double(), -1)
    close(lpd, ref_lp, 1e-5, 1e-6, "log-softmax")
    ref_a = oprng.categorical(key, lpd.numpy()[:, None, :])[:, 0]  # oracle sampling on the device's own log-probs
    assert np.array_equal(act[:, 1].cpu().numpy(), ref_a), "sampled action indices must be bit-exact"
    assert np.array_equal(nxt[:, 2].cpu().numpy(), ref_a + 1)
    close(lp[:, 1], lpd.gather(1, torch.from_numpy(ref_a).long()[:, None])[:, 0], 0, 0, "logp")


def test_threefry_device(L, stream):
    key = oprng.prng_key(1234)
    out = torch.zeros(1000, 2, dtype=torch.int32, device=DEV)
    kd = dev(key.view(np.int32))
    L.call("magpo_threefry_split", kd, out, 1000, stream)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oprng.split(key, 1000))
    bits = torch.zeros(777, dtype=torch.int32, device=DEV)
    L.call("magpo_threefry_random_bits", kd, bits, 777, stream)
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), oprng.random_bits(key, 777))


class DevEnv:
    def __init__(self, L, st, spec, N):
        self.L, self.st, self.spec, self.N = L, st, spec, N
        A, K, TL = spec.num_agents, spec.num_actions, spec.time_limit
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=DEV)
        self.step_count, self.target, self.record = i32(N), i32(N, TL + 1), i32(N, K, TL)
        self.key, self.mkey = i32(N, 2), i32(N, 2)
        self.run_ret, self.run_len = torch.zeros(N, device=DEV), i32(N)
        self.ep_ret, self.ep_len = torch.zeros(N, device=DEV), i32(N)
        self.obs, self.obs_step = torch.zeros(N, A, A + 1, device=DEV), i32(N)
        self.reward = torch.zeros(N, A, device=DEV)
        self.discount = torch.full((N, A), -1.0, device=DEV)
        self.done = torch.zeros(N, dtype=torch.uint8, device=DEV)
        self.m_ret, self.m_len, self.m_term = torch.zeros(N, device=DEV), i32(N), torch.zeros(N, dtype=torch.uint8, device=DEV)

    def _state(self):
        return (self.step_count, self.target, self.record, self.key, self.mkey, self.run_ret, self.run_len, self.ep_ret, self.ep_len)

    def _cfg(self):
        s = self.spec
        return (self.N, s.num_agents, s.num_actions, s.time_limit, s.maxval)

    def reset(self, env_keys):
        self.L.call("magpo_coordsum_reset", *self._state(), *self._cfg(), dev(env_keys.view(np.int32)), self.obs, self.obs_step, self.st)

    def step(self, actions, auto_reset=1):
        self.L.call("magpo_coordsum_step", *self._state(), *self._cfg(), actions, self.spec.num_agents, self.reward, self.discount, self.done,
                    self.obs, self.obs_step, self.m_ret, self.m_len, self.m_term, auto_reset, self.st)


@pytest.mark.parametrize("A,K,TL,maxval,auto", [(4, 20, 100, 60, 1), (2, 3, 4, 5, 1), (8, 15, 100, 100, 1), (3, 10, 20, 30, 0)])
def test_coordsum_env(L, stream, A, K, TL, maxval, auto):
    spec = ocs.CoordSumSpec(A, K, TL, maxval)
    N = 64
    keys = oprng.split(oprng.prng_key(11), N)
    st, ts = ocs.reset(spec, keys)
    env = DevEnv(L, stream, spec, N)
    env.reset(keys)
    assert np.array_equal(env.target.cpu().numpy(), st["target"])
    assert np.array_equal(env.key.cpu().numpy().view(np.uint32), st["key"])
    assert np.array_equal(env.obs.cpu().numpy(), ts["observation"]["agents_view"].astype(np.float32))
    rng = np.random.default_rng(0)
    nsteps = 2 * TL + 7 if auto else TL + 3
    for i in range(nsteps):
        tgt = st["target"][np.arange(N), np.minimum(st["step_count"], TL)]
        acts = rng.integers(0, K, size=(N, A)).astype(np.int32)
        # make the sum match on roughly half the envs so both reward branches run
        fix = rng.random(N) < 0.5
        for n in np.nonzero(fix)[0]:
            rest = int(tgt[n]) - int(acts[n, 1:].sum())
            if 0 <= rest < K:
                acts[n, 0] = rest
        st, ts = ocs.step(spec, st, acts, auto_reset=bool(auto))
        env.step(dev(acts), auto)
        assert np.array_equal(env.reward.cpu().numpy(), ts["reward"]), f"reward step {i}"
        assert np.array_equal(env.discount.cpu().numpy(), ts["discount"]), f"discount step {i}"
        assert np.array_equal(env.done.cpu().numpy().astype(bool), ts["step_type"] == ocs.STEP_LAST)
        assert np.array_equal(env.obs.cpu().numpy(), ts["observation"]["agents_view"].astype(np.float32)), f"obs step {i}"
        assert np.array_equal(env.obs_step.cpu().numpy(), ts["observation"]["step_count"][:, 0])
        assert np.array_equal(env.m_ret.cpu().numpy(), ts["episode_metrics"]["episode_return"])
        assert np.array_equal(env.m_len.cpu().numpy(), ts["episode_metrics"]["episode_length"])
    assert np.array_equal(env.record.cpu().numpy(), st["record"])
    assert np.array_equal(env.target.cpu().numpy(), st["target"])
    assert np.array_equal(env.key.cpu().numpy().view(np.uint32), st["key"])


@pytest.mark.parametrize("T,N,A", [(37, 50, 4), (128, 64, 8), (200, 7, 3), (64, 5, 2), (9, 50, 4), (37, 4096, 4)])
def test_gae(L, stream, T, N, A):
    """multistep.py:24-68.  N * A < 8192 and T >= 16: the wavefront prefix scan over time (one wave per sequence: full tiles, a ragged last
    tile, several tiles with a carried advantage); otherwise one thread per sequence (the last two shapes)."""
    g = torch.Generator().manual_seed(9)
    reward, value = torch.randn(T, N, A, generator=g), torch.randn(T, N, A, generator=g)
    done_env = torch.rand(T, N, generator=g) < 0.1
    last_val = torch.randn(N, A, generator=g)
    last_done_env = torch.rand(N, generator=g) < 0.3
    adv, tg = olearn.calculate_gae(reward.double(), value.double(), done_env[:, :, None].expand(T, N, A), last_val.double(),
                                   last_done_env[:, None].expand(N, A), 0.99, 0.95)
    a = torch.empty(T, N, A, device=DEV); t = torch.empty(T, N, A, device=DEV)
    L.call("magpo_gae", dev(reward), dev(value), dev(done_env.to(torch.uint8)), dev(last_val), dev(last_done_env.to(torch.uint8)),
           a, t, T, N, A, 0.99, 0.95, stream)
    close(a, adv, 1e-5, 1e-5, "adv"); close(t, tg, 1e-5, 1e-5, "targets")


@pytest.mark.parametrize("masked", [False, True])
def test_loss_fwd_bwd(L, stream, masked):
    g = torch.Generator().manual_seed(10)
    R, K = 1500, 20
    sysc = olearn.SystemCfg()
    gl = torch.randn(R, K, generator=g) * 0.7
    al = gl + torch.randn(R, K, generator=g) * 0.6   # far enough that the clip_gpo mask triggers on some rows
    mask = torch.ones(R, K, dtype=torch.bool)
    if masked:
        mask = torch.rand(R, K, generator=g) < 0.8
        mask[:, 0] = True
    action = torch.multinomial(mask.float(), 1, generator=g)[:, 0]
    value, vold, tgt, adv = (torch.randn(R, generator=g) for _ in range(4))
    value = vold + 0.3 * value
    gld, ald, vd = (t.double().requires_grad_(True) for t in (gl, al, value))
    glp = onets.masked_log_softmax(gld, mask); alp = onets.masked_log_softmax(ald, mask)
    g_logp = glp.gather(1, action[:, None])[:, 0]; a_logp = alp.gather(1, action[:, None])[:, 0]
    old = (g_logp + 0.1 * torch.randn(R, generator=g).double()).detach()
    pr = glp.exp()
    ent = -torch.where(pr == 0, torch.zeros_like(pr), pr * glp).sum(-1)
    mb = dict(log_prob=old, adv=adv.double(), value=vold.double(), targets=tgt.double())
    tl_g, gi = olearn.guider_loss(sysc, vd, g_logp, ent, glp, alp, a_logp, mb)
    tl_a, ai = olearn.actor_loss(sysc, glp, alp, a_logp, mb)
    gg, gv = torch.autograd.grad(tl_g, [gld, vd], retain_graph=True)
    (ga,) = torch.autograd.grad(tl_a, [ald])
    pad = lambda x: torch.cat([x, torch.zeros(R, 32 - K)], 1)
    stats = torch.empty(2, device=DEV)
    ws = torch.empty(8 * 1024, dtype=torch.float64, device=DEV)
    L.call("magpo_adv_moments", dev(adv), R, ws, stats, stream)
    close(stats[0], adv.double().mean(), 1e-6, 1e-7, "adv mean")
    close(stats[1], 1 / (adv.double().std(unbiased=False) + 1e-8), 1e-5, 0, "adv rstd")
    dg = torch.empty(R, 32, device=DEV); da = torch.empty(R, 32, device=DEV); dv = torch.empty(R, device=DEV)
    lo = torch.empty(9, device=DEV)
    L.call("magpo_loss_fwd_bwd", dev(pad(gl)), 32, dev(pad(al)), 32, dev(mask.to(torch.uint8)) if masked else None, dev(action.int()),
           dev(old.float()), dev(vold), dev(value), dev(adv), dev(tgt), stats, dg, 32, da, 32, dv, ws, lo,
           R, K, sysc.clip_eps, sysc.clip_gpo, sysc.ent_coef, sysc.vf_coef, sysc.alpha, stream)
    lo = lo.cpu()
    assert float(gi["kl_loss"].detach()) > 0, "test must exercise the clip_gpo mask"
    for i, ref in [(1, gi["value_loss"]), (2, ai["actor_loss"]), (3, gi["guider_loss"]), (4, gi["kl_loss"]), (5, gi["entropy"]),
                   (6, ai["actor_kl"]), (7, tl_g), (8, tl_a)]:
        close(lo[i], ref.detach(), 2e-5, 1e-6, f"loss[{i}]")
    close(dg[:, :K], gg, 2e-4, 1e-9, "dlogits_g"); close(da[:, :K], ga, 2e-4, 1e-9, "dlogits_a"); close(dv, gv, 2e-4, 1e-9, "dvalue")
    assert dg[:, K:].abs().max().item() == 0


def test_clip_adam(L, stream):
    g = torch.Generator().manual_seed(12)
    n = 5000
    for scale in (1e-3, 3.0):  # below / above the clipping threshold
        p = {"w": torch.randn(n, generator=g)}
        gr = {"w": torch.randn(n, generator=g) * scale}
        opt = olearn.adam_init(p)
        pd, md, vd = dev(p["w"]), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        ws = torch.empty(1024, dtype=torch.float64, device=DEV); gn = torch.empty(1, device=DEV)
        for step in range(1, 4):
            p, opt, gnorm = olearn.clip_adam_step(p, gr, opt, 2.5e-4, 0.5)
            bc1 = float(np.float32(1) - np.float32(0.9) ** np.float32(step)); bc2 = float(np.float32(1) - np.float32(0.999) ** np.float32(step))
            L.call("magpo_clip_adam", pd, dev(gr["w"] * 2), md, vd, n, 0.5, 0.5, 2.5e-4, 0.9, 0.999, 1e-5, bc1, bc2, ws, gn, stream)
            close(gn[0], gnorm, 1e-6, 0, "gnorm")
            close(pd, p["w"], 1e-6, 1e-7, "params")


def test_gather_minibatch(L, stream):
    g = torch.Generator().manual_seed(13)
    T, N, A, F, K, mb = 6, 10, 4, 5, 20, 5
    obs = torch.randn(T, N, A, F, generator=g)
    action = torch.randint(0, K, (T, N, A), generator=g, dtype=torch.int32)
    sc = torch.randint(0, 100, (T, N), generator=g, dtype=torch.int32)
    done = (torch.rand(T, N, generator=g) < 0.3).to(torch.uint8)
    val, lp, adv, tg = (torch.randn(T, N, A, generator=g) for _ in range(4))
    bp = torch.randperm(N, generator=g).int(); ap = torch.randperm(A, generator=g).int()
    env_idx = bp[mb:2 * mb]
    R = mb * T * A
    o_obs = torch.empty(R, F, device=DEV); o_act = torch.empty(R, dtype=torch.int32, device=DEV); o_prev = torch.empty_like(o_act)
    o_pos = torch.empty_like(o_act); o_done = torch.empty(mb, T, dtype=torch.uint8, device=DEV)
    o_val, o_lp, o_adv, o_tg = (torch.empty(R, device=DEV) for _ in range(4))
    o_h0 = torch.empty(mb * A, dtype=torch.int32, device=DEV)
    L.call("magpo_gather_minibatch", dev(obs), dev(action), dev(sc), dev(done), None, dev(val), dev(lp), dev(adv), dev(tg),
           dev(env_idx), dev(ap), o_obs, o_act, o_prev, o_pos, o_done, None, o_val, o_lp, o_adv, o_tg, o_h0, T, N, A, F, K, mb, stream)

    def prep(x):  # rec_magpo.py:445-453 on the selected envs
        x = x.index_select(1, env_idx.long()).index_select(2, ap.long()).transpose(0, 1)
        return x.reshape(mb, T * A, *x.shape[3:])
    assert torch.equal(o_obs.cpu().reshape(mb, T * A, F), prep(obs))
    assert torch.equal(o_act.cpu().reshape(mb, T * A), prep(action))
    assert torch.equal(o_val.cpu().reshape(mb, T * A), prep(val))
    assert torch.equal(o_adv.cpu().reshape(mb, T * A), prep(adv))
    sh = onets.shifted_actions(prep(action), K, A, torch.float32).argmax(-1).int()
    assert torch.equal(o_prev.cpu().reshape(mb, T * A), sh)
    assert torch.equal(o_pos.cpu().reshape(mb, T, A)[:, :, 0], sc.index_select(1, env_idx.long()).T)
    assert torch.equal(o_done.cpu(), done.index_select(1, env_idx.long()).T)
    assert torch.equal(o_h0.cpu().reshape(mb, A), env_idx[:, None] * A + ap[None, :])


@pytest.mark.parametrize("R,C,W", [(5000, 7, 384), (3000, 240, 256), (4097, 2121, 64), (100, 300, 64)])
def test_gather_rows_and_class_sum(L, stream, R, C, W):
    """csrc/classtab.hip: out[r] = table[cls[r]]; per-class row sums through a stable order (bit-stable, empty classes = 0)."""
    g = torch.Generator().manual_seed(R + C)
    cls = torch.randint(0, C, (R,), generator=g).int()
    if C > 3:
        cls[cls == 2] = 3     # an empty class
    table = torch.randn(C, W, generator=g)
    out = torch.zeros(R, W + 4, device=DEV)
    L.call("magpo_gather_rows", dev(table), W, dev(cls), out, W + 4, R, W, stream)
    assert torch.equal(out[:, :W].cpu(), table[cls.long()])
    X = torch.randn(R, W, generator=g)
    cd = dev(cls)
    order = torch.sort(cd, stable=True).indices
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(torch.bincount(cd, minlength=C), 0)])
    S = L.call("magpo_class_sum_slots", C)
    part = torch.empty(S, C, W, device=DEV)
    res = [torch.empty(C, W, device=DEV) for _ in range(2)]
    for o in res:
        L.call("magpo_class_sum", dev(X), W, order, offsets, C, W, part, o, stream)
    ref = torch.zeros(C, W, dtype=torch.float64).index_add_(0, cls.long(), X.double())
    close(res[0], ref, 1e-5, 1e-5, "class sums")
    assert torch.equal(res[0], res[1]), "class sums must be bit-stable"
