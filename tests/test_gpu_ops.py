"""GPU parity of the op-level C-ABI entry points against the CPU oracle (same seeded inputs).
Every call goes through ctypes into libdiffpool_hip.so — no torch arithmetic on the checked path.

Tolerances: the kernels compute in exact fp32 (f32 MFMA = fma chain), so differences from the oracle
are reduction-order only: rtol 1e-4 / atol 1e-5 forward, rtol 1e-3 on gradients.
"""
import numpy as np
import pytest
import torch

from graph_pooling_amd import _lib
from oracle import diffpool_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return _lib.load()


def dev(t):
    return t.cuda().contiguous()


def S():
    return torch.cuda.current_stream().cuda_stream


def ws_of(nbytes):
    return torch.empty(max(int(nbytes), 256), device="cuda", dtype=torch.uint8)


def close(a, b, rtol=1e-4, atol=1e-5):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert torch.isfinite(a).all() and torch.isfinite(b).all(), 'non-finite values in a parity check'
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ bgemm
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (50, 60, 500), (500, 40, 500), (89, 20, 37), (33, 129, 65),
                                   (64, 64, 32), (7, 300, 3)])
def test_bgemm_shapes(lib, tA, tB, M, N, K):
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    batch = 3
    A = torch.randn(batch, K, M, generator=g) if tA else torch.randn(batch, M, K, generator=g)
    Bm = torch.randn(batch, N, K, generator=g) if tB else torch.randn(batch, K, N, generator=g)
    C0 = torch.randn(batch, M, N, generator=g)
    bias = torch.randn(N, generator=g)
    Ad, Bd, Cd, bd = dev(A), dev(Bm), dev(C0), dev(bias)
    lda, ldb = A.shape[2], Bm.shape[2]
    rc = lib.dp_bgemm_f32(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), bd.data_ptr(), batch, M, N, K, lda, ldb, N,
                          A.shape[1] * lda, Bm.shape[1] * ldb, M * N, tA, tB, 0.5, 2.0, 0, S())
    _lib.check(rc)
    opA = A.transpose(1, 2) if tA else A
    opB = Bm.transpose(1, 2) if tB else Bm
    ref = 0.5 * (opA.double() @ opB.double()) + 2.0 * C0.double() + bias.double()
    close(Cd, ref.float(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,beta", [(256, 256, 1024, 0.0), (1024, 256, 256, 1.0), (300, 296, 90, 0.0),
                                        (129, 97, 65, 1.0), (7, 5, 3, 0.0), (128, 128, 32, 0.0),
                                        # N <= 64: the 128 x 64 tile (the pooling products at 60 columns)
                                        (256, 60, 1024, 0.0), (1024, 60, 256, 1.0), (130, 64, 70, 0.0),
                                        (200, 33, 50, 1.0)])
def test_bgemm_split_bf16_is_fp32_grade(lib, tA, tB, M, N, K, beta):
    """dp_bgemm_split_bf16: both fp32 operands split into three bf16 planes, six plane products on the bf16 MFMA.
    Wide-dynamic-range operands (values over 6 decades, mixed signs); against an fp64 product the error must be within a
    few fp32 roundings of the row's |a|.|b| — the bar an fp32 accumulation itself meets (rtol 2e-6 of sum |a_k b_k|)."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    batch = 2

    def wide(*shape):
        return torch.randn(*shape, generator=g) * torch.pow(10.0, torch.randint(-3, 3, shape, generator=g).float())
    A = wide(batch, K, M) if tA else wide(batch, M, K)
    Bm = wide(batch, N, K) if tB else wide(batch, K, N)
    C0 = torch.randn(batch, M, N, generator=g)
    Ad, Bd, Cd = dev(A), dev(Bm), dev(C0)
    lda, ldb = A.shape[2], Bm.shape[2]
    _lib.check(lib.dp_bgemm_split_bf16(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), batch, M, N, K, lda, ldb, N,
                                       A.shape[1] * lda, Bm.shape[1] * ldb, M * N, tA, tB, beta, S()))
    opA = (A.transpose(1, 2) if tA else A).double()
    opB = (Bm.transpose(1, 2) if tB else Bm).double()
    ref = opA @ opB + beta * C0.double()
    mag = opA.abs() @ opB.abs() + beta * C0.double().abs()
    err = (Cd.cpu().double() - ref).abs()
    assert torch.isfinite(Cd).all()
    assert float((err / mag).max()) < 2e-6, float((err / mag).max())


def test_bgemm_split_bf16_identity_asymmetric(lib):
    # A = I against an asymmetric B: catches a transposed C write or a k permutation that differs between the operands
    M = N = K = 160
    Bm = torch.arange(K * N, dtype=torch.float32).reshape(K, N) * 0.25
    Ad, Bd = dev(torch.eye(M).repeat(2, 1, 1)), dev(Bm)
    Cd = torch.zeros(2, M, N, device="cuda")
    _lib.check(lib.dp_bgemm_split_bf16(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), 2, M, N, K, K, N, N, M * K, 0, M * N,
                                       0, 0, 0.0, S()))
    close(Cd, Bm.repeat(2, 1, 1), 0, 0)


def test_bgemm_split_bf16_identity_narrow_tile(lib):
    # the same with a 60-column B: the 128 x 64 tile, one bf16 per LDS store on the row-contiguous operand
    M = K = 160
    N = 60
    Bm = torch.arange(K * N, dtype=torch.float32).reshape(K, N) * 0.25
    Ad, Bd = dev(torch.eye(M).repeat(2, 1, 1)), dev(Bm)
    Cd = torch.zeros(2, M, N, device="cuda")
    _lib.check(lib.dp_bgemm_split_bf16(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), 2, M, N, K, K, N, N, M * K, 0, M * N,
                                       0, 0, 0.0, S()))
    close(Cd, Bm.repeat(2, 1, 1), 0, 0)


def test_bgemm_asymmetric_identity_and_strides(lib):
    # A = I with an asymmetric B catches a transposed C-write; leading dims > widths; broadcast B (stride 0)
    M = N = K = 48
    A = torch.eye(M).repeat(2, 1, 1)
    Bm = (torch.arange(K * 70, dtype=torch.float32).reshape(K, 70) * 0.01)
    Cd = torch.zeros(2, M, 80, device="cuda")
    Ad, Bd = dev(A), dev(Bm)
    _lib.check(lib.dp_bgemm_f32(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), None, 2, M, N, K, K, 70, 80, M * K, 0,
                                M * 80, 0, 0, 1.0, 0.0, 1, S()))
    close(Cd[:, :, :N], Bm[:, :N].repeat(2, 1, 1), 0, 0)
    assert float(Cd[:, :, N:].abs().max()) == 0.0          # nothing written outside the N columns


# ------------------------------------------------------------------ adjacency aggregation (panel kernel)
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("B,n,C", [(3, 16, 5), (2, 100, 40), (20, 500, 70), (2, 500, 128), (1, 1024, 33),
                                   (2, 1100, 20), (2, 67, 9), (2, 260, 130), (2, 512, 276), (1, 1024, 256)])
def test_adj_aggregate(lib, trans, B, n, C):
    g = torch.Generator().manual_seed(n + C)
    adj = (torch.rand(B, n, n, generator=g) < 0.1).float() * torch.rand(B, n, n, generator=g)
    V = torch.randn(B, n, C + 3, generator=g)            # ldv > C
    U0 = torch.randn(B, n, C, generator=g)
    ad, Vd, Ud = dev(adj), dev(V), dev(U0)
    _lib.check(lib.dp_adj_aggregate(ad.data_ptr(), Vd.data_ptr(), C + 3, Ud.data_ptr(), C, B, n, C, trans, 0.5, S()))
    opA = adj.transpose(1, 2) if trans else adj
    ref = opA.double() @ V[:, :, :C].double() + 0.5 * U0.double()
    close(Ud, ref.float(), 1e-4, 1e-4)


def test_adj_aggregate_padding_is_inert(lib):
    # graphs padded with zero rows/columns and NaN-free garbage beyond n must not leak into the result
    B, n, C = 2, 36, 20
    adj = torch.zeros(B, n, n)
    adj[:, :30, :30] = (torch.rand(B, 30, 30) < 0.3).float()
    V = torch.randn(B, n, C)
    ad, Vd = dev(adj), dev(V)
    U = torch.full((B, n, C), 7.0, device="cuda")
    _lib.check(lib.dp_adj_aggregate(ad.data_ptr(), Vd.data_ptr(), C, U.data_ptr(), C, B, n, C, 0, 0.0, S()))
    close(U, adj @ V, 1e-5, 1e-5)
    assert float(U[:, 30:].abs().max()) == 0.0


# ------------------------------------------------------------------ packed (bf16-exact) adjacency path
def _packed_aggregate(lib, adj, V, C, trans, beta, U0):
    B, n = adj.shape[0], adj.shape[1]
    ad, Vd, Ud = dev(adj), dev(V), dev(U0)
    nb = lib.dp_adj_pack_bytes(B, n)
    pk = torch.empty(nb, device="cuda", dtype=torch.uint8)
    pkt = torch.empty(nb, device="cuda", dtype=torch.uint8)
    flag = torch.full((64,), 7, device="cuda", dtype=torch.int32)
    _lib.check(lib.dp_adj_pack(ad.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), B, n, S()))
    wsb = lib.dp_adj_aggregate_packed_workspace_bytes(B, n, C)
    ws = ws_of(wsb)
    _lib.check(lib.dp_adj_aggregate_packed(ad.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(),
                                           Vd.data_ptr(), V.shape[2], Ud.data_ptr(), C, B, n, C, trans, beta, 0,
                                           ws.data_ptr(), wsb, S()))
    if beta == 0.0:      # the split left in the workspace is reusable: a presplit call must give the same bits
        U2 = torch.empty_like(Ud)
        _lib.check(lib.dp_adj_aggregate_packed(ad.data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(),
                                               Vd.data_ptr(), V.shape[2], U2.data_ptr(), C, B, n, C, trans, beta, 1,
                                               ws.data_ptr(), wsb, S()))
        assert torch.equal(U2, Ud)
    return Ud, int(flag[0])


@pytest.mark.parametrize("B,n", [(2, 128), (3, 131), (2, 500), (1, 68), (2, 4)])
def test_adj_pack_images_are_the_bf16_bits_and_their_transpose(lib, B, n):
    """dp_adj_pack: P[b, r, c] = high 16 bits of A[b, r, c], Pt its transpose, rows padded with zeros to a multiple of
    8 elements; the flag word is 0 iff every entry is bf16-exact.  Bit-exact (integer) comparison."""
    g = torch.Generator().manual_seed(n)
    vals = torch.tensor([0.0, 1.0, 0.5, -3.0])
    adj = vals[torch.randint(0, 4, (B, n, n), generator=g)]
    ld = (n + 7) // 8 * 8
    nb = lib.dp_adj_pack_bytes(B, n)
    assert nb >= B * n * ld * 2
    for exact in (True, False):
        a = adj.clone()
        if not exact:
            a[B - 1, n - 1, n - 2] = 1.0 + 2.0 ** -20          # needs the low mantissa bits
        pk = torch.full((nb,), 0xAB, device="cuda", dtype=torch.uint8)
        pkt = torch.full((nb,), 0xCD, device="cuda", dtype=torch.uint8)
        flag = torch.full((64,), 7, device="cuda", dtype=torch.int32)
        _lib.check(lib.dp_adj_pack(dev(a).data_ptr(), pk.data_ptr(), pkt.data_ptr(), flag.data_ptr(), B, n, S()))
        bits = (a.contiguous().view(torch.int32) >> 16).to(torch.int16)
        want = torch.zeros(B, n, ld, dtype=torch.int16)
        want[:, :, :n] = bits
        want_t = torch.zeros(B, n, ld, dtype=torch.int16)
        want_t[:, :, :n] = bits.transpose(1, 2)
        got = pk[:B * n * ld * 2].view(torch.int16).view(B, n, ld).cpu()
        got_t = pkt[:B * n * ld * 2].view(torch.int16).view(B, n, ld).cpu()
        assert torch.equal(got, want)
        assert torch.equal(got_t, want_t)
        assert (int(flag[0]) == 0) == exact


@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("B,n,C", [(3, 128, 5), (20, 500, 40), (4, 500, 70), (2, 1024, 33), (2, 516, 128),
                                   (2, 200, 50), (3, 131, 17), (2, 512, 276), (1, 1024, 256), (2, 260, 130),
                                   (130, 500, 40), (64, 1024, 48), (2, 300, 320), (3, 129, 150), (1, 2100, 200)])
def test_packed_aggregate_binary_adjacency_is_fp32_exact(lib, trans, B, n, C):
    """0/1 adjacency: the bf16 x (hi+mid+lo) path must agree with the fp64 product to fp32 rounding — every
    product is exact, only the fp32 accumulation order differs."""
    g = torch.Generator().manual_seed(n * 3 + C)
    adj = (torch.rand(B, n, n, generator=g) < 0.05).float()          # not symmetric on purpose
    V = torch.randn(B, n, C + 1, generator=g) * torch.exp(torch.randn(B, n, C + 1, generator=g) * 3)  # wide range
    U0 = torch.randn(B, n, C, generator=g)
    Ud, flag = _packed_aggregate(lib, adj, V, C, trans, 0.25, U0)
    assert flag == 0
    opA = adj.transpose(1, 2) if trans else adj
    ref = opA.double() @ V[:, :, :C].double() + 0.25 * U0.double()
    close(Ud, ref.float(), 2e-6, 1e-5 * float(ref.abs().max()) / 10)


def test_packed_aggregate_other_bf16_exact_values(lib):
    # any bf16-representable weights qualify, not just 0/1
    B, n, C = 2, 256, 20
    g = torch.Generator().manual_seed(1)
    vals = torch.tensor([0.0, 0.5, 1.0, -2.0, 3.0, 0.0078125])
    adj = vals[torch.randint(0, 6, (B, n, n), generator=g)]
    V = torch.randn(B, n, C, generator=g)
    Ud, flag = _packed_aggregate(lib, adj, V, C, 0, 0.0, torch.zeros(B, n, C))
    assert flag == 0
    close(Ud, (adj.double() @ V.double()).float(), 1e-5, 1e-5)


@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("B,n,C", [(2, 260, 24), (2, 260, 150), (2, 258, 150), (140, 512, 24)])
def test_packed_aggregate_falls_back_when_not_bf16_exact(lib, trans, B, n, C):
    # a single entry with low mantissa bits set flips the device flag: the fp32 loop must run (and be right);
    # the wide shapes (C > 128, or >= 512 row tiles) take it as a flag-gated second launch
    g = torch.Generator().manual_seed(2)
    adj = (torch.rand(B, n, n, generator=g) < 0.1).float()
    adj[1, 200, 37] = 0.3                                     # not representable in bf16
    V = torch.randn(B, n, C, generator=g)
    Ud, flag = _packed_aggregate(lib, adj, V, C, trans, 0.0, torch.zeros(B, n, C))
    assert flag != 0
    opA = adj.transpose(1, 2) if trans else adj
    close(Ud, (opA.double() @ V.double()).float(), 1e-4, 1e-4)


# ------------------------------------------------------------------ A1 GraphConv
@pytest.mark.parametrize("add_self,bias,normalize", [(0, 1, 1), (1, 1, 1), (0, 0, 1), (1, 0, 0)])
@pytest.mark.parametrize("B,n,fin,fout", [(3, 16, 5, 8), (2, 100, 89, 20), (2, 37, 20, 50)])
def test_gcn_layer_fwd_bwd(lib, add_self, bias, normalize, B, n, fin, fout):
    x, adj, nn_, _ = O.make_batch(B, n, fin, n_min=1, p=0.2, seed=11, onehot=False)
    g = torch.Generator().manual_seed(5)
    adj = adj * torch.rand(adj.shape, generator=g)          # general (non-binary, non-symmetric) adjacency
    W = torch.randn(fin, fout, generator=g) * 0.3
    b = torch.randn(fout, generator=g) * 0.1 if bias else None
    dy = torch.randn(B, n, fout, generator=g)
    xo, ao, Wo = x.clone().requires_grad_(True), adj.clone().requires_grad_(True), W.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if bias else None
    yo = O.graph_conv(xo, ao, Wo, bo, bool(add_self), bool(normalize))
    (yo * dy).sum().backward()

    flags = (1 if add_self else 0) | (2 if normalize else 0)
    xd, ad, Wd, dyd = dev(x), dev(adj), dev(W), dev(dy)
    bd = dev(b) if bias else None
    y = torch.empty(B, n, fout, device="cuda")
    invn = torch.empty(B, n, device="cuda")
    wsb = lib.dp_gcn_layer_workspace_bytes(B, n, fin, fout)
    ws = ws_of(wsb)
    _lib.check(lib.dp_gcn_layer_fwd(xd.data_ptr(), fin, ad.data_ptr(), Wd.data_ptr(), _lib.ptr(bd), y.data_ptr(), fout,
                                    invn.data_ptr(), B, n, fin, fout, flags, ws.data_ptr(), wsb, S()))
    close(y, yo)
    dx, dadj = torch.empty_like(xd), torch.empty_like(ad)
    dW, db = torch.empty_like(Wd), torch.empty(fout, device="cuda")
    _lib.check(lib.dp_gcn_layer_bwd(xd.data_ptr(), fin, ad.data_ptr(), Wd.data_ptr(), y.data_ptr(), fout,
                                    invn.data_ptr(), dyd.data_ptr(), fout, dx.data_ptr(), fin, dW.data_ptr(),
                                    db.data_ptr(), dadj.data_ptr(), B, n, fin, fout, flags, ws.data_ptr(), wsb, S()))
    close(dx, xo.grad, 1e-3, 1e-5)
    close(dadj, ao.grad, 1e-3, 1e-5)
    close(dW, Wo.grad, 1e-3, 1e-4)
    if bias:
        close(db, bo.grad, 1e-3, 1e-4)


@pytest.mark.parametrize("name", [f"g1_graphconv_s{s}b{b}n{n}" for s in (0, 1) for b in (0, 1) for n in (1, 0)])
def test_gcn_layer_against_reference_golden_g1(lib, golden, name):
    """dp_gcn_layer_fwd/bwd against G1: the reference's own GraphConv text (encoders.py:944-974) run in the build
    container (oracle/make_golden.py P2) — add_self x bias x normalize, padded zero rows, zero OUTPUT rows."""
    a, p, g = golden(name)
    add_self, bias, normalize = [int(v) for v in a["cfg"]]
    x, adj, W = torch.from_numpy(a["x"]), torch.from_numpy(a["adj"]), p["weight"]
    B, n, fin = x.shape
    fout = W.shape[1]
    flags = (1 if add_self else 0) | (2 if normalize else 0)
    xd, ad, Wd, dyd = dev(x), dev(adj), dev(W), dev(torch.from_numpy(a["gy"]))
    bd = dev(p["bias"]) if bias else None
    y = torch.empty(B, n, fout, device="cuda")
    invn = torch.empty(B, n, device="cuda")
    wsb = lib.dp_gcn_layer_workspace_bytes(B, n, fin, fout)
    ws = ws_of(wsb)
    _lib.check(lib.dp_gcn_layer_fwd(xd.data_ptr(), fin, ad.data_ptr(), Wd.data_ptr(), _lib.ptr(bd), y.data_ptr(), fout,
                                    invn.data_ptr(), B, n, fin, fout, flags, ws.data_ptr(), wsb, S()))
    close(y, torch.from_numpy(a["y"]))
    dx, dadj = torch.empty_like(xd), torch.empty_like(ad)
    dW, db = torch.empty_like(Wd), torch.empty(fout, device="cuda")
    _lib.check(lib.dp_gcn_layer_bwd(xd.data_ptr(), fin, ad.data_ptr(), Wd.data_ptr(), y.data_ptr(), fout,
                                    invn.data_ptr(), dyd.data_ptr(), fout, dx.data_ptr(), fin, dW.data_ptr(),
                                    db.data_ptr() if bias else None, dadj.data_ptr(), B, n, fin, fout, flags,
                                    ws.data_ptr(), wsb, S()))
    # zero OUTPUT rows (bias off, padded rows): F.normalize's clamp gives d u = d y / 1e-12, so some gradient entries
    # are ~1e12 next to O(1) ones — huge entries are compared relatively, the rest with the usual atol
    def mixed(got, ref):
        got, ref = got.detach().cpu(), ref.detach().cpu()
        assert torch.isfinite(got).all()
        big = ref.abs() > 1e6
        torch.testing.assert_close(got[big], ref[big], rtol=1e-3, atol=0.0)
        torch.testing.assert_close(got[~big], ref[~big], rtol=1e-3, atol=1e-5)
    mixed(dx, torch.from_numpy(a["gx"]))
    mixed(dadj, torch.from_numpy(a["gadj"]))
    mixed(dW, g["weight"])
    if bias:
        mixed(db, g["bias"])


def test_gcn_layer_zero_rows_use_the_clamp_branch(lib):
    # rows with ||u|| < 1e-12 (padded rows with zero bias): y = 0 and d u = d y / 1e-12 (F.normalize's clamp_min)
    B, n, f = 1, 8, 4
    x = torch.zeros(B, n, f)
    x[0, :3] = torch.randn(3, f)
    adj = torch.zeros(B, n, n)
    adj[0, :3, :3] = 1.0
    W = torch.randn(f, f)
    xd, ad, Wd = dev(x), dev(adj), dev(W)
    y = torch.empty(B, n, f, device="cuda")
    invn = torch.empty(B, n, device="cuda")
    wsb = lib.dp_gcn_layer_workspace_bytes(B, n, f, f)
    ws = ws_of(wsb)
    _lib.check(lib.dp_gcn_layer_fwd(xd.data_ptr(), f, ad.data_ptr(), Wd.data_ptr(), None, y.data_ptr(), f,
                                    invn.data_ptr(), B, n, f, f, 2, ws.data_ptr(), wsb, S()))
    assert float(y[0, 3:].abs().max()) == 0.0
    assert float(invn[0, 3:].min()) == pytest.approx(1e12, rel=1e-6)
    yo = O.graph_conv(x, adj, W, None, False, True)
    close(y, yo)


# ------------------------------------------------------------------ A3 apply_bn
@pytest.mark.parametrize("relu", [0, 1])
def test_bn_node_fwd_bwd(lib, relu, golden):
    a, _, _ = golden("g2_apply_bn")
    x = torch.from_numpy(a["x"])
    B, n, F_ = x.shape
    gy = torch.from_numpy(a["gy"])
    xo = x.clone().requires_grad_(True)
    yo = O.bn_node(torch.relu(xo) if relu else xo)
    (yo * gy).sum().backward()
    if not relu:                                        # golden vector of the reference's own apply_bn
        close(yo, torch.from_numpy(a["y"]), 1e-5, 1e-6)
    xd, gyd = dev(x), dev(gy)
    y = torch.empty_like(xd)
    stats = torch.empty(n, 2, device="cuda")
    wsb = lib.dp_bn_node_workspace_bytes(B, n, F_)
    ws = ws_of(wsb)
    _lib.check(lib.dp_bn_node_fwd(xd.data_ptr(), F_, y.data_ptr(), F_, stats.data_ptr(), B, n, F_, relu,
                                  ws.data_ptr(), wsb, S()))
    close(y, yo, 1e-4, 2e-5)
    dx = torch.empty_like(xd)
    _lib.check(lib.dp_bn_node_bwd(xd.data_ptr(), F_, y.data_ptr(), F_, stats.data_ptr(), gyd.data_ptr(), F_,
                                  dx.data_ptr(), F_, B, n, F_, relu, ws.data_ptr(), wsb, S()))
    close(dx, xo.grad, 1e-3, 2e-4)


# ------------------------------------------------------------------ A5 assignment head
def test_assign_softmax_mask(lib):
    B, n, Din, K = 3, 20, 13, 7
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, n, Din, generator=g)
    Wp, bp = torch.randn(K, Din, generator=g) * 0.5, torch.randn(K, generator=g)
    nn_ = np.array([20, 1, 9], dtype=np.int32)
    dS = torch.randn(B, n, K, generator=g)
    zo, Wo, bo = z.clone().requires_grad_(True), Wp.clone().requires_grad_(True), bp.clone().requires_grad_(True)
    So = torch.softmax(zo @ Wo.t() + bo, dim=-1) * O.node_mask(n, nn_)
    (So * dS).sum().backward()
    zd, Wd, bd, dSd = dev(z), dev(Wp), dev(bp), dev(dS)
    nd = torch.from_numpy(nn_).cuda()
    Sd = torch.empty(B, n, K, device="cuda")
    wsb = lib.dp_assign_workspace_bytes(B, n, Din, K)
    ws = ws_of(wsb)
    _lib.check(lib.dp_assign_softmax_mask_fwd(zd.data_ptr(), Din, Wd.data_ptr(), bd.data_ptr(), nd.data_ptr(),
                                              Sd.data_ptr(), B, n, Din, K, ws.data_ptr(), wsb, S()))
    close(Sd, So)
    assert float(Sd[1, 1:].abs().max()) == 0.0
    dz, dW, db = torch.empty_like(zd), torch.empty_like(Wd), torch.empty_like(bd)
    _lib.check(lib.dp_assign_softmax_mask_bwd(zd.data_ptr(), Din, Wd.data_ptr(), Sd.data_ptr(), dSd.data_ptr(),
                                              nd.data_ptr(), dz.data_ptr(), Din, dW.data_ptr(), db.data_ptr(), B, n,
                                              Din, K, ws.data_ptr(), wsb, S()))
    close(dz, zo.grad, 1e-3, 1e-5)
    close(dW, Wo.grad, 1e-3, 1e-5)
    close(db, bo.grad, 1e-3, 1e-5)


# ------------------------------------------------------------------ A6 pooling
@pytest.mark.parametrize("B,n,K,D", [(3, 16, 4, 9), (2, 100, 10, 60), (1, 130, 33, 70)])
def test_pool_fwd_bwd(lib, B, n, K, D):
    g = torch.Generator().manual_seed(n)
    Sm = torch.softmax(torch.randn(B, n, K, generator=g), -1)
    Z = torch.randn(B, n, D, generator=g)
    adj = (torch.rand(B, n, n, generator=g) < 0.2).float() * torch.rand(B, n, n, generator=g)   # non-symmetric
    dX, dA = torch.randn(B, K, D, generator=g), torch.randn(B, K, K, generator=g)
    So, Zo, Ao = (t.clone().requires_grad_(True) for t in (Sm, Z, adj))
    Xp = So.transpose(1, 2) @ Zo
    Ap = So.transpose(1, 2) @ Ao @ So
    ((Xp * dX).sum() + (Ap * dA).sum()).backward()
    Sd, Zd, Ad, dXd, dAd = dev(Sm), dev(Z), dev(adj), dev(dX), dev(dA)
    Xd, Apd, Td = (torch.empty(B, K, D, device="cuda"), torch.empty(B, K, K, device="cuda"),
                   torch.empty(B, K, n, device="cuda"))
    _lib.check(lib.dp_pool_fwd(Sd.data_ptr(), Zd.data_ptr(), D, Ad.data_ptr(), Xd.data_ptr(), Apd.data_ptr(),
                               Td.data_ptr(), B, n, K, D, S()))
    close(Xd, Xp, 1e-4, 1e-4)
    close(Apd, Ap, 1e-4, 1e-4)
    dS = torch.empty_like(Sd)
    dZ = torch.zeros_like(Zd)
    dadj = torch.zeros_like(Ad)
    wsb = lib.dp_pool_bwd_workspace_bytes(B, n, K, D)
    ws = ws_of(wsb)
    _lib.check(lib.dp_pool_bwd(Sd.data_ptr(), Zd.data_ptr(), D, Ad.data_ptr(), Td.data_ptr(), dXd.data_ptr(),
                               dAd.data_ptr(), dS.data_ptr(), dZ.data_ptr(), D, dadj.data_ptr(), B, n, K, D,
                               ws.data_ptr(), wsb, S()))
    close(dS, So.grad, 1e-3, 1e-3)
    close(dZ, Zo.grad, 1e-3, 1e-4)
    close(dadj, Ao.grad, 1e-3, 1e-4)


# ------------------------------------------------------------------ A7 readout
def test_masked_max(lib):
    B, n, F_ = 4, 23, 70
    g = torch.Generator().manual_seed(2)
    Z = torch.randn(B, n, F_, generator=g)
    Z[0, :, :5] = -Z[0, :, :5].abs() - 0.1      # all-negative columns: a masked zero row must win
    Z[1, 2, 7] = Z[1, 5, 7] = 9.0               # tie: first index wins
    Z[2, 0, 3] = 0.0
    Z[2, 1:, 3] = -1.0                          # valid max exactly 0 ties with the masked zeros: valid row wins
    nn_ = np.array([10, 23, 4, 1], dtype=np.int32)
    dout = torch.randn(B, F_, generator=g)
    Zo = Z.clone().requires_grad_(True)
    out_o, _ = (Zo * O.node_mask(n, nn_)).max(dim=1)
    (out_o * dout).sum().backward()
    Zd, dd = dev(Z), dev(dout)
    nd = torch.from_numpy(nn_).cuda()
    out = torch.empty(B, F_, device="cuda")
    arg = torch.empty(B, F_, device="cuda", dtype=torch.int32)
    _lib.check(lib.dp_masked_max_fwd(Zd.data_ptr(), F_, nd.data_ptr(), out.data_ptr(), F_, arg.data_ptr(), B, n, F_,
                                     S()))
    close(out, out_o, 0, 0)
    assert int(arg[1, 7]) == 2 and int(arg[0, 0]) == -1 and int(arg[2, 3]) == 0
    dZ = torch.zeros_like(Zd)
    _lib.check(lib.dp_masked_max_bwd(dd.data_ptr(), F_, arg.data_ptr(), dZ.data_ptr(), F_, B, n, F_, S()))
    close(dZ, Zo.grad, 0, 0)
    # no mask at all (base encoder / pooled levels)
    _lib.check(lib.dp_masked_max_fwd(Zd.data_ptr(), F_, None, out.data_ptr(), F_, arg.data_ptr(), B, n, F_, S()))
    close(out, Z.max(dim=1)[0], 0, 0)


# ------------------------------------------------------------------ A8 link loss
@pytest.mark.parametrize("B,n,K", [(3, 16, 4), (2, 100, 10), (3, 200, 20), (2, 150, 40), (2, 132, 50), (2, 130, 90),
                                   (1, 200, 128), (1, 140, 150), (1, 260, 256)])
def test_linkpred_loss(lib, B, n, K):
    g = torch.Generator().manual_seed(K)
    _, adj, nn_, _ = O.make_batch(B, n, 3, n_min=1, p=0.2, seed=K)
    Sm = torch.softmax(torch.randn(B, n, K, generator=g) * 2, -1) * O.node_mask(n, nn_)
    Sm[0, 0] = 0.0
    Sm[0, 0, 1] = 1.0        # a one-hot row: (S S^T)_00 == 1 exactly -> the tie branch of torch.min
    So = Sm.clone().requires_grad_(True)
    lo = O.link_pred_loss(So, adj, nn_)
    (lo * 1.7).backward()
    Sd, Ad = dev(Sm), dev(adj)
    nd = torch.from_numpy(nn_).cuda()
    loss = torch.empty(1, device="cuda")
    wsb = lib.dp_linkpred_workspace_bytes(B, n, K)
    ws = ws_of(wsb)
    _lib.check(lib.dp_linkpred_loss_fwd(Sd.data_ptr(), Ad.data_ptr(), nd.data_ptr(), loss.data_ptr(), B, n, K,
                                        ws.data_ptr(), wsb, S()))
    close(loss[0], lo, 1e-5, 1e-6)
    dS = torch.empty_like(Sd)
    dl = torch.tensor([1.7], device="cuda")
    _lib.check(lib.dp_linkpred_loss_bwd(Sd.data_ptr(), Ad.data_ptr(), nd.data_ptr(), dl.data_ptr(), dS.data_ptr(), 0,
                                        B, n, K, ws.data_ptr(), wsb, S()))
    close(dS, So.grad, 1e-3, 1e-5)


def test_cross_entropy(lib):
    B, Cc = 20, 6
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(B, Cc, generator=g) * 3
    label = torch.randint(0, Cc, (B,), generator=g)
    lo_in = logits.clone().requires_grad_(True)
    lo = torch.nn.functional.cross_entropy(lo_in, label)
    lo.backward()
    ld, yd = dev(logits), label.cuda()
    loss = torch.empty(1, device="cuda")
    prob = torch.empty(B, Cc, device="cuda")
    _lib.check(lib.dp_cross_entropy_fwd(ld.data_ptr(), yd.data_ptr(), loss.data_ptr(), prob.data_ptr(), B, Cc, S()))
    close(loss[0], lo, 1e-5, 1e-6)
    dl = torch.empty_like(ld)
    _lib.check(lib.dp_cross_entropy_bwd(prob.data_ptr(), yd.data_ptr(), None, dl.data_ptr(), B, Cc, S()))
    close(dl, lo_in.grad, 1e-4, 1e-6)


# ------------------------------------------------------------------ A9 mean aggregator
def test_mean_aggregator_golden(lib, golden):
    a, _, _ = golden("g8_mean_aggregator")
    from graph_pooling_amd.aggregators import mean_aggregate
    table = torch.from_numpy(a["table"]).cuda().requires_grad_(True)
    ip = torch.from_numpy(a["indptr"].astype(np.int32)).cuda()
    idx = torch.from_numpy(a["indices"].astype(np.int32)).cuda()
    out = mean_aggregate(table, ip, idx)
    close(out, torch.from_numpy(a["out"]), 1e-5, 1e-6)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(0))
    (out * g.cuda()).sum().backward()
    to = torch.from_numpy(a["table"]).requires_grad_(True)
    neighs = [a["indices"][a["indptr"][i]:a["indptr"][i + 1]].tolist() for i in range(len(a["indptr"]) - 1)]
    (O.mean_aggregate(to, a["nodes"].tolist(), neighs) * g).sum().backward()
    close(table.grad, to.grad, 1e-4, 1e-6)


def _g8_sets(a):
    return [set(a["indices"][a["indptr"][i]:a["indptr"][i + 1]].tolist()) for i in range(len(a["indptr"]) - 1)]


def test_mean_aggregator_module_against_reference_golden(golden):
    """MeanAggregator.forward itself (aggregators.py:30-63, num_sample=None): host neighbour SETS -> CSR, the `features`
    callable looked up with a LongTensor of the unique nodes on the device, then the gather kernel — against G8, which
    the reference's own class produced; gradient through the embedding table against the oracle."""
    a, _, _ = golden("g8_mean_aggregator")
    from graph_pooling_amd.aggregators import MeanAggregator
    emb = torch.nn.Embedding(a["table"].shape[0], a["table"].shape[1]).cuda()
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(a["table"]))
    seen = []

    def features(ids):
        assert ids.dtype == torch.int64 and ids.is_cuda
        seen.append(ids)
        return emb(ids)
    agg = MeanAggregator(features, cuda=True, gcn=False)
    out = agg(a["nodes"].tolist(), _g8_sets(a), num_sample=None)
    close(out, torch.from_numpy(a["out"]), 1e-5, 1e-6)
    assert len(seen) == 1 and len(set(seen[0].tolist())) == seen[0].numel()      # one lookup, unique nodes only
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    (out * g.cuda()).sum().backward()
    to = torch.from_numpy(a["table"]).requires_grad_(True)
    (O.mean_aggregate(to, a["nodes"].tolist(), [sorted(s) for s in _g8_sets(a)]) * g).sum().backward()
    close(emb.weight.grad, to.grad, 1e-4, 1e-6)


def test_mean_aggregator_module_gcn_PARITY_UNPINNED(golden):
    """gcn=True (the node itself joins its neighbour set): the reference's line raises (`set + set`, aggregators.py:47,
    SURVEY Appendix B D10), so this is pinned to the oracle's restatement of the intended semantics only."""
    a, _, _ = golden("g8_mean_aggregator")
    from graph_pooling_amd.aggregators import MeanAggregator
    table = torch.from_numpy(a["table"]).cuda()
    agg = MeanAggregator(lambda ids: table[ids], cuda=True, gcn=True)
    nodes = a["nodes"].tolist()
    out = agg(nodes, _g8_sets(a), num_sample=None)
    ref = O.mean_aggregate(torch.from_numpy(a["table"]), nodes, [sorted(s) for s in _g8_sets(a)], gcn=True)
    close(out, ref, 1e-5, 1e-6)
    assert float((out.cpu() - torch.from_numpy(a["out"])).abs().max()) > 1e-3       # and it differs from gcn=False


def test_mean_aggregator_module_sampling_matches_the_oracle_on_the_same_samples(golden):
    """num_sample=3 (aggregators.py:38-42): sets with >= 3 neighbours are sub-sampled with `random.sample`.  The
    reference's draw is unseeded, so the check replays the module's draws: same `random.seed`, same call order, and
    the oracle is fed the sampled sets."""
    import random
    a, _, _ = golden("g8_mean_aggregator")
    from graph_pooling_amd.aggregators import MeanAggregator
    table = torch.from_numpy(a["table"]).cuda()
    agg = MeanAggregator(lambda ids: table[ids], cuda=True, gcn=False)
    nodes, sets = a["nodes"].tolist(), _g8_sets(a)
    random.seed(1234)
    out = agg(nodes, sets, num_sample=3)
    random.seed(1234)
    sampled = [sorted(random.sample(sorted(s), 3)) if len(s) >= 3 else sorted(s) for s in sets]
    assert all(len(s) <= 3 for s in sampled) and any(len(s) < len(t) for s, t in zip(sampled, sets))
    close(out, O.mean_aggregate(torch.from_numpy(a["table"]), nodes, sampled), 1e-5, 1e-6)


def test_supervised_graphsage_head_on_a_stub_encoder():
    """graphsage.py:7-26 with the D10 repairs: scores = enc(nodes) @ weight (the library's GEMM), cross-entropy on the
    raw scores; gradients reach the head's weight and the encoder."""
    from graph_pooling_amd.graphsage import SupervisedGraphSage

    class Enc(torch.nn.Module):
        embed_dim = 5

        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.randn(11, 5))

        def forward(self, nodes):
            return self.table[torch.as_tensor(nodes, device=self.table.device)]

    torch.manual_seed(0)
    head = SupervisedGraphSage(3, Enc()).cuda()
    nodes, labels = [0, 4, 7, 10], torch.tensor([[0], [2], [1], [2]]).cuda()
    scores = head(nodes)
    assert tuple(scores.shape) == (4, 3)
    t, w = head.enc.table.detach().cpu().requires_grad_(True), head.weight.detach().cpu().requires_grad_(True)
    ref_scores = t[torch.tensor(nodes)] @ w
    close(scores, ref_scores, 1e-5, 1e-6)
    ref = torch.nn.functional.cross_entropy(ref_scores, labels.cpu().squeeze())
    loss = head.loss(nodes, labels)
    close(loss, ref, 1e-5, 1e-6)
    loss.backward()
    ref.backward()
    close(head.weight.grad, w.grad, 1e-4, 1e-6)
    close(head.enc.table.grad, t.grad, 1e-4, 1e-6)
    with pytest.raises(RuntimeError, match="GPU"):          # a CPU encoder output is refused, not computed
        SupervisedGraphSage(3, Enc())([0, 1])
