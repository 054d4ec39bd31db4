# scratch: time the main contraction shapes of the DD step through dp_bgemm_f32 (HIP events)
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_pooling_amd import _lib
lib = _lib.load()
def t(name, batch, M, N, K, tA=0, tB=0, iters=100):
    A = torch.randn(batch, K if tA else M, M if tA else K, device='cuda')
    B = torch.randn(batch, N if tB else K, K if tB else N, device='cuda')
    C = torch.empty(batch, M, N, device='cuda')
    st = torch.cuda.current_stream()
    def go():
        _lib.check(lib.dp_bgemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, batch, M, N, K, A.shape[2], B.shape[2], N,
            A.shape[1]*A.shape[2], B.shape[1]*B.shape[2], M*N, tA, tB, 1.0, 0.0, 0, st.cuda_stream))
    for _ in range(10): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): go()
    e1.record(st); e1.synchronize()
    print(f"{name:28s} M={M:4d} N={N:4d} K={K:4d} tA={tA} tB={tB}: {e0.elapsed_time(e1)*1000/iters:7.2f} us")
t("agg fwd A.P", 20, 500, 40, 500)
t("agg fwd A.P (70)", 20, 500, 70, 500)
t("agg bwd At.dU", 20, 500, 40, 500, 1, 0)
t("T = St.A", 20, 50, 500, 500, 1, 0)
t("transform X.W", 20, 500, 20, 89)
t("dW = Xt.G", 20, 89, 20, 500, 1, 0)
t("dxin = G.Wt", 20, 500, 20, 20, 0, 1)
t("level1 agg", 20, 50, 20, 50)
def ta(name, B, n, C, trans, iters=100):
    A = (torch.rand(B, n, n, device='cuda') < 0.02).float()
    V = torch.randn(B, n, C, device='cuda'); U = torch.empty(B, n, C, device='cuda')
    st = torch.cuda.current_stream()
    def go():
        _lib.check(lib.dp_adj_aggregate(A.data_ptr(), V.data_ptr(), C, U.data_ptr(), C, B, n, C, trans, 0.0, st.cuda_stream))
    for _ in range(10): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): go()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1)*1000/iters
    print(f"{name:28s} B={B} n={n} C={C} trans={trans}: {us:7.2f} us  {B*n*n*4/us/1e3:7.1f} GB/s(A only)")
ta("panel NN", 20, 500, 40, 0); ta("panel NN", 20, 500, 70, 0); ta("panel TN", 20, 500, 40, 1); ta("panel TN 50", 20, 500, 50, 1)
ta("panel NN B=160", 160, 500, 40, 0); ta("panel TN B=160", 160, 500, 40, 1)
ta("panel NN ER", 64, 1024, 40, 0); ta("panel TN ER", 64, 1024, 40, 1)
