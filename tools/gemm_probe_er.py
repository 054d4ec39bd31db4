"""Time the ER-shaped fp32 contractions through dp_bgemm_f32 (HIP events): TFLOP/s against the 157 TFLOP/s fp32 MFMA peak."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_pooling_amd import _lib
lib = _lib.load()
def t(name, batch, M, N, K, tA=0, tB=0, iters=20):
    A = torch.randn(batch, K if tA else M, M if tA else K, device='cuda')
    B = torch.randn(batch, N if tB else K, K if tB else N, device='cuda')
    C = torch.empty(batch, M, N, device='cuda')
    st = torch.cuda.current_stream()
    def go():
        _lib.check(lib.dp_bgemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, batch, M, N, K, A.shape[2], B.shape[2], N,
            A.shape[1]*A.shape[2], B.shape[1]*B.shape[2], M*N, tA, tB, 1.0, 0.0, 0, st.cuda_stream))
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): go()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1)*1000/iters
    print(f"{name:24s} b={batch} M={M:4d} N={N:4d} K={K:4d} tA={tA} tB={tB}: {us:8.1f} us  {2.0*batch*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)
t("A' = Tt^T S", 256, 256, 256, 1024, 1, 0)
t("X' = S^T Z", 256, 256, 192, 1024, 1, 0)
t("dS = Z dX'^T", 256, 1024, 256, 192, 0, 1)
t("dZ = S dX'", 256, 1024, 192, 256, 0, 0)
t("dS += T dA'", 256, 1024, 256, 256, 0, 0)
t("X W", 256, 1024, 64, 64, 0, 0)
t("level1 A'X", 256, 256, 64, 256, 0, 0)
