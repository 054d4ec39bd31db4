"""Per-shape rate of the split-bf16 GEMM against the fp32-MFMA GEMM at the ER model's big contractions (B = 256).
PYTHONPATH=. python tools/gemm_split_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from graph_pooling_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def run(name, M, N, K, tA, tB, batch=256):
    A = torch.randn(batch, K, M, device="cuda") if tA else torch.randn(batch, M, K, device="cuda")
    B = torch.randn(batch, N, K, device="cuda") if tB else torch.randn(batch, K, N, device="cuda")
    C = torch.empty(batch, M, N, device="cuda")
    bias = torch.zeros(N, device="cuda")
    lda, ldb = A.shape[2], B.shape[2]
    sA, sB = A.shape[1] * lda, B.shape[1] * ldb

    def split():
        lib.dp_bgemm_split_bf16(A.data_ptr(), B.data_ptr(), C.data_ptr(), batch, M, N, K, lda, ldb, N, sA, sB, M * N, tA, tB,
                                0.0, st)

    def fp32():        # alpha != 1 keeps the call on the fp32-MFMA kernel (the split kernel takes alpha = 1 only)
        lib.dp_bgemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr(), batch, M, N, K, lda, ldb, N, sA, sB,
                         M * N, tA, tB, 0.5, 0.0, 0, st)
    out = []
    for fn in (split, fp32):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        out.append((us, 2.0 * batch * M * N * K / us / 1e6))
    print(f"{name:30s} split {out[0][0]:8.1f} us {out[0][1]:7.1f} TF | fp32 MFMA {out[1][0]:8.1f} us {out[1][1]:7.1f} TF "
          f"| x{out[1][0] / out[0][0]:.2f}")


run("A' = Tt^T S  TN 256x256x1024", 256, 256, 1024, 1, 0)
run("T dA'        NN 1024x256x256", 1024, 256, 256, 0, 0)
run("S dA'^T      NT 1024x256x256", 1024, 256, 256, 0, 1)
run("logits       NT 1024x256x296", 1024, 256, 296, 0, 1)
run("dZa          NN 1024x296x256", 1024, 296, 256, 0, 0)
run("dWp          TN 256x296x1024", 256, 296, 1024, 1, 0)
run("dS = Z dX'^T NT 1024x256x60", 1024, 256, 60, 0, 1)
run("A'=T^T S lvl1 TN 256x256x40?", 256, 256, 40, 0, 1)
run("dZ += S dX'  NN 1024x60x256", 1024, 60, 256, 0, 0)
run("X' = S^T Z   TN 256x60x1024", 256, 60, 1024, 1, 0)
