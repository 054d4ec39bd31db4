"""Measure the per-kernel cost of tiny dependent kernels under hipGraph replay and in eager mode (no profiler)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_pooling_amd import _lib
lib = _lib.load()
B, Cc = 20, 6
logits = torch.randn(B, Cc, device="cuda"); label = torch.randint(0, Cc, (B,), device="cuda")
loss = torch.empty(1, device="cuda"); prob = torch.empty(B, Cc, device="cuda")
def chain(n):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        lib.dp_cross_entropy_fwd(logits.data_ptr(), label.data_ptr(), loss.data_ptr(), prob.data_ptr(), B, Cc, st)
for n in (50, 200):
    chain(n); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): chain(n)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20 / n * 1e6
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(n)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(n)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50 / n * 1e6
    print(f"n={n}: eager {eager:.2f} us/kernel, graph {graph:.2f} us/kernel")
