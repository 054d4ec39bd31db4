"""Achieved bandwidth of the CSR neighbour aggregation (dp_csr_aggregate: MeanAggregator, aggregators.py:50-62, and
the A x of the CSR GraphConv).  Algorithmic bytes = nnz * feat * 4 (gathered rows) + rows * feat * 4 (output) + index
bytes.  PYTHONPATH=. python tools/csr_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from graph_pooling_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def run(name, n, deg, feat, iters=50):
    g = torch.Generator(device="cuda").manual_seed(0)
    indices = torch.randint(0, n, (n * deg,), device="cuda", generator=g, dtype=torch.int32)
    indptr = (torch.arange(n + 1, device="cuda", dtype=torch.int64) * deg).to(torch.int32)
    table = torch.randn(n, feat, device="cuda", generator=g)
    out = torch.empty(n, feat, device="cuda")
    call = lambda: lib.dp_csr_aggregate(table.data_ptr(), feat, indptr.data_ptr(), indices.data_ptr(), out.data_ptr(), feat,
                                        n, feat, 1, 0.0, st)
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    alg = n * deg * feat * 4 + n * feat * 4 + n * deg * 4
    print(f"{name:44s} {us:9.1f} us  {alg / us / 1e3:8.1f} GB/s algorithmic (gathered rows + output + indices)")


run("DD largest graph: n=5748 deg=5 feat=89", 5748, 5, 89)
run("n=65536 deg=10 feat=64 (table 16 MB: L2/MALL)", 65536, 10, 64)
run("n=1048576 deg=10 feat=64 (table 268 MB: HBM)", 1048576, 10, 64)
run("n=262144 deg=32 feat=128", 262144, 32, 128)
