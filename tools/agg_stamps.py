"""Phase timing inside the panel aggregation kernel (diagnostic build, see tools/small_kernel_stamps.py): the DD probe
launch (B=20, N=500, C=40, packed path), workgroup 0."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graph_pooling_amd import _lib

lib = _lib.load()
w = bench.WORKLOADS["dd"]
print(bench.roofline_probe(w, torch.device("cuda"), iters=20)["us_per_launch"], "us per launch (events)")
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))
model.train()
with torch.no_grad():
    for _ in range(30):                      # warm: the stamps of the last launches are the ones read back
        model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.dp_debug_agg_stamps.restype = C.c_int
assert lib.dp_debug_agg_stamps(buf) == 0
names = ["prologue", "V prefetch + panel DMA issue", "barrier (DMA wait)", "multiply loop", "barrier", "reduce store + barrier",
         "sum + store"]
t = [buf[i] for i in range(8)]
print("plain-store launch (A^T S of the forward pass, workgroup 0):")
print({n: t[i + 1] - t[i] for i, n in enumerate(names)}, "total", t[7] - t[0])
t = [buf[16 + i] for i in range(9)]
print("fused-tail launch (last forward GraphConv of the level, workgroup 0):")
print({n: t[i + 1] - t[i] for i, n in enumerate(names[:-1] + ["sum + bias/self + tile", "row tail"])}, "total", t[8] - t[0])
