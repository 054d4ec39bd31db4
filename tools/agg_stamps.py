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
buf = (C.c_ulonglong * 16)()
lib.dp_debug_agg_stamps.restype = C.c_int
assert lib.dp_debug_agg_stamps(buf) == 0
names = ["prologue", "V prefetch + panel DMA issue", "barrier (DMA wait)", "multiply loop", "barrier", "reduce store + barrier",
         "sum + store"]
t = [buf[i] for i in range(8)]
print({n: t[i + 1] - t[i] for i, n in enumerate(names)}, "total", t[7] - t[0])
