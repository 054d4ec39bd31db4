"""Where the time of the step's GEMM launches goes (diagnostic build: DP_STAMP=1 csrc/build.sh, run with
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so).  Per launch of one eager DD step: the span from the first
workgroup's start to the last one's end, and the phases of workgroup (0,0), in microseconds (100 MHz counter)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graph_pooling_amd import _lib  # noqa: E402

lib = _lib.load()
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "dd"]
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))
model.train()
for _ in range(5):
    y = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
    loss = model.loss(y, batch["label"])
    loss.backward()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
lib.dp_debug_gemm_stamps.restype = C.c_int
assert lib.dp_debug_gemm_stamps(buf) == 0
rows = [[buf[i * 8 + j] for j in range(8)] for i in range(64)]
rows = [r for r in rows if r[1] and r[0] != 2 ** 64 - 1]
rows.sort(key=lambda r: r[0])
step = rows[-10:]
print("problem 0 (MxNxK), problems | span us | wg0: entry->loads issued | ->first slabs in LDS | ->K loop done | ->stored")
for r in step:
    ident = r[7]
    m, n, k, cnt = ident & 0xFFFF, (ident >> 16) & 0xFFFF, (ident >> 32) & 0xFFFFFF, ident >> 56
    f = lambda a, b: (r[b] - r[a]) / 100.0 if r[a] and r[b] else float("nan")
    print(f"{m:4d}x{n:3d}x{k:3d} x{cnt} | {(r[1] - r[0]) / 100.0:6.2f} | {f(2, 3):5.2f} | {f(3, 4):5.2f} | {f(4, 5):5.2f} | {f(5, 6):5.2f}"
          f" | wg0 starts {(r[2] - r[0]) / 100.0:5.2f} after the first, ends {(r[1] - r[6]) / 100.0 if r[6] else float('nan'):5.2f} before the last")
