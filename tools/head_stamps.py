"""Phase timing inside the fused head kernels (diagnostic build: DP_STAMP=1 csrc/build.sh, run with
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so): workgroup 0 of the last forward / backward of a warm DD step."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graph_pooling_amd import _lib  # noqa: E402

lib = _lib.load()
w = bench.WORKLOADS["dd"]
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))
model.train()
for _ in range(20):
    y = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
    loss = model.loss(y, batch["label"])
    loss.backward()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib.dp_debug_head_stamps.restype = C.c_int
assert lib.dp_debug_head_stamps(buf) == 0
t = list(buf)
print("forward  (cycles): staging issue+commit", t[1] - t[0], "| readout + barrier", t[2] - t[1], "| MLP", t[3] - t[2],
      "| total", t[3] - t[0])
print("backward (cycles): staging", t[9] - t[8], "| upper layers", t[10] - t[9], "| first-layer dW + bias", t[11] - t[10],
      "| d(features) + scatter", t[12] - t[11], "| total", t[12] - t[8])
