"""Phase timing inside the whole-level forward kernel (diagnostic build: DP_STAMP=1 graph_pooling_amd/csrc/build.sh, then
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. python tools/level_kernel_stamps.py)."""
import ctypes as C
import torch
import bench
from graph_pooling_amd import _lib

lib = _lib.load()
w = bench.WORKLOADS["dd"]
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))
for _ in range(5):
    model.zero_grad(set_to_none=True)
    y = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
    model.loss(y, batch["label"]).backward()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
lib.dp_debug_stamps.restype = C.c_int
assert lib.dp_debug_stamps(buf) == 0
t = [buf[i] for i in range(32)]
print("entry->issued", t[8] - t[7], "wait", t[9] - t[8])
for l in range(3):
    o = 10 + 6 * l
    names = ["P=XW", "U=AP", "normalise", "barrier", "load partials", "stats+X"]
    prev = t[9] if l == 0 else t[o - 1]
    row = {}
    for i, nm in enumerate(names):
        if t[o + i] == 0 or (l == 2 and i > 2):
            break
        row[nm] = t[o + i] - prev
        prev = t[o + i]
    print("layer", l, row)
print("layer 2 U=AP: wave 0 mma done after", t[30] - t[22], "of", t[23] - t[22])
print("total", t[24] - t[7])
