"""Phase timing inside the pooled-level kernels (diagnostic build: DP_STAMP=1 graph_pooling_amd/csrc/build.sh, then
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. python tools/small_kernel_stamps.py).
Prints shader-clock deltas between the phase stamps of workgroup 0 of the LAST forward / backward small-level launch."""
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
fw = [buf[i] for i in range(5)]
bw = [buf[32 + i] for i in range(8)]
names_f = ["stage+bn", "P=XW", "U=AP", "normalise"]
names_b = ["stage", "bn sums", "dU rows", "G=A^T dU (+P, db)", "dW", "dX", "dA'", "part2_prev"]
print("forward  (cycles):", {n: fw[i + 1] - fw[i] for i, n in enumerate(names_f)}, "total", fw[-1] - fw[0])
print("backward staging: issue", buf[32 + 8] - bw[0], "wait+barrier", buf[32 + 9] - buf[32 + 8], "bn sums", bw[1] - buf[32 + 9])
print("backward G phase: to-db-end", buf[32 + 10] - bw[2], "G tile", buf[32 + 11] - buf[32 + 10], "P (wave 0: none)", buf[32 + 12] - buf[32 + 11], "barrier wait", bw[3] - buf[32 + 12])
print("backward (cycles):", {n: bw[i + 1] - bw[i] for i, n in enumerate(names_b[:-1])}, "total", bw[-1] - bw[0])
