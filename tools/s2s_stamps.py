"""Cycle split of the Set2Set recurrence kernels (DP_STAMP build; DP_LIB=...stamp.so PYTHONPATH=. python tools/s2s_stamps.py):
cycles per phase of workgroup 0, summed over the n steps, forward and backward, at the S-S2S shape."""
import ctypes as C
import torch
import bench
from graph_pooling_amd import _lib

lib = _lib.load()
w = bench.WORKLOADS["enzymes_s2s"]
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))
for _ in range(3):
    model.zero_grad(set_to_none=True)
    y = model(batch["x"], batch["adj"], batch["nn"])
    model.loss(y, batch["label"]).backward()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.dp_debug_s2s_stamps.restype = C.c_int
assert lib.dp_debug_s2s_stamps(buf) == 0
n = w["N"]
for k, (label, names) in enumerate((("forward", ["gates (+ QP save)", "LSTM cell (+ saves)", "e = emb h", "softmax (+ Aw save)", "r = a^T emb"]),
                                    ("backward", ["da = emb dr (+ DR save, prefetch)", "de, DE save", "dh += de^T emb", "cell backward (+ DG save)", "[dh, dr] = Wt dg"]))):
    t = [buf[k * 16 + i] for i in range(5)]
    print(f"--- {label}: {sum(t)} cycles over {n} steps = {sum(t) // n} per step")
    for nm, v in zip(names, t):
        print(f"  {nm:36s} {v // n:7d} per step")
