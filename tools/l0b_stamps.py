"""Phase timing inside the persistent level-0 BACKWARD kernel (DP_STAMP build, see tools/l0_stamps.py)."""
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
buf = (C.c_ulonglong * 192)()
lib.dp_debug_l0b_stamps.restype = C.c_int
assert lib.dp_debug_l0b_stamps(buf) == 0
names = {1: "rows, dX', dA', dZe staged + sync", 2: "4 pooling products + sync", 3: "V split, graph barrier, A rows in LDS",
         4: "A V aggregate + sync", 5: "reduce + softmax bwd + sync", 6: "dWp, dbp, dZa (+ stores)"}
for i in range(3):
    o = 8 + 8 * i
    names.update({o: f"layer {2 - i}: rownorm bwd + sync", o + 1: f"layer {2 - i}: bias sums, split, x_in / W staged",
                  o + 2: f"layer {2 - i}: graph barrier (+ A^T rows)", o + 3: f"layer {2 - i}: A^T dU aggregate + sync",
                  o + 4: f"layer {2 - i}: reduce to G + sync", o + 5: f"layer {2 - i}: dW, dx_in + sync",
                  o + 6: f"layer {2 - i}: dW stored, BN partials", o + 7: f"layer {2 - i}: exchange (polled at the next phase) / sync"})
names.update({40: "graph barrier (gradients)", 41: "combine + sync"})
sub = {}
for i in range(3):
    sub[8 + 8 * i] = [(44 + 2 * i, "BN partials combined"), (45 + 2 * i, "first item's loads landed, dot done")]
for wgi, label in enumerate(("first", "middle", "last")):
    t = [buf[wgi * 64 + i] for i in range(64)]
    print(f"--- {label} workgroup: total {t[41] - t[0]} cycles")
    prev = t[0]
    for i in sorted(names):
        if t[i] == 0:
            continue
        for si, sn in sub.get(i, []):
            if t[si]:
                print(f"      . {sn:38s} {t[si] - prev:7d} (since phase start)")
        print(f"  {names[i]:44s} {t[i] - prev:7d}")
        prev = t[i]
