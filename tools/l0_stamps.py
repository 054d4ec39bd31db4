"""Phase timing inside the persistent level-0 forward kernel (diagnostic build: DP_STAMP=1 graph_pooling_amd/csrc/build.sh,
then DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. python tools/l0_stamps.py).  Shader-clock cycles
(100 MHz s_memtime ticks are NOT used: s_memtime counts shader cycles on gfx950) between stamps, for the first, the
middle and the last workgroup of the grid."""
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
lib.dp_debug_l0_stamps.restype = C.c_int
assert lib.dp_debug_l0_stamps(buf) == 0
names = {0: "entry", 1: "side zero", 2: "A rows -> LDS + pkA", 3: "x, W0 staged + sync", 4: "A^T strip", 5: "P0 mma + sync",
         6: "P0 split written", 7: "graph barrier 0"}
for l in range(3):
    o = 8 + 8 * l
    names.update({o: f"L{l} aggregate", o + 1: f"L{l} sync", o + 2: f"L{l} tail", o + 3: f"L{l} (no barrier: polled below)",
                  o + 4: f"L{l} BN entries polled + apply + sync", o + 5: f"L{l} transform + sync", o + 6: f"L{l} split written",
                  o + 7: f"L{l} graph barrier"})
names.update({40: "Wp staged + sync", 41: "logits mma + sync", 42: "softmax + sync", 43: "S split written",
              44: "graph barrier S", 45: "A^T rows in LDS + sync", 46: "A^T S aggregate", 47: "sync", 48: "T reduce + sync",
              49: "partial X', A' + sync", 50: "partials stored, max partial", 51: "graph barrier P", 52: "combine + sync"})
for wgi, label in enumerate(("first", "middle", "last")):
    t = [buf[wgi * 64 + i] for i in range(64)]
    print(f"--- {label} workgroup: total {t[52] - t[0]} cycles")
    prev = t[0]
    for i in sorted(names):
        if t[i] == 0 or i == 0:
            continue
        print(f"  {names[i]:28s} {t[i] - prev:7d}")
        prev = t[i]
