"""Soak run of the captured DD step: many replays, then the device error word and the outputs are checked (the persistent
kernels' barriers / tagged polls are bounded and REPORT a give-up; this shows none happens over a long run).
PYTHONPATH=. python tools/soak.py [steps]"""
import sys
import time
import torch
import bench
from graph_pooling_amd import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lib = _lib.load()
w = bench.WORKLOADS["dd"]
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))


def fwd_bwd():
    model.zero_grad(set_to_none=True)
    y = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
    loss = model.loss(y, batch["label"])
    loss.backward()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
model.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = fwd_bwd()
g.replay()
torch.cuda.synchronize()
ref_loss = float(loss)
ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
t0 = time.perf_counter()
for i in range(steps):
    g.replay()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
err = lib.dp_device_error(0)
same = float(loss) == ref_loss and all(torch.equal(p.grad, ref_grads[k]) for k, p in model.named_parameters())
print(f"{steps} replays in {dt:.2f} s ({dt / steps * 1e3:.4f} ms/step); device error word {err}; "
      f"loss and every gradient bit-identical to the first replay: {same}")
assert err == 0 and same
