import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
w = bench.WORKLOADS["dd"]
model, batch, _ = bench.make_model_and_batch(w, False, torch.device("cuda"))
model.train()
for i in range(2):
    if i == 1: sys.stderr.write("=== step\n")
    y = model(batch["x"], batch["adj"], batch["nn"], assign_x=batch["x"])
    loss = model.loss(y, batch["label"])
    sys.stderr.write("--- backward\n") if i == 1 else None
    loss.backward()
torch.cuda.synchronize()
