import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from oracle import diffpool_oracle as O
from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
B, N, F_, H, Cc, ratio = 2, 600, 9, 12, 3, 0.5
x, adj, nn_, label = O.make_batch(B, N, F_, n_min=N // 8, p=0.05, seed=3, n_classes=Cc)
model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, pred_hidden_dims=[50], linkpred=False)
params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3, bias_scale=0.1)
model.load_state_dict(params); model = model.cuda()
runs = []
for i in range(6):
    model.zero_grad(set_to_none=True)
    y = model(x.cuda(), adj.cuda(), nn_, assign_x=x.cuda())
    loss = model.loss(y, label.cuda()); loss.backward()
    runs.append({k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()})
def ref(dtype):
    P = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x.to(dtype), adj.to(dtype), nn_, x.to(dtype), num_layers=3, n_pred_hidden=1)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj.to(dtype), nn_, False)
    lo.backward()
    return {k: v.grad for k, v in P.items()}
r32 = ref(torch.float32)
try:
    r64 = ref(torch.float64)
except Exception as e:
    print("fp64 oracle failed:", e); r64 = None
for k in runs[0]:
    sc = float(runs[0][k].abs().max())
    nd = max(float((runs[i][k] - runs[0][k]).abs().max()) for i in range(1, 6))
    d32 = float((runs[0][k] - r32[k]).abs().max())
    line = f"{k:45s} scale {sc:9.3e} run-to-run {nd:9.3e} vs oracle32 {d32:9.3e}"
    if r64 is not None:
        line += f" gpu-vs-64 {float((runs[0][k].double() - r64[k]).abs().max()):9.3e} oracle32-vs-64 {float((r32[k].double() - r64[k]).abs().max()):9.3e}"
    print(line)
