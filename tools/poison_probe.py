"""Uninitialised-read hunt: freed device blocks are filled with NaN (or a large finite value) before every model run, so
the caching allocator hands poisoned memory to the workspaces; any read of memory the kernels did not write shows up
as a NaN / a large error against the oracle.  Runs a sequence of configurations in one process, several rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import diffpool_oracle as O
from graph_pooling_amd.encoders import SoftPoolingGcnEncoder

CFGS = [(2, 96, 9, 150, 3, 0.25), (2, 64, 9, 260, 3, 0.25), (2, 600, 9, 12, 3, 0.5), (2, 200, 9, 40, 3, 0.5),
        (20, 500, 89, 20, 2, 0.1), (3, 160, 7, 12, 2, 0.25)]
poison = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")

def run(cfg, rnd):
    B, N, F_, H, Cc, ratio = cfg
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=max(1, N // 8), p=0.05, seed=3, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, pred_hidden_dims=[50], linkpred=False)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3, bias_scale=0.1)
    model.load_state_dict(params); model = model.cuda()
    xd, ad, ld = x.cuda(), adj.cuda(), label.cuda()
    junk = torch.full((96 * 1024 * 1024,), poison, device="cuda"); del junk        # 384 MB of poisoned free blocks
    y = model(xd, ad, nn_, assign_x=xd)
    loss = model.loss(y, ld); loss.backward()
    torch.cuda.synchronize()
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_, x, num_layers=3, n_pred_hidden=1)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, False)
    lo.backward()
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        g, r = p.grad.detach().cpu(), P[k].grad
        sc = float(r.abs().max()) + 1e-30
        d = float((g - r).abs().max()) / sc if torch.isfinite(g).all() else float("inf")
        if d > worst[1]: worst = (k, d)
    dy = float((y.detach().cpu() - yo.detach()).abs().max())
    print(f"round {rnd} cfg {cfg}: |dy| {dy:.2e}  worst grad {worst[0]} rel-to-scale {worst[1]:.2e}", flush=True)

for rnd in range(3):
    for cfg in CFGS:
        run(cfg, rnd)
