"""One rank of the two-rank sync-BN parity run (launched by tests/test_gpu_parallel.py under torch.distributed.run;
both ranks share GPU 0 and talk over gloo — RCCL refuses two ranks on one device).  Each rank runs the HIP forward /
loss / backward on ITS shard with DataParallelEncoder(sync_bn=True); after the flat-gradient all-reduce every rank
checks the result against the CPU oracle evaluated ONCE on the concatenated batch."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_pooling_amd.encoders import SoftPoolingGcnEncoder          # noqa: E402
from graph_pooling_amd.parallel import DataParallelEncoder            # noqa: E402
from oracle import diffpool_oracle as O                               # noqa: E402
from tests.parity import close, grads_close, gpu_winners              # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = sys.argv[1] if len(sys.argv) > 1 else "small"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Bl, N, F_, H, Cc, ratio, p_edge = (3, 40, 5, 8, 3, 0.25, 0.15) if case == "small" else (4, 160, 7, 12, 2, 0.1, 0.04)
    B = Bl * world
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=max(2, N // 8), p=p_edge, seed=21, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, linkpred=True)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=5, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    dp = DataParallelEncoder(model, sync_bn=True)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    xd, ad, nd, ld = x[sl].cuda(), adj[sl].cuda(), nn_[sl], label[sl].cuda()
    ypred = dp(xd, ad, nd, assign_x=xd)
    win_local = gpu_winners(model, 2)
    loss = model.loss(ypred, ld, ad, nd)
    loss.backward()
    dp.reduce_gradients()
    torch.cuda.synchronize()
    assert model._sync_bn.error is None, model._sync_bn.error
    assert model._sync_bn.calls > 0
    # the winners of every rank, in batch order, for the oracle's forced arg-max (tests/parity.py)
    wins = []
    for w in win_local:
        parts = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(parts, w)
        wins.append(torch.cat(parts, 0))
    losses = [torch.empty(1) for _ in range(world)]
    dist.all_gather(losses, loss.detach().cpu().reshape(1))
    # ---- the oracle: ONE step on the concatenated batch
    close(ypred, O.softpool_forward(params, x, adj, nn_, x)[0][sl])
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_, x, winners=wins)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, True)
    lo.backward()
    close(ypred, yo[sl])
    close(model.assign_tensor, inter["assign_0"][sl], 1e-4, 1e-6)
    close(torch.stack(losses).mean(), lo, 1e-4, 1e-6)       # the mean of the per-rank losses IS the batch loss
    grads_close(model, {k: v.grad for k, v in P.items()})
    # and it is NOT what local BatchNorm statistics give (the test would be vacuous otherwise)
    y_local, _ = O.softpool_forward(params, x[sl], adj[sl], nn_[sl], x[sl])
    assert float((y_local - yo[sl].detach()).abs().max()) > 1e-3
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: sync-BN step equals the oracle on the concatenated batch (B = {B})")


if __name__ == "__main__":
    main()
