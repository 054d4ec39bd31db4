"""One rank of the two-rank data-parallel parity runs (launched by tests/test_gpu_parallel.py under
torch.distributed.run; both ranks share GPU 0 and talk over gloo — RCCL refuses two ranks on one device).  Each rank runs
the HIP forward / loss / backward on ITS shard through DataParallelEncoder and, after the flat-gradient all-reduce,
checks the result against the CPU oracle:

  sync-BN cases  ("small", "packed", "dd"):  ONE oracle step on the concatenated batch (SURVEY 8(e) mode ii);
  local-BN case  ("dd_local", the default throughput mode): each rank's ypred equals the oracle on its OWN shard and
                 the reduced gradients equal the mean of the per-shard oracle gradients (mode i).

"dd" / "dd_local" are BASELINE configs[3]'s shard shape: B = 20 per rank, N = 500, F = 89, H = 20, ratio 0.1, p = 0.02
(packed bf16 adjacency path, the BatchNorm combine kernels with Bs = B x world partial blocks)."""
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

CASES = {
    # Bl, N, F, H, C, ratio, p_edge, linkpred, sync_bn      (one-hot node features, as DD's node labels)
    "small": (3, 40, 5, 8, 3, 0.25, 0.15, True, True),
    "packed": (4, 160, 7, 12, 2, 0.1, 0.04, True, True),
    "dd": (20, 500, 89, 20, 2, 0.1, 0.02, True, True),
    "dd_local": (20, 500, 89, 20, 2, 0.1, 0.02, False, False),
}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = sys.argv[1] if len(sys.argv) > 1 else "small"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Bl, N, F_, H, Cc, ratio, p_edge, linkpred, sync_bn = CASES[case]
    B = Bl * world
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=max(2, N // 8), p=p_edge, seed=21, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, linkpred=linkpred)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=5, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    dp = DataParallelEncoder(model, sync_bn=sync_bn)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    xd, ad, nd, ld = x[sl].cuda(), adj[sl].cuda(), nn_[sl], label[sl].cuda()
    ypred = dp(xd, ad, nd, assign_x=xd)
    win_local = gpu_winners(model, 2)
    loss = model.loss(ypred, ld, ad, nd) if linkpred else model.loss(ypred, ld)
    loss.backward()
    dp.reduce_gradients()
    torch.cuda.synchronize()
    if sync_bn:
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
    if sync_bn:
        # ---- the oracle: ONE step on the concatenated batch
        close(ypred, O.softpool_forward(params, x, adj, nn_, x)[0][sl])
        P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        yo, inter = O.softpool_forward(P, x, adj, nn_, x, winners=wins)
        lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, linkpred)
        lo.backward()
        close(ypred, yo[sl])
        close(model.assign_tensor, inter["assign_0"][sl], 1e-4, 1e-6)
        close(torch.stack(losses).mean(), lo, 1e-4, 1e-6)       # the mean of the per-rank losses IS the batch loss
        grads_close(model, {k: v.grad for k, v in P.items()})
        # and it is NOT what local BatchNorm statistics give (the test would be vacuous otherwise)
        y_local, _ = O.softpool_forward(params, x[sl], adj[sl], nn_[sl], x[sl])
        assert float((y_local - yo[sl].detach()).abs().max()) > 1e-3
        what = "sync-BN step equals the oracle on the concatenated batch"
    else:
        # ---- local statistics: world independent reference steps with averaged gradients
        ref = None
        for r in range(world):
            s_r = slice(r * Bl, (r + 1) * Bl)
            P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
            yo, inter = O.softpool_forward(P, x[s_r], adj[s_r], nn_[s_r], x[s_r], winners=[w[s_r] for w in wins])
            lo, _ = O.softpool_loss(yo, label[s_r], inter["assign_0"], adj[s_r], nn_[s_r], linkpred)
            lo.backward()
            if r == rank:
                close(ypred, O.softpool_forward(params, x[s_r], adj[s_r], nn_[s_r], x[s_r])[0])   # the oracle's own arg-max
                close(ypred, yo)
                close(model.assign_tensor, inter["assign_0"], 1e-4, 1e-6)
                close(loss, lo, 1e-4, 1e-6)
            g = {k: v.grad / world for k, v in P.items()}
            ref = g if ref is None else {k: ref[k] + g[k] for k in g}
        grads_close(model, ref)
        # the shards' gradients differ, so a skipped all-reduce would not pass
        own = {k: v.grad for k, v in P.items()}
        assert max(float((own[k] - ref[k]).abs().max()) for k in ref) > 1e-6
        what = "local-BN step equals the mean of the per-shard oracle steps"
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {what} (B = {B})")


if __name__ == "__main__":
    main()
