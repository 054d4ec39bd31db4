#!/usr/bin/env python3
"""Which ingredient of "a training step with its gradient all-reduce captured into a hipGraph" crashes capture_end on a
one-rank RCCL group (round 3: host segfault in torch/cuda/graphs.py capture_end)?  One variant per process:

    python tools/rccl_capture_probe.py <pg:0|1> <allreduce:0|1> <mode:global|thread_local> [op:avg|sum] [wrap:raw|dp] [pre:none|oracle]
"""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    faulthandler.enable()
    use_pg, use_ar, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    op = sys.argv[4] if len(sys.argv) > 4 else "avg"
    wrap = sys.argv[5] if len(sys.argv) > 5 else "raw"
    pre = sys.argv[6] if len(sys.argv) > 6 else "none"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if use_pg:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    from oracle import diffpool_oracle as O
    B, N, F_, H, Cc = 6, 160, 7, 12, 2
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=20, p=0.04, seed=33, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.1, linkpred=True).cuda()
    xd, ad, ld, nd = x.cuda(), adj.cuda(), label.cuda(), torch.from_numpy(nn_).cuda()
    rop = dist.ReduceOp.AVG if op == "avg" else dist.ReduceOp.SUM
    dp = None
    if wrap == "dp":
        from graph_pooling_amd.parallel import DataParallelEncoder
        dp = DataParallelEncoder(model, sync_bn=False, force=True)
    if pre == "oracle":       # what the test worker does before capturing: winners read back, a CPU autograd run
        from tests.parity import gpu_winners
        y = model(xd, ad, nd, assign_x=xd)
        wins = gpu_winners(model, 2)
        params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        yo, inter = O.softpool_forward(P, x, adj, nn_, x, winners=wins)
        lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, True)
        lo.backward()

    def step():
        model.zero_grad(set_to_none=True)
        y = model(xd, ad, nd, assign_x=xd)
        loss = model.loss(y, ld, ad, nd)
        loss.backward()
        if use_ar and dp is not None:
            dp.reduce_gradients()
        elif use_ar:
            dist.all_reduce(model._last_flat_grad, op=rop)
        return loss

    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    model.zero_grad(set_to_none=True)
    kw = {} if mode == "global" else {"capture_error_mode": mode}
    with torch.cuda.graph(g, **kw):
        l = step()
    g.replay()
    torch.cuda.synchronize()
    print(f"VARIANT pg={use_pg} allreduce={use_ar} mode={mode} op={op} wrap={wrap} pre={pre}: capture + replay ok, loss {float(l):.6f}", flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
