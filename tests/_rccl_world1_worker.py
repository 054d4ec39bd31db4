"""The RCCL code paths of graph_pooling_amd.parallel on ONE GPU (launched in a fresh child process by
tests/test_gpu_parallel.py): a world-size-1 `nccl` process group with DataParallelEncoder(force=True), which keeps every
collective on over the single rank.  Executes, on the real RCCL backend:

  1. the flat-parameter broadcast and the gradient all-reduce with ReduceOp.AVG;
  2. the sync-BN exchange through the library's callback (dist.all_gather_into_tensor) and the link-normaliser
     all-reduce — the step must equal the oracle on the batch (one rank: global statistics == local statistics);
  4. collective_capture_works (a captured + replayed all-reduce on a throw-away communicator), run last;
  3. a whole step — forward, loss, backward and the gradient all-reduce — captured into ONE hipGraph; replay must
     reproduce the eager step (every tensor except `*.bias`, whose float atomics differ in the last place between any
     two runs).  Mode `local` (argv[1], default): local BatchNorm statistics, the captured collective is the gradient
     all-reduce — bench.py's N > 1 configuration.  Mode `sync`: sync-BN, the eight all-gathers are captured too.

It is a one-rank run: it proves the calls execute and compose with graph capture on this runtime, not that two ranks
agree (at one rank RCCL short-cuts an in-place all-reduce to nothing and an all-gather to a device copy).  Any RCCL
failure is printed as `RCCL-ERROR: <text>` and the process exits non-zero; nothing is retried."""
import faulthandler
import os
import sys
import traceback

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    faulthandler.enable()          # a host crash inside the runtime still names the Python frame it came from
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    mode = sys.argv[1] if len(sys.argv) > 1 else "local"
    sync_bn = mode == "sync"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    from graph_pooling_amd.parallel import DataParallelEncoder, collective_capture_works
    from oracle import diffpool_oracle as O
    from tests.parity import close, grads_close, gpu_winners

    B, N, F_, H, Cc = 6, 160, 7, 12, 2
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=20, p=0.04, seed=33, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.1, linkpred=True)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=9, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    dp = DataParallelEncoder(model, sync_bn=sync_bn, force=True)
    assert dp.backend == "nccl" and dp.reduce_op == dist.ReduceOp.AVG
    xd, ad, ld = x.cuda(), adj.cuda(), label.cuda()
    nd = torch.from_numpy(nn_).cuda()             # resident, as bench.py passes it: no H2D copy inside the capture

    def step():
        model.zero_grad(set_to_none=True)
        ypred = dp(xd, ad, nd, assign_x=xd)
        loss = model.loss(ypred, ld, ad, nd)
        loss.backward()
        dp.reduce_gradients()
        return ypred, loss

    # ---- 1 + 2: eager step over RCCL, against the oracle
    ypred, loss = step()
    torch.cuda.synchronize()
    n_calls = 0
    if sync_bn:
        assert model._sync_bn.error is None, model._sync_bn.error
        n_calls = model._sync_bn.calls
        assert n_calls == 8, n_calls          # 2 (L - 1) (1 + 2 P) all-gathers at L = 3, P = 1
    wins = gpu_winners(model, 2)
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_, x, winners=wins)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, True)
    lo.backward()
    close(ypred, yo)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, {k: v.grad for k, v in P.items()})
    eager = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    eager_y, eager_loss = ypred.detach().clone(), loss.detach().clone()
    print("eager RCCL step (AVG all-reduce, all_gather_into_tensor x %d, link-norm all-reduce) equals the oracle"
          % n_calls, flush=True)
    # Drop every reference to the eager step's autograd graph before capturing: its AccumulateGrad nodes belong to the
    # DEFAULT stream, and a backward pass inside the capture that reuses them synchronises with that (non-capturing)
    # stream — which invalidates the capture and, on this runtime, segfaults capture_end instead of raising (round 3,
    # tools/rccl_capture_probe.py: the crash needs no collective at all, only a live graph from another stream).
    # The parameters' AccumulateGrad nodes live as long as ANY graph that uses them, and every forward re-uses the live
    # ones: the module's own references (assign_tensor, link_loss are graph outputs) must go too, so that the warm-up
    # below creates fresh nodes on the capture's side stream.
    del ypred, loss, yo, lo, inter, P
    model.assign_tensor = None
    model.link_loss = None
    model._last_save = None
    import gc
    gc.collect()

    # ---- 3: the whole step, collectives included, in one hipGraph
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    model.zero_grad(set_to_none=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        gy, gl = step()
    captured_calls = model._sync_bn.calls if sync_bn else 0
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    if sync_bn:
        assert model._sync_bn.calls == captured_calls   # replays re-run the captured collectives, not the callback
    torch.testing.assert_close(gy, eager_y, rtol=0, atol=0)
    torch.testing.assert_close(gl, eager_loss, rtol=0, atol=0)
    for k, p in model.named_parameters():
        if sync_bn and k.endswith(".bias"):
            # sync-BN runs level 0 on the per-layer kernels, whose bias sums are float atomics (last-place differences
            # between any two runs); the persistent kernels of the default mode have none
            torch.testing.assert_close(p.grad, eager[k], rtol=1e-4, atol=1e-7)
        else:
            torch.testing.assert_close(p.grad, eager[k], rtol=0, atol=0, msg=lambda m, k=k: f"{k}: {m}")
    print(f"captured step ({mode} BatchNorm) with the RCCL collectives inside replays the eager step", flush=True)
    # ---- 4: a captured collective on a throw-away communicator (bench.py's probe).  Run LAST: at one rank RCCL
    # short-cuts the in-place all-reduce to nothing, the probe's graph is EMPTY, and a step captured after an empty
    # graph crashed capture_end on this runtime (round 3, tools/rccl_capture_probe.py: every other order and capture
    # mode passes) — with two or more ranks the probe's graph holds the collective's kernel
    ok = collective_capture_works(device, 1, 0)
    print(f"collective_capture_works -> {ok}", flush=True)
    assert ok, "this runtime did not capture + replay an RCCL all-reduce"

    dist.destroy_process_group()
    print(f"RCCL world-1 run complete ({mode})", flush=True)


if __name__ == "__main__":
    try:
        main()
    except Exception as e:          # noqa: BLE001 — report and exit non-zero; never retried
        traceback.print_exc()
        print(f"RCCL-ERROR: {type(e).__name__}: {e}", flush=True)
        sys.exit(1)
