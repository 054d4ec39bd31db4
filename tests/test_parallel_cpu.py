"""N > 1 path on CPU: two `gloo` ranks exercise the data-parallel wrapper (flat parameter broadcast, the single
flat-gradient all-reduce with gradients from the oracle's backward on each shard, and the sync-BN partial exchange)
without any GPU.  The HIP kernels meet the collectives in tests/test_gpu_parallel.py (two ranks on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
        from graph_pooling_amd.parallel import DataParallelEncoder, shard_batch
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        model = SoftPoolingGcnEncoder(16, 3, 8, 8, 2, 3, 8, assign_ratio=0.25, linkpred=False)
        dp = DataParallelEncoder(model)                    # broadcasts rank 0's flat buffer
        flat0 = model._flat.clone()
        gathered = [torch.empty_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, flat0)
        same_params = all(torch.equal(gathered[0], g) for g in gathered)
        # every parameter is a view into the flat buffer
        views_ok = all(p.data_ptr() == model._flat.data_ptr() + 4 * off
                       for p, (off, _, _) in zip(model._flat_params, model._flat_index))

        # (a) gradients that alias one flat buffer (what _EncoderFn.backward hands to autograd)
        flat_g = torch.full((model._flat.numel(),), float(rank + 1))
        model._last_flat_grad = flat_g
        for p, (off, numel, shape) in zip(model._flat_params, model._flat_index):
            p.grad = flat_g[off:off + numel].view(shape)
        dp.reduce_gradients()
        expect = sum(range(1, world + 1)) / world
        aliased_ok = all(torch.allclose(p.grad, torch.full_like(p.grad, expect)) for p in model.parameters())
        aliased_one_buffer = flat_g.data_ptr() == model._flat_params[0].grad.data_ptr()

        # (b) independent gradient tensors (e.g. after accumulation): flatten -> all-reduce -> scatter back
        model._last_flat_grad = None
        for i, p in enumerate(model.parameters()):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
        dp.reduce_gradients()
        loose_ok = all(torch.allclose(p.grad, torch.full_like(p.grad, expect * (i + 1)))
                       for i, p in enumerate(model.parameters()))

        batch = {"num_nodes": torch.arange(8), "adj": torch.zeros(8, 4, 4)}
        sh = shard_batch(batch, rank, world)
        shard_ok = sh["num_nodes"].tolist() == list(range(rank * 4, rank * 4 + 4)) and sh["adj"].shape[0] == 4
        q.put((rank, same_params, views_ok, aliased_ok, aliased_one_buffer, loose_ok, shard_ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_flat_gradient_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for res in results:
        assert all(res[1:]), res


def _oracle_worker(rank, world, port, q):
    """Each rank's gradients come from the ORACLE's backward on its shard (the kernels need a GPU); after the
    wrapper's all-reduce they must equal the mean of the per-shard oracle gradients, which every rank also computes
    directly.  Then the sync-BN exchange: per-row BatchNorm partials of each shard, all-gathered through
    SyncBatchNormExchange and Chan-combined, must reproduce apply_bn over the concatenated batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
        from graph_pooling_amd.parallel import DataParallelEncoder, SyncBatchNormExchange
        from oracle import diffpool_oracle as O
        Bl, N, F_, H, Cc = 3, 16, 3, 8, 2
        x, adj, nn_, label = O.make_batch(Bl * world, N, F_, n_min=2, p=0.3, seed=9, n_classes=Cc)
        model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, linkpred=True)
        params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3, bias_scale=0.1)
        model.load_state_dict(params)
        dp = DataParallelEncoder(model)

        def shard_grads(r):
            sl = slice(r * Bl, (r + 1) * Bl)
            P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
            y, inter = O.softpool_forward(P, x[sl], adj[sl], nn_[sl], x[sl])
            loss, _ = O.softpool_loss(y, label[sl], inter["assign_0"], adj[sl], nn_[sl], True)
            loss.backward()
            return {k: v.grad for k, v in P.items()}
        mine = shard_grads(rank)
        for k, p_ in model.named_parameters():
            p_.grad = mine[k].clone()
        dp.reduce_gradients()
        every = [shard_grads(r) for r in range(world)]
        mean_ok = all(torch.allclose(p_.grad, sum(g[k] for g in every) / world, rtol=1e-6, atol=1e-8)
                      for k, p_ in model.named_parameters())

        # sync-BN exchange: local (row mean, row M2) partials -> gathered, batch-major -> Chan combine
        ex = SyncBatchNormExchange(None, world)
        g = torch.Generator().manual_seed(4)
        act = torch.randn(Bl * world, N, 7, generator=g)
        loc = act[rank * Bl:(rank + 1) * Bl]
        mean = loc.mean(dim=2)
        part = torch.stack([mean, ((loc - mean.unsqueeze(2)) ** 2).sum(dim=2)], dim=2).contiguous()     # [Bl, N, 2]
        allp = torch.empty(world * Bl, N, 2)
        ex.all_gather(allp, part)
        mu = allp[:, :, 0].mean(dim=0)
        var = (allp[:, :, 1] + 7 * (allp[:, :, 0] - mu) ** 2).sum(dim=0) / (world * Bl * 7)
        mine_bn = (loc - mu.view(1, N, 1)) / torch.sqrt(var.view(1, N, 1) + 1e-5)
        bn_ok = torch.allclose(mine_bn, O.bn_node(act)[rank * Bl:(rank + 1) * Bl], rtol=1e-5, atol=1e-6)
        q.put((rank, mean_ok, bn_ok, ex.calls == 1))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_oracle_gradients_and_sync_bn_exchange():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_oracle_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for res in results:
        assert all(res[1:]), res


def test_shard_batch_rejects_uneven_split():
    from graph_pooling_amd.parallel import shard_batch
    with pytest.raises(ValueError):
        shard_batch({"num_nodes": torch.arange(5)}, 0, 2)
