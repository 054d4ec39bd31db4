"""N > 1 path on CPU: two `gloo` ranks exercise the data-parallel wrapper (flat parameter broadcast and
the single flat-gradient all-reduce) without any GPU.  The kernels are not involved — gradients are
injected — so this checks the collective plumbing bench.py / train loops rely on."""
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


def test_shard_batch_rejects_uneven_split():
    from graph_pooling_amd.parallel import shard_batch
    with pytest.raises(ValueError):
        shard_batch({"num_nodes": torch.arange(5)}, 0, 2)
