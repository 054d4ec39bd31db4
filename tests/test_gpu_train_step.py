"""N1 + N2 end to end: the packed-adjacency builder and encoder entries (dp_build_batch_packed,
dp_encoder_forward_packed / _backward_packed), the device-counted Adam step (dp_clip_adam_step_counted) and the
training step captured as one hipGraph (train_step.CapturedTrainStep) — each against the fp32 / eager path it
replaces, bit for bit where the arithmetic is the same."""
import numpy as np
import pytest
import torch

from graph_pooling_amd.batch_builder import DeviceBatchBuilder, EdgeListDataset
from graph_pooling_amd.tu_dataset import TUGraph

pytestmark = pytest.mark.gpu


def _graphs(count, n_min, n_max, n_labels, p, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.integers(n_min, n_max + 1))
        a = np.triu((rng.random((n, n)) < p).astype(np.float32), 1)
        out.append(TUGraph(a + a.T, rng.integers(0, n_labels, n), int(rng.integers(0, 2))))
    return out


def _model(N, F_, H, ratio, seed=0):
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    torch.manual_seed(seed)
    return SoftPoolingGcnEncoder(N, F_, H, H, 2, 3, H, assign_ratio=ratio, linkpred=False).cuda()


@pytest.mark.parametrize("B,N,F_,H,ratio", [(6, 160, 8, 12, 0.1), (20, 500, 89, 20, 0.1)])
def test_packed_builder_drives_the_encoder_bit_identically_to_the_fp32_builder(B, N, F_, H, ratio):
    """Same edge lists through dp_build_batch (dense fp32) and dp_build_batch_packed (bf16 rows): the packed rows equal
    dp_adj_pack of the dense batch, and forward, loss and every gradient are bit-identical (the persistent level-0
    kernels multiply from the same bf16 values either way)."""
    from graph_pooling_amd.encoders import PackedAdjacency
    graphs = _graphs(B + 3, N // 4, N, F_, 0.03, seed=B + N)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    idx = list(range(1, B + 1))
    builder = DeviceBatchBuilder(ds, N, F_, "cuda")
    dense = builder.build(idx)
    packed = builder.build(idx, packed=True)
    assert isinstance(packed["adj"], PackedAdjacency) and packed["adj"].pk is packed["adj"].pkt
    ref = PackedAdjacency.from_dense(dense["adj"])
    assert torch.equal(packed["adj"].pk, ref.pk) and torch.equal(packed["adj"].pk, ref.pkt)
    assert torch.equal(packed["feats"], dense["feats"])
    assert torch.equal(packed["num_nodes_device"], dense["num_nodes_device"])
    model = _model(N, F_, H, ratio)
    outs = []
    for batch in (dense, packed):
        model.zero_grad(set_to_none=True)
        y = model(batch["feats"], batch["adj"], batch["num_nodes_device"], assign_x=batch["assign_feats"])
        loss = model.loss(y, batch["label"])
        loss.backward()
        outs.append((y.detach().clone(), loss.detach().clone(), model.assign_tensor.detach().clone(),
                     {k: p.grad.clone() for k, p in model.named_parameters()}))
    (y0, l0, s0, g0), (y1, l1, s1, g1) = outs
    assert torch.isfinite(y0).all()
    assert torch.equal(y0, y1) and torch.equal(l0, l1) and torch.equal(s0, s1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # an asymmetric pair A / A^T through from_dense takes the same entry
    model.zero_grad(set_to_none=True)
    y2 = model(dense["feats"], ref, dense["num_nodes_device"], assign_x=dense["feats"])
    assert torch.equal(y2, y0)


def test_packed_entry_refuses_a_configuration_outside_the_persistent_plan():
    """N = 48 < 64 runs the small-level kernels, which read the fp32 adjacency: DP_ERR_UNSUPPORTED, not a wrong result."""
    graphs = _graphs(6, 10, 48, 3, 0.1, seed=3)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    batch = DeviceBatchBuilder(ds, 48, 3, "cuda").build(range(6), packed=True)
    model = _model(48, 3, 8, 0.25)
    with pytest.raises(RuntimeError, match="packed-adjacency entry needs the persistent"):
        model(batch["feats"], batch["adj"], batch["num_nodes_device"], assign_x=batch["feats"])


def test_device_counted_adam_equals_the_host_counted_step():
    from graph_pooling_amd import _lib
    lib = _lib.load()
    n = 18672
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.dp_clip_adam_workspace_bytes(), device="cuda", dtype=torch.uint8)
    pa, ma, va = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb, mb, vb = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    counter = torch.zeros(1, device="cuda", dtype=torch.int32)
    tn = torch.zeros(2, device="cuda")
    for step in range(1, 8):
        grad = (torch.randn(n, generator=g) * (5.0 if step == 3 else 0.2)).cuda()
        ga, gb = grad.clone(), grad.clone()
        _lib.check(lib.dp_clip_adam_step(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), n, step, 1e-3, 0.9,
                                         0.999, 1e-8, 2.0, tn[0:1].data_ptr(), ws.data_ptr(), ws.numel(), st))
        _lib.check(lib.dp_clip_adam_step_counted(pb.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), n,
                                                 counter.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 2.0, tn[1:2].data_ptr(),
                                                 ws.data_ptr(), ws.numel(), st))
        assert int(counter.item()) == step
        assert torch.equal(ga, gb) and float(tn[0]) == float(tn[1])
        # the bias corrections come from pow() in double on the device instead of the host: equal to the last float bit
        # or one off
        np.testing.assert_allclose(pb.cpu().numpy(), pa.cpu().numpy(), rtol=3e-7, atol=1e-9)


def test_captured_training_step_follows_the_eager_training_loop():
    """Five training steps (different batches) two ways from the same initial model: eager — packed builder, forward,
    loss, backward, FusedClipAdam with the host-side step count — and CapturedTrainStep (one graph launch per step, edge
    lists read from pinned memory by the builder kernel, Adam's count on the device).  Same losses, same parameters."""
    from graph_pooling_amd.optim import FusedClipAdam
    from graph_pooling_amd.train_step import CapturedTrainStep
    B, N, F_, H = 6, 160, 8, 12
    graphs = _graphs(5 * B, N // 4, N, F_, 0.03, seed=77)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    batches = [list(range(i * B, (i + 1) * B)) for i in range(5)]
    batches[3] = list(reversed(batches[3]))
    builder = DeviceBatchBuilder(ds, N, F_, "cuda")

    eager = _model(N, F_, H, 0.1, seed=4)
    opt_e = FusedClipAdam(eager, lr=1e-2, clip=2.0)
    losses_e = []
    for idx in batches:
        b = builder.build(idx, packed=True)
        eager.zero_grad(set_to_none=True)
        y = eager(b["feats"], b["adj"], b["num_nodes_device"], assign_x=b["feats"])
        loss = eager.loss(y, b["label"])
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
    del y, loss

    cap = _model(N, F_, H, 0.1, seed=4)
    opt_c = FusedClipAdam(cap, lr=1e-2, clip=2.0, device_step_counter=True)
    p_init = {k: v.detach().clone() for k, v in cap.named_parameters()}
    step = CapturedTrainStep(cap, opt_c, builder, B)
    for k, v in cap.named_parameters():              # capturing (and its warm-up) left the model where it was
        assert torch.equal(v.detach(), p_init[k]), k
    assert opt_c.step_count == 0 and int(opt_c.step_dev.item()) == 0
    losses_c = [float(step(idx)) for idx in batches]
    assert step.skipped_entries() == 0
    assert int(opt_c.step_dev.item()) == 5 and opt_c.step_count == 5
    np.testing.assert_allclose(losses_c, losses_e, rtol=1e-5)
    assert len(set(losses_c)) == 5                   # five different batches were really trained on
    for (k, pe), (_, pc) in zip(eager.named_parameters(), cap.named_parameters()):
        np.testing.assert_allclose(pc.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


def test_packed_adjacency_from_dense_refuses_values_bf16_cannot_hold():
    """PackedAdjacency.from_dense is dp_adj_pack + its exactness flag: a weighted adjacency must use the fp32 entry."""
    from graph_pooling_amd.encoders import PackedAdjacency
    adj = torch.zeros(2, 128, 128, device="cuda")
    adj[0, 3, 5] = adj[0, 5, 3] = 1.0
    pa = PackedAdjacency.from_dense(adj)
    assert pa.pk.shape == (2, 128, 128) and int(pa.pk[0, 3, 5]) == 0x3F80 and int(pa.pkt[0, 5, 3]) == 0x3F80
    adj[1, 7, 9] = 0.1                                   # not representable in 8 mantissa bits
    with pytest.raises(ValueError, match="bf16 cannot hold"):
        PackedAdjacency.from_dense(adj)
