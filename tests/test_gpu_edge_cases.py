"""Edge shapes of the DiffPool path on the GPU against the CPU oracle (PARITY UNPINNED by the reference for the
configurations its train.py never builds — they are pinned against the oracle, which the golden fixtures pin for
the default configuration): one cluster, one graph, one node per graph, no / several pred_model hidden layers,
2- and 4-layer GCN stacks, a separate assign-feature width, weighted (non-bf16-exact) adjacency at a packed size.
Tolerances as in test_gpu_model.py."""
import numpy as np
import pytest
import torch

from graph_pooling_amd.encoders import GcnEncoderGraph, SoftPoolingGcnEncoder
from oracle import diffpool_oracle as O
from tests.parity import close, grads_close, gpu_winners

pytestmark = pytest.mark.gpu


def _run(B, N, F_, H, Cc, ratio, *, linkpred=False, num_layers=3, pred_hidden=(50,), assign_input_dim=-1,
         n_min=None, n_max=None, p=0.2, weighted=False, seed=3, atol_rel=2e-5):
    n_min = max(1, N // 8) if n_min is None else n_min
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=n_min, n_max=n_max, p=p, seed=seed, n_classes=Cc)
    if weighted:
        g = torch.Generator().manual_seed(seed)
        w = torch.rand(adj.shape, generator=g)
        adj = adj * (w + w.transpose(1, 2)) * 0.5
    assign_x = x
    if assign_input_dim > 0:
        g = torch.Generator().manual_seed(seed + 1)
        assign_x = torch.randn(B, N, assign_input_dim, generator=g) * O.node_mask(N, nn_)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, num_layers, H, assign_ratio=ratio, linkpred=linkpred,
                                  pred_hidden_dims=list(pred_hidden), assign_input_dim=assign_input_dim)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), adj.cuda(), nn_, assign_x=assign_x.cuda())
    win = gpu_winners(model, 2)
    loss = model.loss(ypred, label.cuda(), adj.cuda(), nn_) if linkpred else model.loss(ypred, label.cuda())
    loss.backward()
    close(ypred, O.softpool_forward(params, x, adj, nn_, assign_x, num_layers=num_layers,
                                    n_pred_hidden=len(pred_hidden))[0])
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_, assign_x, num_layers=num_layers, n_pred_hidden=len(pred_hidden),
                                   winners=win)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, linkpred)
    lo.backward()
    close(ypred, yo)
    close(model.assign_tensor, inter["assign_0"], 1e-4, 1e-6)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, {k: v.grad for k, v in P.items()}, atol_rel=atol_rel)


def test_single_cluster():
    _run(4, 16, 3, 8, 3, 0.1, linkpred=True)            # K = int(16 * 0.1) = 1: S is the node mask itself


def test_single_graph_batch():
    _run(1, 40, 5, 8, 2, 0.25, linkpred=True)


def test_every_graph_has_one_node():
    _run(5, 12, 3, 8, 3, 0.25, linkpred=True, n_min=1, n_max=1)


@pytest.mark.parametrize("hidden", [(), (16, 8)])
def test_pred_model_depths(hidden):
    _run(6, 30, 4, 8, 4, 0.2, pred_hidden=hidden)


@pytest.mark.parametrize("L", [2, 4])
def test_gcn_stack_depths(L):
    _run(4, 36, 5, 8, 3, 0.25, num_layers=L, linkpred=True)


def test_separate_assign_feature_width():
    _run(4, 32, 6, 8, 3, 0.25, assign_input_dim=9)


def test_weighted_adjacency_takes_the_fp32_fallback_at_a_packed_size():
    _run(3, 160, 7, 12, 2, 0.25, weighted=True, linkpred=True, p=0.05)


def test_wide_hidden_dims():
    _run(2, 130, 10, 70, 3, 0.3, p=0.05)                 # C = 140 per joint layer: wide kernel + fused tail fallbacks


@pytest.mark.parametrize("H,N,ratio", [
    (40, 200, 0.5),       # widths in (32, 64], K = 100 in (64, 128]: the NK = 4 row kernels, softmax <8, 2>
    (150, 96, 0.25),      # widths in (128, 256]: rownorm <16>, generic bn_apply
    (260, 64, 0.25),      # widths > 256: the any-width row kernels
    (12, 600, 0.5),       # K = 300 > 256: the any-K softmax kernels
])
def test_row_kernel_width_classes(H, N, ratio):
    """The row kernels (rownorm backward, bn apply, the softmax plan pair) are compiled per width class with the row
    held in registers; one configuration per class, against the oracle at fp32 tolerance."""
    _run(2, N, 9, H, 3, ratio, p=0.05)


@pytest.mark.parametrize("weighted", [False, True])
def test_widening_assign_layer_runs_in_the_reference_association(weighted):
    """K = 180 clusters from 8 hidden units: the last assign layer has 188 joint output columns against 16 input
    columns, so the plan runs it as (A x) W — encoders.py:966-968's own order, one narrow pass over A — instead of
    A (x W) (dp_model.hip layer_agg_first).  0/1 adjacency takes the packed bf16 pass, a weighted one its fp32
    fallback on the gathered inputs; both against the oracle, linkpred on so dS reaches the assign stack twice."""
    _run(3, 300, 7, 8, 2, 0.6, weighted=weighted, linkpred=True, p=0.05)


@pytest.mark.parametrize("B,N,F_,H,ratio", [
    (70, 264, 5, 20, 0.5),     # B > 64: no split-K, batch-combined BatchNorm statistics, K = 132 (quad row kernels)
    (3, 140, 4, 16, 0.95),     # K = 133: not a multiple of 4 -> the 4-byte row kernels and the GEMM products
    (2, 320, 6, 8, 1.0),       # K = 320 = N: the five-quad forms, 328 joint columns
])
def test_widening_layer_shape_sweep(B, N, F_, H, ratio):
    """More shapes through the (A x) W association and its row kernels, each against the oracle (the fused link
    loss takes K <= 256 and says so otherwise: off for the 320-cluster case)."""
    _run(B, N, F_, H, 2, ratio, linkpred=int(N * ratio) <= 256, p=0.04)


def test_widening_layer_with_more_than_twenty_hidden_units():
    """hidden 28: the row kernels of the widening layer (k_widen_fwd, k_rownorm_bwd_mv) take their 32-deep form (weight
    rows 28..31 zero in LDS); K = 140 clusters."""
    _run(2, 280, 6, 28, 2, 0.5, linkpred=True, p=0.05)


@pytest.mark.parametrize("concat", [True, False])
def test_widening_last_layer_of_the_base_encoder(concat):
    """GcnEncoderGraph with embedding_dim 160 from 16 hidden units: conv_last widens 10x and runs as (A x [+ x]) W
    (concat=False is add_self=True, encoders.py:1003); against the oracle's base_forward."""
    B, N, F_, H, E, Cc = 3, 200, 7, 16, 160, 3
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=N, p=0.05, seed=11, n_classes=Cc)
    model = GcnEncoderGraph(F_, H, E, Cc, 3, pred_hidden_dims=[20], concat=concat, bn=True)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=11, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), adj.cuda(), nn_)
    loss = model.loss(ypred, label.cuda())
    loss.backward()
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo = O.base_forward(P, x, adj, n_pred_hidden=1, bn=True, concat=concat)
    lo = torch.nn.functional.cross_entropy(yo, label)
    lo.backward()
    close(ypred, yo)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, {k: v.grad for k, v in P.items()})


def test_forward_is_bit_reproducible_with_split_k_pooling():
    """A level of 600 nodes at B = 2 pools with split-K contractions over the node index (X' = S^T Z, A' = Tt^T S).
    Their ranges are combined in a fixed order (partial tiles + tickets, dp_gemm.hip), not with float atomics, so the
    forward pass is bit-identical from run to run -- with atomics the last bit varied, which could flip a near-tied
    max-readout winner and reroute a gradient ('Forward determinism')."""
    B, N, F_, H, Cc = 2, 600, 9, 12, 3
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=N // 8, p=0.05, seed=3, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.5, pred_hidden_dims=[50])
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    first = None
    for i in range(40):
        if i % 3 == 1:                                   # perturb allocator state and timing between runs
            junk = torch.randn((1 + i % 5) * 1024 * 1024, device="cuda")
            del junk
        with torch.no_grad():
            y = model(xd, ad, nn_, assign_x=xd).clone()
        s = model.assign_tensor.clone()
        if first is None:
            first = (y, s)
        else:
            assert torch.equal(y, first[0]) and torch.equal(s, first[1]), f"run {i} differs from run 0"


def test_grid_barrier_give_up_is_reported_not_silent():
    """The whole-level kernels' grid barrier (dp_small.hip) rests on co-residency of one workgroup per graph.  When that
    assumption breaks the bounded spin gives up; this forces the give-up path with the test-only knob
    DP_TEST_BARRIER_FAIL (read once per process, hence the child process) and checks that it surfaces as an error
    through the C ABI and that the fused optimizer leaves the parameters alone — tests/_barrier_fail_worker.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DP_TEST_BARRIER_FAIL="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_barrier_fail_worker.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert "barrier give-up path complete" in r.stdout, r.stdout


def test_fused_optimizer_skips_a_non_finite_gradient_and_reports_it():
    from graph_pooling_amd import _lib
    from graph_pooling_amd.optim import FusedClipAdam
    lib = _lib.load()
    torch.manual_seed(0)
    model = SoftPoolingGcnEncoder(16, 3, 8, 8, 2, 3, 8, assign_ratio=0.25, linkpred=False).cuda()
    model._ensure_flat(torch.device("cuda", 0))
    before = model._flat.detach().clone()
    opt = FusedClipAdam(model, lr=1e-2, clip=2.0)
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    next(model.parameters()).grad[0, 0] = float("nan")
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(model._flat, before)
    assert lib.dp_device_error(0) == _lib.DEVERR_NONFINITE_GRAD
    with pytest.raises(RuntimeError, match="non-finite gradient norm"):
        opt.step()                                       # the NEXT entry reports it (and clears the word)
    assert lib.dp_device_error(0) == 0
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    opt.step()                                           # a finite step goes through again
    torch.cuda.synchronize()
    assert not torch.equal(model._flat, before) and torch.isfinite(model._flat).all()
