"""The persistent level-0 kernels (csrc/dp_level0.hip: k_level0_fwd / k_level0_bwd, one launch per direction for the
whole level-0 forward / backward of SoftPoolingGcnEncoder, encoders.py:1254-1279) against the CPU oracle over their
geometries — 16 / 32 / 48 / 64 rows per workgroup, ragged last row blocks, graphs of one node, no mask — plus the paths
only some inputs take: a WEIGHTED adjacency (not bf16-exact: the in-kernel fp32 fallback), evaluation mode, the link loss
arriving as d_assign.  Every case asserts through the library's launch counters that the persistent kernels really ran
(the plan falls back to the per-phase kernels silently when a shape is outside their envelope)."""
import ctypes as C

import numpy as np
import pytest
import torch

from graph_pooling_amd import _lib
from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
from oracle import diffpool_oracle as O
from tests.parity import close, grads_close, gpu_winners

pytestmark = pytest.mark.gpu


class _Counted:
    """Counts the eager launches of the two persistent kernels inside the block."""

    def __enter__(self):
        self.lib = _lib.load()
        _lib.check(self.lib.dp_profile_level0(1))
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.n = []
        for which in (0, 1):
            us, n = C.c_double(0.0), C.c_int(0)
            _lib.check(self.lib.dp_profile_level0_read(which, C.byref(us), C.byref(n)))
            self.n.append(n.value)
        _lib.check(self.lib.dp_profile_level0(0))
        return False


def _run_case(B, N, F_, H, Cc, ratio, p, linkpred, *, seed=1, n_min=None, sizes=None, weighted=False, masked=True,
              onehot=True):
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=n_min or max(1, N // 10), p=p, seed=seed, n_classes=Cc,
                                      sizes=sizes, onehot=onehot)
    if weighted:        # edge weights that bf16 cannot hold: the kernels must notice and multiply in fp32
        g = torch.Generator().manual_seed(seed + 100)
        w = torch.rand(B, N, N, generator=g) + 0.5
        adj = adj * (w + w.transpose(1, 2))
    nn_arg = nn_ if masked else None
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, linkpred=linkpred)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed - 1, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    with _Counted() as cnt:
        ypred = model(xd, ad, nn_arg, assign_x=xd)
        win = gpu_winners(model, 2)
        loss = model.loss(ypred, label.cuda(), ad, nn_arg) if linkpred else model.loss(ypred, label.cuda())
        loss.backward()
    assert cnt.n == [1, 1], f"persistent level-0 kernels launched {cnt.n} times (forward, backward): the plan fell back"
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_arg, x, winners=win)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_arg, linkpred)
    lo.backward()
    close(ypred, yo)
    close(model.assign_tensor, inter["assign_0"], 1e-4, 1e-6)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, {k: v.grad for k, v in P.items()})
    return model, (xd, ad, nn_arg)


@pytest.mark.parametrize("B,N,F_,H,Cc,ratio,p,linkpred,tag", [
    (20, 100, 3, 20, 6, 0.1, 0.10, True, "the ENZYMES shape (N < 128: no packed aggregation kernels, persistent pair only)"),
    (4, 64, 5, 8, 3, 0.25, 0.15, False, "smallest level the pair takes"),
    (6, 160, 8, 12, 3, 0.1, 0.05, False, "16-row blocks"),
    (5, 132, 8, 12, 3, 0.1, 0.05, True, "16-row blocks, last block of 4 rows, link loss"),
    (20, 256, 16, 20, 2, 0.2, 0.03, False, "32-row blocks"),
    (20, 500, 89, 20, 2, 0.1, 0.02, False, "48-row blocks (the DD shape)"),
    (40, 384, 8, 16, 2, 0.05, 0.02, True, "64-row blocks, link loss"),
    (3, 640, 4, 8, 2, 0.1, 0.01, False, "two 512-column segments"),
])
def test_persistent_level0_geometries_against_the_oracle(B, N, F_, H, Cc, ratio, p, linkpred, tag):
    _run_case(B, N, F_, H, Cc, ratio, p, linkpred)


def test_persistent_level0_with_tiny_and_full_graphs():
    """Graphs of 1 and 2 nodes (every row block but the first is padding) next to graphs that fill N."""
    _run_case(6, 160, 8, 12, 3, 0.1, 0.3, True, sizes=[1, 2, 160, 17, 160, 3])


def test_persistent_level0_without_a_mask():
    _run_case(4, 144, 6, 12, 2, 0.1, 0.05, True, masked=False, n_min=144, onehot=False)


@pytest.mark.parametrize("B,N", [(6, 160), (20, 500)])
def test_persistent_level0_weighted_adjacency_takes_the_fp32_products(B, N):
    """An adjacency that is not exactly representable in bf16 (edge weights): the kernels flag the graph on the device
    and aggregate with the fp32 MFMA from the fp32 rows — same results as the oracle's fp32 matmul."""
    _run_case(B, N, 8, 12, 2, 0.1, 0.04, True, weighted=True)


def test_persistent_level0_mixed_exact_and_weighted_graphs_in_one_batch():
    """The exactness decision is per graph: a batch with both kinds."""
    B, N, F_, H = 6, 160, 8, 12
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=20, p=0.05, seed=3, n_classes=2)
    adj[1] *= 0.37
    adj[4] *= 1.0 / 3.0
    model = SoftPoolingGcnEncoder(N, F_, H, H, 2, 3, H, assign_ratio=0.1, linkpred=False)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=2, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    with _Counted() as cnt:
        ypred = model(x.cuda(), adj.cuda(), nn_, assign_x=x.cuda())
        win = gpu_winners(model, 2)
        model.loss(ypred, label.cuda()).backward()
    assert cnt.n == [1, 1]
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_, x, winners=win)
    O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, False)[0].backward()
    close(ypred, yo)
    grads_close(model, {k: v.grad for k, v in P.items()})


def test_persistent_level0_evaluation_forward_equals_training_forward():
    model, (xd, ad, nn_) = _run_case(6, 160, 8, 12, 3, 0.1, 0.05, False)
    y_train = model(xd, ad, nn_, assign_x=xd).detach().clone()
    with _Counted() as cnt, torch.no_grad():
        y_eval = model(xd, ad, nn_, assign_x=xd)
        labels = model.predict(xd, ad, nn_, assign_x=xd)
    assert cnt.n == [2, 0]
    assert torch.equal(y_eval, y_train)
    assert torch.equal(labels.cpu(), y_train.argmax(1).cpu())


def test_persistent_level0_is_bit_reproducible_run_to_run():
    """Same inputs, five runs: forward outputs and every gradient bit-identical (block-ordered combines, no float
    atomics anywhere on the persistent path)."""
    B, N, F_, H = 20, 500, 89, 20
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=50, p=0.02, seed=4, n_classes=2)
    torch.manual_seed(0)
    model = SoftPoolingGcnEncoder(N, F_, H, H, 2, 3, H, assign_ratio=0.1, linkpred=True).cuda()
    xd, ad, ld = x.cuda(), adj.cuda(), label.cuda()
    runs = []
    for _ in range(5):
        model.zero_grad(set_to_none=True)
        y = model(xd, ad, nn_, assign_x=xd)
        model.loss(y, ld, ad, nn_).backward()
        runs.append((y.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    for y, g in runs[1:]:
        assert torch.equal(y, runs[0][0])
        for k in g:
            assert torch.equal(g[k], runs[0][1][k]), k
