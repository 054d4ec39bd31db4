"""N4: the CSR path for graphs beyond the padded dense one (graph_pooling_amd.sparse) against the oracle's DENSE
restatement of GcnEncoderGraph on the same single graph (B = 1) — PARITY UNPINNED by the reference, which drops such
graphs (load_data.py:79) — plus the CSR aggregation op against a dense matrix product."""
import numpy as np
import pytest
import torch

from graph_pooling_amd import _lib
from graph_pooling_amd.sparse import CsrGraph, SparseGcnEncoderGraph, hip_linear
from oracle import diffpool_oracle as O
from tests.parity import close, grads_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mean", [0, 1])
@pytest.mark.parametrize("n,feat,deg", [(50, 7, 3), (700, 89, 6), (300, 130, 70)])
def test_csr_aggregate_matches_dense_product(n, feat, deg, mean):
    g = torch.Generator().manual_seed(n + feat)
    adj = (torch.rand(n, n, generator=g) < deg / n).float()
    adj[:, 0] = 1.0                                    # no empty rows (mean of nothing is NaN in the reference too)
    table = torch.randn(n, feat, generator=g)
    csr = CsrGraph.from_dense(adj.cuda())
    out = torch.full((n, feat), 2.0, device="cuda")
    lib = _lib.load()
    _lib.check(lib.dp_csr_aggregate(table.cuda().data_ptr(), feat, csr.indptr.data_ptr(), csr.indices.data_ptr(),
                                    out.data_ptr(), feat, n, feat, mean, 0.5, torch.cuda.current_stream().cuda_stream))
    ref = adj.double() @ table.double()
    if mean:
        ref = ref / adj.sum(1, keepdim=True).double()
    close(out, (ref + 1.0).float(), 1e-5, 1e-5)


def test_hip_linear_matches_torch():
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(37, 120, generator=g), torch.randn(6, 120, generator=g), torch.randn(6, generator=g)
    xs, ws, bs = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    y = hip_linear(xs, ws, bs)
    gy = torch.randn(37, 6, generator=g)
    (y * gy.cuda()).sum().backward()
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    yo = torch.nn.functional.linear(xo, wo, bo)
    (yo * gy).sum().backward()
    close(y, yo)
    close(xs.grad, xo.grad, 1e-4, 1e-5)
    close(ws.grad, wo.grad, 1e-4, 1e-4)
    close(bs.grad, bo.grad, 1e-4, 1e-5)


@pytest.mark.parametrize("n,concat,bn,hidden", [(300, True, True, [50]), (64, False, True, []), (5748, True, True, [50]),
                                                (97, True, False, [])])
def test_sparse_encoder_equals_dense_oracle_PARITY_UNPINNED(n, concat, bn, hidden):
    """n = 5748 is DD's largest graph: 0.3 MB as CSR here, 132 MB as the dense fp32 block the reference
    would need.  Symmetric 0/1 adjacency, mean degree ~5, one isolated node."""
    F_, H, E, Cc = 9, 20, 20, 2
    g = torch.Generator().manual_seed(n)
    m = 5 * n // 2
    src = torch.randint(1, n, (m,), generator=g).numpy()
    dst = torch.randint(1, n, (m,), generator=g).numpy()
    keep = src != dst
    csr = CsrGraph.from_edges(n, src[keep], dst[keep], "cuda", symmetric=True)      # node 0 stays isolated
    adj = torch.zeros(n, n)
    adj[src[keep], dst[keep]] = 1.0
    adj = torch.maximum(adj, adj.t())
    x = torch.randn(n, F_, generator=g)
    label = torch.tensor([1])
    model = SparseGcnEncoderGraph(F_, H, E, Cc, 3, pred_hidden_dims=hidden, concat=concat, bn=bn)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=n, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), csr)
    loss = torch.nn.functional.cross_entropy(ypred, label.cuda())
    loss.backward()
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo = O.base_forward(P, x.unsqueeze(0), adj.unsqueeze(0), n_pred_hidden=len(hidden), bn=bn, concat=concat)
    lo = torch.nn.functional.cross_entropy(yo, label)
    lo.backward()
    close(ypred, yo)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, {k: v.grad for k, v in P.items()}, rtol=2e-3, atol_rel=1e-4)
    assert int(model.predict(x.cuda(), csr)) == int(yo.argmax(dim=1))
