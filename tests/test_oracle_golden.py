"""Pin the CPU oracle (oracle/diffpool_oracle.py) against golden vectors produced by the
reference's own classes (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import diffpool_oracle as O
from tests.conftest import load_golden

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert np.isfinite(a).all() and np.isfinite(b).all(), "non-finite values in a parity check"
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def req(params):
    return {k: v.clone().requires_grad_(True) for k, v in params.items()}


G1_CASES = [f"g1_graphconv_s{s}b{b}n{n}" for s in (0, 1) for b in (0, 1) for n in (1, 0)]


@pytest.mark.parametrize("name", G1_CASES)
def test_g1_graphconv_reference_text(name):
    """A1 pinned by the reference's own (commented-out) GraphConv text, encoders.py:944-974 — see make_golden.py P2."""
    a, p, g = load_golden(name)
    add_self, bias, normalize = [bool(v) for v in a["cfg"]]
    assert bias == ("bias" in p)
    x = T(a["x"]).requires_grad_(True)
    adj = T(a["adj"]).requires_grad_(True)
    P = req(p)
    y = O.graph_conv(x, adj, P["weight"], P.get("bias"), add_self, normalize)
    close(y, a["y"], 0, 0)               # same torch ops in the same order: bit-identical
    (y * T(a["gy"])).sum().backward()
    close(x.grad, a["gx"], 1e-6, 1e-7)
    close(adj.grad, a["gadj"], 1e-6, 1e-7)
    for k in g:
        close(P[k].grad, g[k], 1e-6, 1e-7)


def test_g2_apply_bn():
    a, _, _ = load_golden("g2_apply_bn")
    x = T(a["x"]).requires_grad_(True)
    y = O.bn_node(x)
    close(y, a["y"])
    (y * T(a["gy"])).sum().backward()
    close(x.grad, a["gx"], rtol=1e-4, atol=1e-5)


def test_g3_gcn_forward_masked():
    a, p, g = load_golden("g3_gcn_forward")
    P = req(p)
    x = T(a["x"]).requires_grad_(True)
    mask = O.node_mask(x.shape[1], a["num_nodes"])
    close(mask, a["mask"], 0, 0)
    z = O.gcn_stack(x, T(a["adj"]), P, ["conv_first", "conv_block.0", "conv_last"], mask)
    close(z, a["z"])
    (z * T(a["gz"])).sum().backward()
    close(x.grad, a["gx"], rtol=1e-4, atol=1e-6)
    for k in g:
        if k.startswith("pred_model"):
            continue
        close(P[k].grad, g[k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["g4_softpool_n16_f3", "g5_softpool_n16_f3_link",
                                  "g4_softpool_n100_f89", "g5_softpool_n100_f89_link",
                                  "g9_enzymes_batch"])
def test_g4_g5_g9_softpool(name):
    a, p, g = load_golden(name)
    P = req(p)
    x = T(a["x"])
    if "adj" in a:
        adj = T(a["adj"])
    else:
        N = x.shape[1]
        adj = T(np.unpackbits(a["adj_bits"], axis=-1)[..., :N].astype(np.float32))
    linkpred = "link_loss" in a
    ypred, inter = O.softpool_forward(P, x, adj, a["num_nodes"], x, want_intermediates=True)
    close(ypred, a["ypred"], rtol=1e-4, atol=1e-5)
    close(inter["assign_0"], a["assign"], rtol=1e-4, atol=1e-6)
    if "xpool" in a:
        close(inter["xpool_0"], a["xpool"], rtol=1e-4, atol=1e-5)
        close(inter["adjpool_0"], a["adjpool"], rtol=1e-4, atol=1e-5)
    loss, link = O.softpool_loss(ypred, T(a["label"]), inter["assign_0"], adj, a["num_nodes"], linkpred)
    close(loss, a["loss"], rtol=1e-5, atol=1e-6)
    if linkpred:
        close(link, a["link_loss"], rtol=1e-5, atol=1e-6)
    loss.backward()
    for k in g:
        close(P[k].grad, g[k], rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("tag", ["concat", "addself", "nobn"])
def test_g6_base_encoder(tag):
    a, p, g = load_golden(f"g6_base_{tag}")
    concat, bn, nh = [int(v) for v in a["cfg"]]
    P = req(p)
    ypred = O.base_forward(P, T(a["x"]), T(a["adj"]), n_pred_hidden=nh, bn=bool(bn), concat=bool(concat))
    close(ypred, a["ypred"], rtol=1e-4, atol=1e-5)
    loss, _ = O.softpool_loss(ypred, T(a["label"]))
    close(loss, a["loss"])
    loss.backward()
    for k in g:
        close(P[k].grad, g[k], rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("n", [7, 100])
def test_g7_set2set(n):
    a, p, g = load_golden(f"g7_set2set_n{n}")
    P = req(p)
    emb = T(a["emb"]).requires_grad_(True)
    out = O.set2set_forward(emb, P)
    close(out, a["out"], rtol=1e-4, atol=1e-5)
    (out * T(a["gout"])).sum().backward()
    close(emb.grad, a["gemb"], rtol=2e-3, atol=1e-5)
    for k in g:
        close(P[k].grad, g[k], rtol=2e-3, atol=1e-5)


def test_g7_set2set_encoder():
    a, p, g = load_golden("g7_set2set_encoder")
    P = req(p)
    ypred = O.set2set_encoder_forward(P, T(a["x"]), T(a["adj"]), a["num_nodes"])
    close(ypred, a["ypred"], rtol=1e-4, atol=1e-5)
    loss, _ = O.softpool_loss(ypred, T(a["label"]))
    close(loss, a["loss"])
    loss.backward()
    for k in g:
        close(P[k].grad, g[k], rtol=2e-3, atol=1e-5)


def test_g8_mean_aggregator():
    a, _, _ = load_golden("g8_mean_aggregator")
    ip, idx = a["indptr"], a["indices"]
    neighs = [idx[ip[i]:ip[i + 1]].tolist() for i in range(len(ip) - 1)]
    out = O.mean_aggregate(T(a["table"]), a["nodes"].tolist(), neighs, gcn=False)
    close(out, a["out"], rtol=1e-5, atol=1e-6)


def test_g10_adam_two_steps():
    """Pins train.py:173,209-210 (Adam lr 1e-3, clip 2.0) on top of the oracle's fwd/bwd."""
    a, p, _ = load_golden("g10_adam_two_steps")
    P = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(P.values()), lr=0.001)
    x, adj, label = T(a["x"]), T(a["adj"]), T(a["label"])
    for step in range(2):
        opt.zero_grad()
        ypred, inter = O.softpool_forward(P, x, adj, a["num_nodes"], x)
        loss, _ = O.softpool_loss(ypred, label, inter["assign_0"], adj, a["num_nodes"], True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(P.values()), 2.0)
        opt.step()
        close(loss, a["losses"][step], rtol=1e-5, atol=1e-6)
    for k, v in a["after"].items():
        close(P[k], v, rtol=1e-4, atol=1e-6)


def test_multi_pool_unpinned_runs():
    """num_pooling = 2 cannot run in the reference: PARITY UNPINNED.
    This only checks that the build-defined semantics execute and give finite numbers."""
    shapes = O.softpool_param_shapes(max_num_nodes=32, input_dim=5, hidden_dim=8, embedding_dim=8,
                                     label_dim=3, num_layers=3, assign_hidden_dim=8, assign_ratio=0.25,
                                     num_pooling=2)
    P = O.init_params(shapes, seed=3, bias_scale=0.1)
    x, adj, nn_, label = O.make_batch(3, 32, 5, n_min=4, seed=3, n_classes=3)
    ypred, inter = O.softpool_forward(P, x, adj, nn_, x, num_pooling=2, want_intermediates=True)
    assert ypred.shape == (3, 3) and torch.isfinite(ypred).all()
    assert inter["assign_1"].shape == (3, 8, 2)
