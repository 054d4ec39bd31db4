"""The sampler's node-feature modes (graph_sampler.py:33-87) as tu_dataset.collate builds them, checked against
networkx (clustering, degrees) and hand-built expectations."""
import networkx as nx
import numpy as np
import pytest

from graph_pooling_amd.tu_dataset import (MAX_DEG, TUGraph, clustering_coefficients, collate, feature_dim,
                                          normalized_adjacency)


def _graphs(seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for n, p in ((1, 0.0), (6, 0.6), (14, 0.9), (9, 0.3)):
        a = np.triu((rng.random((n, n)) < p).astype(np.float32), 1)
        out.append(TUGraph(a + a.T, rng.integers(0, 3, n), int(rng.integers(0, 2))))
    return out


def test_clustering_matches_networkx():
    for g in _graphs():
        G = nx.from_numpy_array(g.adj)
        ref = np.array([nx.clustering(G)[i] for i in range(g.num_nodes)])
        np.testing.assert_allclose(clustering_coefficients(g.adj), ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("mode", ["default", "id", "deg-num", "deg", "struct"])
def test_feature_modes(mode):
    graphs, N, F_ = _graphs(1), 16, 3
    b = collate(graphs, N, F_, features=mode)
    assert b["feats"].shape == (len(graphs), N, feature_dim(mode, F_, N)) and b["assign_feats"] is b["feats"]
    for k, g in enumerate(graphs):
        n = g.num_nodes
        f = b["feats"][k]
        deg = g.adj.sum(1)
        onehot = np.zeros((n, F_), dtype=np.float32)
        onehot[np.arange(n), g.node_label] = 1
        if mode == "default":
            np.testing.assert_array_equal(f[:n], onehot)
            assert not f[n:].any()
        elif mode == "id":
            np.testing.assert_array_equal(f, np.identity(N, dtype=np.float32))      # padded rows included
        elif mode == "deg-num":
            np.testing.assert_array_equal(f[:n, 0], deg)
            assert not f[n:].any()
        else:
            capped = np.minimum(deg.astype(int), MAX_DEG)
            assert (f[:n, :MAX_DEG + 1].argmax(1) == capped).all() and (f[:n, :MAX_DEG + 1].sum(1) == 1).all()
            tail = f[:n, MAX_DEG + 1:]
            if mode == "struct":
                G = nx.from_numpy_array(g.adj)
                np.testing.assert_allclose(tail[:, 0], [nx.clustering(G)[i] for i in range(n)], rtol=1e-6)
                tail = tail[:, 1:]
            np.testing.assert_array_equal(tail, onehot)
            assert not f[n:].any()


def test_degree_cap_and_assign_identity_and_normalisation():
    n = MAX_DEG + 4
    star = np.zeros((n, n), dtype=np.float32)
    star[0, 1:] = star[1:, 0] = 1                       # hub degree 13 > cap
    g = TUGraph(star, np.zeros(n, dtype=np.int64), 0)
    b = collate([g], n + 2, 2, features="deg", assign_feat="id")
    assert b["feats"][0, 0, MAX_DEG] == 1 and b["feats"][0, 1, 1] == 1
    assert b["assign_feats"].shape == (1, n + 2, (n + 2) + MAX_DEG + 1 + 2)
    np.testing.assert_array_equal(b["assign_feats"][0, :, :n + 2], np.identity(n + 2, dtype=np.float32))
    np.testing.assert_array_equal(b["assign_feats"][0, :, n + 2:], b["feats"][0])
    an = normalized_adjacency(star)
    np.testing.assert_allclose(an[0, 1], 1.0 / np.sqrt(13.0 * 1.0), rtol=1e-6)
    bn = collate([g], n, 2, normalize=True)
    np.testing.assert_allclose(bn["adj"][0], an, rtol=1e-6)
    with pytest.raises(ValueError):
        collate([g], n, 2, features="nope")
    with pytest.raises(ValueError):
        collate([g], n, 2, assign_feat="nope")
