"""TU-format graph reader and padded-batch builder (numpy only, no networkx).

Mirrors what the reference's loader + sampler deliver to the model, so real
ENZYMES / DD-format batches can be fed to the HIP encoders:

* node order inside a graph = order of first appearance in ``DS_A.txt`` (the
  reference builds each graph with ``nx.from_edgelist`` and relabels in
  ``G.nodes`` order — load_data.py:61-108); nodes without any edge are dropped,
  as they are there;
* graphs with more than ``max_nodes`` nodes are skipped (load_data.py:79);
* node feature = one-hot node label (train.py:477-481);
* a batch is ``adj [B,N,N]`` (0/1, zero padded), ``feats [B,N,F]`` (zero rows
  for padding), ``num_nodes [B]``, ``label [B]`` (graph_sampler.py:97-109 with
  ``normalize=False`` as cross_val.py:29 passes it).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np


class TUGraph:
    __slots__ = ("adj", "node_label", "label", "node_attr")

    def __init__(self, adj, node_label, label, node_attr=None):
        self.adj = adj                  # [n, n] float32, symmetric 0/1
        self.node_label = node_label    # [n] int (0-based) or None
        self.label = label              # int
        self.node_attr = node_attr      # [n, a] float or None

    @property
    def num_nodes(self) -> int:
        return self.adj.shape[0]


def _read_ints(path: str) -> np.ndarray:
    with open(path) as f:
        return np.array([int(line) for line in f if line.strip() != ""], dtype=np.int64)


def read_tu_graphs(datadir: str, name: str, max_nodes: Optional[int] = None) -> List[TUGraph]:
    prefix = os.path.join(datadir, name, name)
    graph_of_node = _read_ints(prefix + "_graph_indicator.txt")          # 1-based graph id per node
    node_labels = None
    if os.path.exists(prefix + "_node_labels.txt"):
        node_labels = _read_ints(prefix + "_node_labels.txt") - 1         # load_data.py:31
    node_attrs = None
    if os.path.exists(prefix + "_node_attributes.txt"):
        node_attrs = np.loadtxt(prefix + "_node_attributes.txt", delimiter=",", ndmin=2)
    glabels = _read_ints(prefix + "_graph_labels.txt")
    # load_data.py:53-59: labels are shifted to start at 0 unless a 0 label exists
    if not (glabels == 0).any():
        glabels = glabels - 1
    n_graphs = len(glabels)
    edges = np.loadtxt(prefix + "_A.txt", delimiter=",", dtype=np.int64, ndmin=2)
    gid = graph_of_node[edges[:, 0] - 1]
    order = np.argsort(gid, kind="stable")
    edges, gid = edges[order], gid[order]
    starts = np.searchsorted(gid, np.arange(1, n_graphs + 2))
    out: List[TUGraph] = []
    for g in range(n_graphs):
        e = edges[starts[g]:starts[g + 1]]
        if len(e) == 0:
            continue
        flat = e.reshape(-1)
        # first-appearance order of node ids (e0 then e1 of each edge, file order)
        _, first = np.unique(flat, return_index=True)
        nodes = flat[np.sort(first)]
        n = len(nodes)
        if max_nodes is not None and n > max_nodes:
            continue
        local = {int(u): i for i, u in enumerate(nodes)}
        adj = np.zeros((n, n), dtype=np.float32)
        ii = np.fromiter((local[int(u)] for u in e[:, 0]), dtype=np.int64, count=len(e))
        jj = np.fromiter((local[int(u)] for u in e[:, 1]), dtype=np.int64, count=len(e))
        adj[ii, jj] = 1.0
        adj[jj, ii] = 1.0
        nl = node_labels[nodes - 1] if node_labels is not None else None
        na = node_attrs[nodes - 1] if node_attrs is not None else None
        out.append(TUGraph(adj, nl, int(glabels[g]), na))
    return out


def num_node_label_classes(datadir: str, name: str) -> int:
    prefix = os.path.join(datadir, name, name)
    return int(_read_ints(prefix + "_node_labels.txt").max())           # load_data.py:32


MAX_DEG = 10          # one-hot degree features are capped here (graph_sampler.py:45,61)


def node_degrees(adj: np.ndarray) -> np.ndarray:
    return adj.sum(axis=1)


def clustering_coefficients(adj: np.ndarray) -> np.ndarray:
    """Local clustering coefficient of every node of an unweighted undirected graph — what `nx.clustering(G)`
    returns (graph_sampler.py:69): triangles through the node / pairs of neighbours, 0 for degree < 2."""
    a = (adj > 0).astype(np.float64)
    deg = a.sum(axis=1)
    tri = np.einsum("ij,jk,ki->i", a, a, a) / 2.0
    pairs = deg * (deg - 1.0) / 2.0
    return np.where(pairs > 0, tri / np.maximum(pairs, 1.0), 0.0)


def feature_dim(features: str, node_feat_dim: int, max_nodes: int) -> int:
    dims = {"default": node_feat_dim, "id": max_nodes, "deg-num": 1, "deg": MAX_DEG + 1 + node_feat_dim,
            "struct": MAX_DEG + 2 + node_feat_dim}
    if features not in dims:
        raise ValueError(f"unknown feature mode {features!r} (default | id | deg-num | deg | struct)")
    return dims[features]


def graph_features(g: TUGraph, features: str, max_nodes: int, node_feat_dim: int) -> np.ndarray:
    """[max_nodes, feature_dim] node-feature matrix of one graph in the sampler's feature modes
    (graph_sampler.py:33-83); the node features `feat` are the one-hot node labels (train.py:477-481).
      default  one-hot node label                                  (:34-38)
      id       identity of size max_nodes — padded rows included     (:39-40)
      deg-num  the degree as one scalar column                      (:41-45)
      deg      one-hot degree capped at 10, then the node features  (:46-59; the reference's `max_deg` there is an
               undefined name — the cap it means is self.max_deg = 10)
      struct   one-hot capped degree, clustering coefficient, node features   (:60-81)"""
    n = g.num_nodes
    base = np.zeros((max_nodes, node_feat_dim), dtype=np.float32)
    if g.node_label is not None and node_feat_dim > 0:
        base[np.arange(n), g.node_label] = 1.0
    if features == "default":
        return base
    if features == "id":
        return np.identity(max_nodes, dtype=np.float32)
    deg = node_degrees(g.adj)
    if features == "deg-num":
        out = np.zeros((max_nodes, 1), dtype=np.float32)
        out[:n, 0] = deg
        return out
    onehot = np.zeros((max_nodes, MAX_DEG + 1), dtype=np.float32)
    onehot[np.arange(n), np.minimum(deg.astype(np.int64), MAX_DEG)] = 1.0
    if features == "deg":
        return np.concatenate([onehot, base], axis=1)
    if features == "struct":
        clus = np.zeros((max_nodes, 1), dtype=np.float32)
        clus[:n, 0] = clustering_coefficients(g.adj)
        return np.concatenate([onehot, clus, base], axis=1)
    raise ValueError(f"unknown feature mode {features!r} (default | id | deg-num | deg | struct)")


def normalized_adjacency(adj: np.ndarray) -> np.ndarray:
    """D^-1/2 A D^-1/2 (GraphSampler(normalize=True), graph_sampler.py:27-29).  The DiffPool drivers pass
    normalize=False (cross_val.py:29,37); note that a normalised adjacency is not bf16-exact, so the encoders run
    their fp32 aggregation path on it."""
    deg = adj.sum(axis=0)
    inv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-30)), 0.0)
    return (adj * inv[:, None]) * inv[None, :]


def collate(graphs: List[TUGraph], max_nodes: int, feat_dim: int, features: str = "default",
            assign_feat: str = "default", normalize: bool = False) -> Dict[str, np.ndarray]:
    """One padded batch as GraphSampler.__getitem__ + the DataLoader's default collate deliver it
    (graph_sampler.py:97-109).  `feat_dim` is the width of the node features (one-hot node labels); the returned
    `feats` have `feature_dim(features, feat_dim, max_nodes)` columns.  assign_feat='id' prepends the identity to
    the assignment features (graph_sampler.py:85-87)."""
    B = len(graphs)
    fdim = feature_dim(features, feat_dim, max_nodes)
    adj = np.zeros((B, max_nodes, max_nodes), dtype=np.float32)
    feats = np.zeros((B, max_nodes, fdim), dtype=np.float32)
    num_nodes = np.zeros((B,), dtype=np.int32)
    label = np.zeros((B,), dtype=np.int64)
    for b, g in enumerate(graphs):
        n = g.num_nodes
        adj[b, :n, :n] = normalized_adjacency(g.adj) if normalize else g.adj
        feats[b] = graph_features(g, features, max_nodes, feat_dim)
        num_nodes[b] = n
        label[b] = g.label
    assign = feats
    if assign_feat == "id":
        eye = np.broadcast_to(np.identity(max_nodes, dtype=np.float32), (B, max_nodes, max_nodes))
        assign = np.concatenate([eye, feats], axis=2)
    elif assign_feat != "default":
        raise ValueError(f"unknown assign_feat {assign_feat!r} (default | id)")
    return {"adj": adj, "feats": feats, "num_nodes": num_nodes, "label": label, "assign_feats": assign}
