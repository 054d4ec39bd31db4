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


def collate(graphs: List[TUGraph], max_nodes: int, feat_dim: int) -> Dict[str, np.ndarray]:
    B = len(graphs)
    adj = np.zeros((B, max_nodes, max_nodes), dtype=np.float32)
    feats = np.zeros((B, max_nodes, feat_dim), dtype=np.float32)
    num_nodes = np.zeros((B,), dtype=np.int32)
    label = np.zeros((B,), dtype=np.int64)
    for b, g in enumerate(graphs):
        n = g.num_nodes
        adj[b, :n, :n] = g.adj
        feats[b, np.arange(n), g.node_label] = 1.0
        num_nodes[b] = n
        label[b] = g.label
    return {"adj": adj, "feats": feats, "num_nodes": num_nodes, "label": label,
            "assign_feats": feats}
