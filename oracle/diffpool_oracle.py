"""CPU oracle for the DiffPool hot path (TEST INFRASTRUCTURE — not product code).

This file is a plain-torch (CPU, fp32 or fp64) restatement of the algorithm the
reference JiaxuanYou/graph-pooling runs on its DiffPool path.  It exists only
so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
check / time the HIP path against it.  Nothing under graph_pooling_amd/ may
import it: the product path fails loudly when the HIP extension is missing.

Pinning: the reference holds no tests or golden vectors, so this
restatement is pinned against outputs of the reference's own classes run in
the build container (oracle/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  Rows the reference cannot execute
(num_pooling > 1, MeanAggregator with gcn=True) are "parity unpinned" and say
so where they are tested.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Parameters are passed as a dict keyed by the reference's
state_dict names so fixtures can be exchanged 1:1.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5          # nn.BatchNorm1d default, encoders.py:1051
L2_EPS = 1e-12         # F.normalize default, encoders.py:972
LINK_EPS = 1e-7        # encoders.py:1307


# --------------------------------------------------------------------------- A4
def node_mask(max_nodes: int, num_nodes, dtype=torch.float32) -> Tensor:
    """[B, max_nodes, 1] with ones in the first num_nodes[b] rows.
    Follows encoders.py:1035-1046 (construct_mask)."""
    nn_ = torch.as_tensor(num_nodes).reshape(-1, 1).to(torch.int64)
    idx = torch.arange(max_nodes, dtype=torch.int64).reshape(1, -1)
    return (idx < nn_).to(dtype).unsqueeze(2)


# --------------------------------------------------------------------------- A1
def graph_conv(x: Tensor, adj: Tensor, weight: Tensor, bias: Optional[Tensor],
               add_self: bool = False, normalize: bool = True) -> Tensor:
    """y = l2norm_rows((adj @ x [+ x]) @ W [+ b]).
    Follows the commented-out DiffPool GraphConv.forward, encoders.py:962-974
    (dropout is applied by the caller, encoders.py:963-964)."""
    y = torch.matmul(adj, x)
    if add_self:
        y = y + x
    y = torch.matmul(y, weight)
    if bias is not None:
        y = y + bias
    if normalize:
        # the op the reference names at encoders.py:972; its backward is finite at all-zero rows
        # (padded rows with zero bias), where a hand-written sqrt() would give 0/0
        y = F.normalize(y, p=2, dim=2, eps=L2_EPS)
    return y


class GraphConv(nn.Module):
    """Module form of graph_conv with the ctor the DiffPool classes call
    (encoders.py:946-960 / 1011-1018).  Installed into the imported reference
    as patch P2 by oracle/make_golden.py, so the restatement of A1 is itself
    exercised through the reference's gcn_forward / forward."""

    def __init__(self, input_dim, output_dim, add_self=False, normalize_embedding=False,
                 dropout=0.0, bias=True):
        super().__init__()
        self.add_self = add_self
        self.dropout = dropout
        self.normalize_embedding = normalize_embedding
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.weight = nn.Parameter(torch.zeros(input_dim, output_dim))
        self.bias = nn.Parameter(torch.zeros(output_dim)) if bias else None

    def forward(self, x, adj):
        if self.dropout > 0.001:
            x = F.dropout(x, self.dropout, self.training)
        return graph_conv(x, adj, self.weight, self.bias, self.add_self, self.normalize_embedding)


# --------------------------------------------------------------------------- A3
def bn_node(x: Tensor) -> Tensor:
    """Batch-norm over the node index: for each n, statistics over (batch, feature),
    biased variance, eps 1e-5, no affine, always batch statistics.
    Follows encoders.py:1048-1052 (a fresh BatchNorm1d(N) per call)."""
    mu = x.mean(dim=(0, 2), keepdim=True)
    var = (x - mu).pow(2).mean(dim=(0, 2), keepdim=True)
    return (x - mu) / torch.sqrt(var + BN_EPS)


# --------------------------------------------------------------------------- A2
def _stack_keys(first: str, block: str, last: str, n_layers: int) -> List[str]:
    return [first] + [f"{block}.{i}" for i in range(n_layers - 2)] + [last]


def gcn_stack(x: Tensor, adj: Tensor, params: Dict[str, Tensor], keys: Sequence[str],
              mask: Optional[Tensor], bn: bool = True, add_self: bool = False,
              drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """All-layer GCN with ReLU+BN between layers, concat over layers, optional mask.
    Follows encoders.py:1054-1081 (gcn_forward).  `drop[k]` (values 0 or 1/(1-p), shape of layer k's input) stands
    for the nn.Dropout a GraphConv built with dropout > 0 applies to its input (encoders.py:962-964): the layer sees
    h * drop[k], the concat keeps the un-dropped h."""
    outs = []
    h = x
    for li, k in enumerate(keys):
        hin = h * drop[k] if (drop is not None and k in drop) else h
        h = graph_conv(hin, adj, params[k + ".weight"], params.get(k + ".bias"), add_self, True)
        if li < len(keys) - 1:
            h = torch.relu(h)
            if bn:
                h = bn_node(h)
        outs.append(h)
    z = torch.cat(outs, dim=2)
    if mask is not None:
        z = z * mask
    return z


def mlp_head(v: Tensor, params: Dict[str, Tensor], prefix: str, n_hidden: int) -> Tensor:
    """pred_model: Linear(->h)->ReLU ... ->Linear(->C), or a single Linear when
    pred_hidden_dims == [].  Follows encoders.py:1021-1033."""
    if n_hidden == 0:
        return F.linear(v, params[prefix + ".weight"], params[prefix + ".bias"])
    h = v
    for i in range(n_hidden):
        h = torch.relu(F.linear(h, params[f"{prefix}.{2 * i}.weight"], params[f"{prefix}.{2 * i}.bias"]))
    i = n_hidden
    return F.linear(h, params[f"{prefix}.{2 * i}.weight"], params[f"{prefix}.{2 * i}.bias"])


# ------------------------------------------------------------------ level naming
def level_keys(level: int, num_pooling: int, n_layers: int):
    """state_dict prefixes of pooling level `level` (0-based).
    The LAST level carries the reference's attribute names (encoders.py:1185,
    1206-1210: conv_first2 / assign_conv_first / assign_pred ...); earlier levels
    (only reachable with num_pooling > 1, which the reference cannot run —
    SURVEY.md Appendix B D2-D4) use build-defined names."""
    if level == num_pooling - 1:
        emb = _stack_keys("conv_first2", "conv_block2", "conv_last2", n_layers)
        asg = _stack_keys("assign_conv_first", "assign_conv_block", "assign_conv_last", n_layers)
        pred = "assign_pred"
    else:
        emb = _stack_keys(f"conv_first_after_pool_{level}", f"conv_block_after_pool_{level}",
                          f"conv_last_after_pool_{level}", n_layers)
        asg = _stack_keys(f"assign_conv_first_{level}", f"assign_conv_block_{level}",
                          f"assign_conv_last_{level}", n_layers)
        pred = f"assign_pred_{level}"
    return emb, asg, pred


# ------------------------------------------------------------------- A5, A6, A7
def softpool_forward(params: Dict[str, Tensor], x: Tensor, adj: Tensor, num_nodes,
                     assign_x: Optional[Tensor] = None, *, num_layers: int = 3,
                     num_pooling: int = 1, n_pred_hidden: int = 1, bn: bool = True,
                     want_intermediates: bool = False, drop: Optional[Dict[str, Tensor]] = None,
                     winners: Optional[Sequence[Tensor]] = None):
    """SoftPoolingGcnEncoder.forward, encoders.py:1231-1300, with the index fixes of
    SURVEY.md Appendix B (D3: per-level assign_pred; D4: level>=1 assign input = X').

    Returns ypred (and a dict of intermediates).  `bn` is the *after-pool / assign*
    flag; the level-0 embedding GCN always batch-norms (encoders.py:1172-1173 does
    not forward `bn`, so self.bn stays True — and self.bn is what gcn_forward reads
    for every stack, encoders.py:1063).  Hence bn here is effectively always True
    for SoftPoolingGcnEncoder; the argument exists for the base encoders.

    `winners` (tests only): per level an int tensor [B, D] of row indices that REPLACE the arg-max of the max
    readout (-1: the output is the constant 0 of a masked row).  torch.max routes the gradient to whichever row wins;
    two rows tied to within rounding can swap between an fp32 and an fp64 run, which changes whole gradient entries
    although both are valid.  Forcing the decisions of the run under test makes an fp64 run of this function the exact
    gradient of THAT forward pass."""
    def readout(z, level):
        if winners is None:
            return z.max(dim=1)[0]
        idx = winners[level].to(torch.int64)
        got = z.gather(1, idx.clamp_min(0).unsqueeze(1)).squeeze(1)
        return torch.where(idx >= 0, got, torch.zeros((), dtype=z.dtype))
    B, N, _ = x.shape
    x_a = x if assign_x is None else assign_x
    mask = node_mask(N, num_nodes, x.dtype) if num_nodes is not None else None
    inter = {}
    emb0 = _stack_keys("conv_first", "conv_block", "conv_last", num_layers)
    z = gcn_stack(x, adj, params, emb0, mask, bn, drop=drop)            # :1254
    outs = [readout(z, 0)]                                              # :1257
    s0 = None
    for i in range(num_pooling):                                        # :1263
        emb_k, asg_k, pred_k = level_keys(i, num_pooling, num_layers)
        m = mask if i == 0 else None                                    # :1264-1267
        za = gcn_stack(x_a, adj, params, asg_k, m, bn, drop=drop)       # :1269-1271
        logits = F.linear(za, params[pred_k + ".weight"], params[pred_k + ".bias"])
        s = torch.softmax(logits, dim=-1)                               # :1273
        if m is not None:
            s = s * m                                                   # :1274-1275
        if i == 0:
            s0 = s
        xp = torch.matmul(s.transpose(1, 2), z)                         # :1278
        adj = s.transpose(1, 2) @ adj @ s                               # :1279
        x_a = xp                                                        # :1280
        if want_intermediates:
            inter[f"assign_{i}"] = s
            inter[f"xpool_{i}"] = xp
            inter[f"adjpool_{i}"] = adj
        z = gcn_stack(xp, adj, params, emb_k, None, bn, drop=drop)      # :1282-1284
        outs.append(readout(z, i + 1))                                  # :1287
        inter["assign_last"] = s
    feat = torch.cat(outs, dim=1)                                       # :1295-1296
    ypred = mlp_head(feat, params, "pred_model", n_pred_hidden)         # :1299
    inter["assign_0"] = s0
    inter["readout"] = feat
    return ypred, inter


def link_pred_loss(s: Tensor, adj: Tensor, num_nodes) -> Tensor:
    """-A log(P+eps) - (1-A) log(1-P+eps), P = min(S S^T, 1), zero outside the
    n_b x n_b block, summed and divided by sum_b n_b^2.
    Follows encoders.py:1309-1331 with adj_hop = 1; the clamp constant is 1.0
    (Appendix B D5) and the mask is boolean (D6)."""
    B, N, _ = s.shape
    p = torch.minimum(s @ s.transpose(1, 2), torch.ones((), dtype=s.dtype))
    ll = -adj * torch.log(p + LINK_EPS) - (1 - adj) * torch.log(1 - p + LINK_EPS)
    if num_nodes is None:
        return ll.sum() / float(N * N * B)                               # :1323
    m = node_mask(N, num_nodes, s.dtype)
    ll = ll * (m @ m.transpose(1, 2))
    nn_ = torch.as_tensor(num_nodes).to(torch.float64)
    return ll.sum() / float((nn_ * nn_).sum())                           # :1326,1331


def softpool_loss(ypred: Tensor, label: Tensor, s0: Optional[Tensor] = None,
                  adj: Optional[Tensor] = None, num_nodes=None, linkpred: bool = False):
    """SoftPoolingGcnEncoder.loss, encoders.py:1302-1334 (+ base loss :1124-1127).
    Returns (total, link) — link is None when linkpred is off."""
    ce = F.cross_entropy(ypred, label, reduction="mean")
    if not linkpred:
        return ce, None
    link = link_pred_loss(s0, adj, num_nodes)
    return ce + link, link


# -------------------------------------------------------------------------- A11
def base_forward(params: Dict[str, Tensor], x: Tensor, adj: Tensor, *, num_layers: int = 3,
                 n_pred_hidden: int = 0, bn: bool = True, concat: bool = True,
                 drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """GcnEncoderGraph.forward, encoders.py:1083-1122: per-layer max readout, no
    masking at all (the mask is built at :1087 but never used).  `drop`: see gcn_stack."""
    add_self = not concat
    keys = _stack_keys("conv_first", "conv_block", "conv_last", num_layers)
    h = x
    outs = []
    for li, k in enumerate(keys):
        hin = h * drop[k] if (drop is not None and k in drop) else h
        h = graph_conv(hin, adj, params[k + ".weight"], params.get(k + ".bias"), add_self, True)
        if li < num_layers - 1:
            h = torch.relu(h)
            if bn:
                h = bn_node(h)
        outs.append(h.max(dim=1)[0])
    feat = torch.cat(outs, dim=1) if concat else outs[-1]
    return mlp_head(feat, params, "pred_model", n_pred_hidden)


# -------------------------------------------------------------------------- A10
def lstm_cell(xt: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Tensor, b_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """One step of nn.LSTM (gate order i, f, g, o)."""
    g = xt @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    d = h.shape[1]
    i, f, gg, o = g[:, :d], g[:, d:2 * d], g[:, 2 * d:3 * d], g[:, 3 * d:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def set2set_forward(emb: Tensor, params: Dict[str, Tensor], prefix: str = "") -> Tensor:
    """Set2Set.forward, set2set.py:32-57: n = emb.shape[1] steps of
    {LSTM(q*), e = emb.q, a = softmax over ALL n rows, r = sum a*emb, q* = [q, r]},
    then ReLU(Linear(q*)).  Zero initial state (set2set.py:42-45)."""
    B, n, d = emb.shape
    w_ih, w_hh = params[prefix + "lstm.weight_ih_l0"], params[prefix + "lstm.weight_hh_l0"]
    b_ih, b_hh = params[prefix + "lstm.bias_ih_l0"], params[prefix + "lstm.bias_hh_l0"]
    h = emb.new_zeros(B, d)
    c = emb.new_zeros(B, d)
    qs = emb.new_zeros(B, 2 * d)
    for _ in range(n):
        h, c = lstm_cell(qs, h, c, w_ih, w_hh, b_ih, b_hh)
        e = torch.einsum("bnd,bd->bn", emb, h)
        a = torch.softmax(e, dim=1)
        r = torch.einsum("bn,bnd->bd", a, emb)
        qs = torch.cat([h, r], dim=1)
    return torch.relu(F.linear(qs, params[prefix + "pred.weight"], params[prefix + "pred.bias"]))


def set2set_encoder_forward(params: Dict[str, Tensor], x: Tensor, adj: Tensor, num_nodes, *,
                            num_layers: int = 3, n_pred_hidden: int = 0, bn: bool = True) -> Tensor:
    """GcnSet2SetEncoder.forward, encoders.py:1144-1157."""
    mask = node_mask(x.shape[1], num_nodes, x.dtype) if num_nodes is not None else None
    keys = _stack_keys("conv_first", "conv_block", "conv_last", num_layers)
    z = gcn_stack(x, adj, params, keys, mask, bn)
    out = set2set_forward(z, params, "s2s.")
    return mlp_head(out, params, "pred_model", n_pred_hidden)


# --------------------------------------------------------------------------- A9
def mean_aggregate(features: Tensor, nodes: Sequence[int], to_neighs: Sequence[Sequence[int]],
                   gcn: bool = False) -> Tensor:
    """MeanAggregator.forward with num_sample=None, aggregators.py:30-63: row i is the
    mean of features over the neighbour SET of nodes[i] (plus the node itself when
    gcn=True — the reference's `set + set` at :47 raises; Appendix B D10 intended
    semantics)."""
    rows = []
    for i, ng in enumerate(to_neighs):
        s = set(int(v) for v in ng)
        if gcn:
            s = s | {int(nodes[i])}
        idx = torch.tensor(sorted(s), dtype=torch.int64)
        rows.append(features[idx].sum(dim=0) / float(len(s)))
    return torch.stack(rows, dim=0)


# ---------------------------------------------------------------- param helpers
def softpool_param_shapes(*, max_num_nodes: int, input_dim: int, hidden_dim: int, embedding_dim: int,
                          label_dim: int, num_layers: int, assign_hidden_dim: int,
                          assign_ratio: float = 0.25, num_pooling: int = 1,
                          pred_hidden_dims: Sequence[int] = (50,), assign_input_dim: int = -1,
                          bias: bool = True) -> Dict[str, Tuple[int, ...]]:
    """Shapes by state_dict key (encoders.py:1161-1229)."""
    L = num_layers
    D = hidden_dim * (L - 1) + embedding_dim
    if assign_input_dim == -1:
        assign_input_dim = input_dim
    shapes: Dict[str, Tuple[int, ...]] = {}

    def stack(keys, fin, hid, fout):
        dims = [fin] + [hid] * (L - 1) + [fout]
        for li, k in enumerate(keys):
            shapes[k + ".weight"] = (dims[li], dims[li + 1])
            if bias:
                shapes[k + ".bias"] = (dims[li + 1],)

    stack(_stack_keys("conv_first", "conv_block", "conv_last", L), input_dim, hidden_dim, embedding_dim)
    k_dim = int(max_num_nodes * assign_ratio)
    a_in = assign_input_dim
    for i in range(num_pooling):
        emb_k, asg_k, pred_k = level_keys(i, num_pooling, L)
        stack(emb_k, D, hidden_dim, embedding_dim)
        stack(asg_k, a_in, assign_hidden_dim, k_dim)
        d_a = assign_hidden_dim * (L - 1) + k_dim
        shapes[pred_k + ".weight"] = (k_dim, d_a)
        shapes[pred_k + ".bias"] = (k_dim,)
        a_in = D                      # Appendix B D4
        k_dim = int(k_dim * assign_ratio)
    pin = D * (num_pooling + 1)
    if len(pred_hidden_dims) == 0:
        shapes["pred_model.weight"] = (label_dim, pin)
        shapes["pred_model.bias"] = (label_dim,)
    else:
        for i, h in enumerate(pred_hidden_dims):
            shapes[f"pred_model.{2 * i}.weight"] = (h, pin)
            shapes[f"pred_model.{2 * i}.bias"] = (h,)
            pin = h
        i = len(pred_hidden_dims)
        shapes[f"pred_model.{2 * i}.weight"] = (label_dim, pin)
        shapes[f"pred_model.{2 * i}.bias"] = (label_dim,)
    return shapes


def init_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32,
                bias_scale: float = 0.0) -> Dict[str, Tensor]:
    """Reference-style init: xavier_uniform(gain=sqrt 2) on GraphConv weights, zero GraphConv
    bias (encoders.py:1225-1229); nn.Linear-style uniform elsewhere.  `bias_scale` > 0
    perturbs GraphConv biases so tests exercise the 'trained' regime where padded rows
    are non-zero."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        is_gc = k.startswith(("conv_", "assign_conv_"))
        if k.endswith(".weight"):
            if is_gc:
                fan_in, fan_out = shp
                a = math.sqrt(2.0) * math.sqrt(6.0 / (fan_in + fan_out))
            else:
                a = 1.0 / math.sqrt(shp[1])
            out[k] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * a).to(dtype)
        else:
            if is_gc:
                out[k] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bias_scale).to(dtype)
            else:
                wk = k[:-5] + ".weight"
                a = 1.0 / math.sqrt(shapes[wk][1])
                out[k] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * a).to(dtype)
    return out


# ------------------------------------------------------------- synthetic batches
def make_batch(B: int, N: int, F_: int, *, n_min: int, n_max: Optional[int] = None, p: float = 0.1,
               n_classes: int = 2, onehot: bool = True, seed: int = 1, dtype=torch.float32,
               sizes: Optional[Sequence[int]] = None):
    """Seeded synthetic padded batch with the layout contract of graph_sampler.py:97-109
    (SURVEY.md §8(d), A.1): symmetric 0/1 zero-diagonal ER adjacency in the top-left
    n_b x n_b block, zero feature rows for n >= n_b."""
    g = torch.Generator().manual_seed(seed)
    n_max = N if n_max is None else n_max
    if sizes is None:
        sizes = torch.randint(n_min, n_max + 1, (B,), generator=g).tolist()
    adj = torch.zeros(B, N, N, dtype=dtype)
    x = torch.zeros(B, N, F_, dtype=dtype)
    for b, n in enumerate(sizes):
        u = (torch.rand(n, n, generator=g) < p).to(dtype)
        u = torch.triu(u, diagonal=1)
        adj[b, :n, :n] = u + u.t()
        if onehot:
            cls = torch.randint(0, F_, (n,), generator=g)
            x[b, torch.arange(n), cls] = 1.0
        else:
            x[b, :n] = torch.randn(n, F_, generator=g).to(dtype)
    label = torch.randint(0, n_classes, (B,), generator=g)
    import numpy as np
    return x, adj, np.asarray(sizes, dtype=np.int32), label
