"""GCN graph encoder on CSR graphs — for graphs beyond the padded dense path.

The reference pads every graph to `max_nodes` and DROPS the ones above it (load_data.py:79: `if max_nodes is not None
and G.number_of_nodes() > max_nodes: continue`); DD's largest graph has 5 748 nodes (132 MB as a dense fp32 block).
`SparseGcnEncoderGraph` runs the same model as `GcnEncoderGraph` (encoders.py:976-1134: GraphConv -> ReLU -> apply_bn per
layer, max readout per layer, concat, pred_model) on ONE graph given as CSR, so such graphs can be classified with
the parameters trained on the dense path: same constructor arguments, same `state_dict` keys, and on a graph that
fits the dense path the same numbers (tests/test_gpu_sparse.py checks it against the oracle's dense restatement).

The neighbour sum  A x  is the CSR gather of dp_csr_aggregate (the MeanAggregator's kernel with mean = 0,
aggregators.py:50-62), the transform, bias, l2-normalisation and their backward the kernels of the dense path
(dp_sparse_gcn_layer_fwd / bwd); apply_bn on a single graph is dp_bn_node_* with B = 1; the Linear layers of the
prediction head run on dp_bgemm_f32 (`hip_linear`).  No torch arithmetic on the path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .encoders import GraphConv


# ----------------------------------------------------------------------------- Linear on the HIP GEMM
class _LinearFn(torch.autograd.Function):
    """y = x W^T + b (nn.Linear layout, W [out, in]) on dp_bgemm_f32 — forward and both gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _lib.require_gpu_tensor(x, "x")
        x = x.contiguous().float()
        w = weight.contiguous()
        rows, fin = x.shape
        fout = w.shape[0]
        y = torch.empty(rows, fout, device=x.device, dtype=torch.float32)
        st = _lib.current_stream()
        _lib.check(lib.dp_bgemm_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), _lib.ptr(bias), 1, rows, fout, fin, fin, fin,
                                    fout, 0, 0, 0, 0, 1, 1.0, 0.0, 0, st), "dp_bgemm_f32")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        rows, fin = x.shape
        fout = w.shape[0]
        st = _lib.current_stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)          # dx = dy W
            _lib.check(lib.dp_bgemm_f32(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), None, 1, rows, fin, fout, fout, fin,
                                        fin, 0, 0, 0, 0, 0, 1.0, 0.0, 0, st), "dp_bgemm_f32")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)          # dW = dy^T x
            _lib.check(lib.dp_bgemm_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, 1, fout, fin, rows, fout, fin,
                                        fin, 0, 0, 0, 1, 0, 1.0, 0.0, 0, st), "dp_bgemm_f32")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ones = torch.ones(1, rows, device=x.device, dtype=torch.float32)      # db = 1^T dy
            db = torch.empty(fout, device=x.device, dtype=torch.float32)
            _lib.check(lib.dp_bgemm_f32(ones.data_ptr(), dy.data_ptr(), db.data_ptr(), None, 1, 1, fout, rows, rows,
                                        fout, fout, 0, 0, 0, 0, 0, 1.0, 0.0, 0, st), "dp_bgemm_f32")
        return dx, dw, db


def hip_linear(x, weight, bias=None):
    """nn.Linear's arithmetic (x @ weight.T + bias) on the library's fp32 MFMA GEMM; x [rows, in]."""
    return _LinearFn.apply(x, weight, bias)


# ----------------------------------------------------------------------------- CSR helpers
class CsrGraph:
    """CSR of one graph's adjacency on the device: int32 indptr [n + 1], indices [nnz] — row i lists the j with
    A[i, j] != 0 (0/1 adjacency, graph_sampler.py:26) — plus the CSR of A^T for the backward gather (the same arrays
    for an undirected graph)."""

    def __init__(self, indptr, indices, indptr_t=None, indices_t=None):
        self.indptr, self.indices = indptr, indices
        self.indptr_t = indptr if indptr_t is None else indptr_t
        self.indices_t = indices if indices_t is None else indices_t
        self.n = indptr.numel() - 1

    @staticmethod
    def from_edges(n, src, dst, device, symmetric=True):
        """Edge list -> CSR (both directions when symmetric, duplicates removed, no self loops added)."""
        src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
        if symmetric:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
        key = np.unique(src * n + dst)
        rows, cols = key // n, key % n

        def csr(r, c):
            order = np.lexsort((c, r))
            r, c = r[order], c[order]
            indptr = np.zeros(n + 1, dtype=np.int32)
            np.add.at(indptr, r + 1, 1)
            return (torch.from_numpy(np.cumsum(indptr, dtype=np.int64).astype(np.int32)).to(device),
                    torch.from_numpy(c.astype(np.int32)).to(device))
        ip, ix = csr(rows, cols)
        if symmetric:
            return CsrGraph(ip, ix)
        ipt, ixt = csr(cols, rows)
        return CsrGraph(ip, ix, ipt, ixt)

    @staticmethod
    def from_dense(adj):
        """[n, n] tensor (any device) -> CSR on adj's device (testing aid)."""
        a = adj.detach().cpu().numpy() != 0
        r, c = np.nonzero(a)
        return CsrGraph.from_edges(a.shape[0], r, c, adj.device, symmetric=False)


class _SparseGraphConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, g, flags):
        lib = _lib.load()
        _lib.require_gpu_tensor(x, "x")
        x = x.contiguous().float()
        w = weight.contiguous()
        n, fin = x.shape
        fout = w.shape[1]
        y = torch.empty(n, fout, device=x.device, dtype=torch.float32)
        ax = torch.empty(n, fin, device=x.device, dtype=torch.float32)
        invn = torch.empty(n, device=x.device, dtype=torch.float32)
        wsb = lib.dp_sparse_gcn_layer_workspace_bytes(n, fin, fout)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        _lib.check(lib.dp_sparse_gcn_layer_fwd(x.data_ptr(), fin, g.indptr.data_ptr(), g.indices.data_ptr(), w.data_ptr(),
                                               _lib.ptr(bias), y.data_ptr(), fout, ax.data_ptr(), invn.data_ptr(), n,
                                               fin, fout, flags, ws.data_ptr(), wsb, _lib.current_stream()),
                   "dp_sparse_gcn_layer_fwd")
        ctx.save_for_backward(ax, w, y, invn)
        ctx.g, ctx.flags, ctx.ws, ctx.has_bias = g, flags, ws, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        ax, w, y, invn = ctx.saved_tensors
        g = ctx.g
        n, fin = ax.shape
        fout = w.shape[1]
        dy = dy.contiguous()
        dx = torch.empty_like(ax) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(fout, device=ax.device, dtype=torch.float32) if ctx.has_bias else None
        _lib.check(lib.dp_sparse_gcn_layer_bwd(ax.data_ptr(), g.indptr.data_ptr(), g.indices.data_ptr(),
                                               g.indptr_t.data_ptr(), g.indices_t.data_ptr(), w.data_ptr(), y.data_ptr(),
                                               fout, invn.data_ptr(), dy.data_ptr(), fout, _lib.ptr(dx), fin,
                                               dw.data_ptr(), _lib.ptr(db), n, fin, fout, ctx.flags, ctx.ws.data_ptr(),
                                               ctx.ws.numel(), _lib.current_stream()), "dp_sparse_gcn_layer_bwd")
        return dx, dw, db, None, None


class _BnNodeFn(torch.autograd.Function):
    """apply_bn (encoders.py:1048-1052) after ReLU on ONE graph: dp_bn_node_* with a batch of one."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        n, f = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(n, 2, device=x.device, dtype=torch.float32)
        wsb = lib.dp_bn_node_workspace_bytes(1, n, f)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        _lib.check(lib.dp_bn_node_fwd(x.data_ptr(), f, y.data_ptr(), f, stats.data_ptr(), 1, n, f, 1, ws.data_ptr(), wsb,
                                      _lib.current_stream()), "dp_bn_node_fwd")
        ctx.save_for_backward(x, y, stats)
        ctx.ws = ws
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, stats = ctx.saved_tensors
        n, f = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.dp_bn_node_bwd(x.data_ptr(), f, y.data_ptr(), f, stats.data_ptr(), dy.data_ptr(), f, dx.data_ptr(),
                                      f, 1, n, f, 1, ctx.ws.data_ptr(), ctx.ws.numel(), _lib.current_stream()),
                   "dp_bn_node_bwd")
        return dx


class _RowMaxFn(torch.autograd.Function):
    """max over the node rows (torch.max(x, dim=1), encoders.py:1093) through dp_masked_max_* with B = 1."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        n, f = x.shape
        out = torch.empty(1, f, device=x.device, dtype=torch.float32)
        arg = torch.empty(1, f, device=x.device, dtype=torch.int32)
        _lib.check(lib.dp_masked_max_fwd(x.data_ptr(), f, None, out.data_ptr(), f, arg.data_ptr(), 1, n, f,
                                         _lib.current_stream()), "dp_masked_max_fwd")
        ctx.save_for_backward(arg)
        ctx.shape = (n, f)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        n, f = ctx.shape
        dx = torch.zeros(n, f, device=dout.device, dtype=torch.float32)
        dout = dout.contiguous()
        _lib.check(lib.dp_masked_max_bwd(dout.data_ptr(), f, arg.data_ptr(), dx.data_ptr(), f, 1, n, f,
                                         _lib.current_stream()), "dp_masked_max_bwd")
        return dx


class SparseGcnEncoderGraph(nn.Module):
    """`GcnEncoderGraph` (encoders.py:976-1134) on one CSR graph: forward(x [n, F], graph) -> ypred [1, label_dim].

    Same constructor arguments and `state_dict` keys as the dense class (conv_first / conv_block.i / conv_last /
    pred_model...), so parameters move between the two with load_state_dict."""

    def __init__(self, input_dim, hidden_dim, embedding_dim, label_dim, num_layers, pred_hidden_dims=[], concat=True,
                 bn=True, dropout=0.0, args=None):
        super().__init__()
        if dropout > 0.001:
            raise NotImplementedError("dropout on the CSR path")
        self.concat, self.bn, self.num_layers, self.label_dim = concat, bn, num_layers, label_dim
        bias = True if args is None else args.bias
        add_self = not concat
        self.conv_first = GraphConv(input_dim, hidden_dim, add_self=add_self, normalize_embedding=True, bias=bias)
        self.conv_block = nn.ModuleList([GraphConv(hidden_dim, hidden_dim, add_self=add_self, normalize_embedding=True,
                                                   bias=bias) for _ in range(num_layers - 2)])
        self.conv_last = GraphConv(hidden_dim, embedding_dim, add_self=add_self, normalize_embedding=True, bias=bias)
        pin = hidden_dim * (num_layers - 1) + embedding_dim if concat else embedding_dim
        if len(pred_hidden_dims) == 0:
            self.pred_model = nn.Linear(pin, label_dim)
        else:
            layers = []
            for d in pred_hidden_dims:
                layers += [nn.Linear(pin, d), nn.ReLU()]
                pin = d
            layers.append(nn.Linear(pin, label_dim))
            self.pred_model = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, GraphConv):
                nn.init.xavier_uniform_(m.weight.data, gain=nn.init.calculate_gain('relu'))
                if m.bias is not None:
                    nn.init.constant_(m.bias.data, 0.0)

    def _conv(self, m, x, g):
        flags = (_lib.F_ADD_SELF if m.add_self else 0) | (_lib.F_NORMALIZE if m.normalize_embedding else 0)
        return _SparseGraphConvFn.apply(x, m.weight, m.bias, g, flags)

    def forward(self, x, graph: CsrGraph):
        if x.dim() != 2 or x.shape[0] != graph.n:
            raise ValueError(f"expected x [n, F] with n = {graph.n}, got {tuple(x.shape)}")
        outs = []
        h = x
        for m in [self.conv_first] + list(self.conv_block):
            h = self._conv(m, h, graph)
            h = _BnNodeFn.apply(h) if self.bn else torch.relu(h)      # ReLU is fused into the BN kernel
            outs.append(_RowMaxFn.apply(h))
        outs.append(_RowMaxFn.apply(self._conv(self.conv_last, h, graph)))
        feat = torch.cat(outs, dim=1) if self.concat else outs[-1]
        if isinstance(self.pred_model, nn.Linear):
            return hip_linear(feat, self.pred_model.weight, self.pred_model.bias)
        h = feat
        lins = [m for m in self.pred_model if isinstance(m, nn.Linear)]
        for i, lin in enumerate(lins):
            h = hip_linear(h, lin.weight, lin.bias)
            if i < len(lins) - 1:
                h = torch.relu(h)
        return h

    @torch.no_grad()
    def predict(self, x, graph):
        return self.forward(x, graph).argmax(dim=1)
