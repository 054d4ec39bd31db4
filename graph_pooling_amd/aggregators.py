"""GraphSAGE mean aggregator on MI355X behind the reference's surface (aggregators.py:11-63).

The reference builds a dense 0/1 ``[n_batch, n_unique]`` mask, row-normalises it and multiplies it
with the gathered embedding matrix (aggregators.py:50-62).  Here the neighbour sets become a CSR
structure on the host (the sets are host Python objects in the reference too) and the mean is one
gather kernel (dp_mean_aggregate_fwd); the result is the same matrix product.
"""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class _MeanAggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, indptr, indices):
        lib = _lib.load()
        _lib.require_gpu_tensor(table, "features(unique_nodes)")
        table = table.contiguous().float()
        n_rows = indptr.numel() - 1
        feat = table.shape[1]
        out = torch.empty(n_rows, feat, device=table.device, dtype=torch.float32)
        _lib.check(lib.dp_mean_aggregate_fwd(table.data_ptr(), feat, indptr.data_ptr(), indices.data_ptr(),
                                             out.data_ptr(), feat, n_rows, feat, _lib.current_stream()),
                   "dp_mean_aggregate_fwd")
        ctx.save_for_backward(indptr, indices)
        ctx.shape = tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        indptr, indices = ctx.saved_tensors
        n_table, feat = ctx.shape
        dout = dout.contiguous()
        dtable = torch.zeros(n_table, feat, device=dout.device, dtype=torch.float32)
        _lib.check(lib.dp_mean_aggregate_bwd(dout.data_ptr(), feat, indptr.data_ptr(), indices.data_ptr(),
                                             dtable.data_ptr(), feat, indptr.numel() - 1, feat,
                                             _lib.current_stream()), "dp_mean_aggregate_bwd")
        return dtable, None, None


def mean_aggregate(table: torch.Tensor, indptr: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """out[i] = mean(table[indices[indptr[i]:indptr[i+1]]]); int32 CSR tensors on the GPU."""
    return _MeanAggFn.apply(table, indptr, indices)


class MeanAggregator(nn.Module):
    """Aggregates a node's embeddings using the mean of its neighbours' embeddings."""

    def __init__(self, features, cuda=False, gcn=False):
        super().__init__()
        self.features = features
        self.cuda = cuda          # kept for signature parity; the kernel always runs on the GPU
        self.gcn = gcn

    def forward(self, nodes, to_neighs, num_sample=10):
        if num_sample is not None:
            # aggregators.py:38-42 (random.sample on a set needs a sequence on Python >= 3.11)
            samp_neighs = [set(random.sample(sorted(tn), num_sample)) if len(tn) >= num_sample else set(tn)
                           for tn in to_neighs]
        else:
            samp_neighs = [set(tn) for tn in to_neighs]
        if self.gcn:
            # aggregators.py:47 does `set + set` (TypeError); intended: add the node itself (Appendix B D10)
            samp_neighs = [sn | {nodes[i]} for i, sn in enumerate(samp_neighs)]
        unique_nodes_list = list(set.union(*samp_neighs))
        unique_nodes = {n: i for i, n in enumerate(unique_nodes_list)}
        indptr = np.zeros(len(samp_neighs) + 1, dtype=np.int32)
        indptr[1:] = np.cumsum([len(s) for s in samp_neighs])
        indices = np.fromiter((unique_nodes[n] for s in samp_neighs for n in s), dtype=np.int32,
                              count=int(indptr[-1]))
        embed_matrix = self.features(torch.LongTensor(unique_nodes_list).to("cuda"))
        dev = embed_matrix.device
        return mean_aggregate(embed_matrix, torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
