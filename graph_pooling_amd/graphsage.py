"""Supervised GraphSAGE classification head.

Mirrors the public surface of the reference's `SupervisedGraphSage` (graphsage.py:7-26) — constructor
`(num_classes, enc)`, `forward(nodes) -> scores [len(nodes), num_classes]`, `loss(nodes, labels)`, one parameter named
`weight` of shape `[enc.embed_dim, num_classes]` — on top of an encoder that produces node embeddings (here
`graph_pooling_amd.aggregators.MeanAggregator`-based encoders, whose mean aggregation runs in `dp_mean_aggregate_*`).

The reference file cannot run as written: it calls an `init` module it never imports and a
non-existent `nn.softmax`.  Decisions: Xavier-uniform initialisation of `weight` (what `init.xavier_uniform` meant),
and the loss is plain cross-entropy on the raw scores — `CrossEntropyLoss` already applies log-softmax, so the
reference's extra softmax would be a double normalisation.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class SupervisedGraphSage(nn.Module):
    def __init__(self, num_classes: int, enc: nn.Module):
        super().__init__()
        embed_dim = int(getattr(enc, "embed_dim"))
        if num_classes < 1 or embed_dim < 1:
            raise ValueError(f"need num_classes >= 1 and enc.embed_dim >= 1, got {num_classes} / {embed_dim}")
        self.enc = enc
        bound = math.sqrt(6.0 / (embed_dim + num_classes))            # Xavier / Glorot uniform
        self.weight = nn.Parameter(torch.empty(embed_dim, num_classes).uniform_(-bound, bound))

    @property
    def xent(self):
        """Kept for callers that reach for the reference's attribute: the criterion `loss` applies."""
        return F.cross_entropy

    def forward(self, nodes) -> Tensor:
        from . import _lib
        from .sparse import hip_linear
        emb = self.enc(nodes)
        _lib.require_gpu_tensor(emb, "enc(nodes)")            # no CPU path, like every other module of the package
        return hip_linear(emb, self.weight.t())               # the library's fp32 MFMA GEMM (scores = emb @ weight)

    def loss(self, nodes, labels: Tensor) -> Tensor:
        target = labels.reshape(-1).to(dtype=torch.long)
        return F.cross_entropy(self.forward(nodes), target)
