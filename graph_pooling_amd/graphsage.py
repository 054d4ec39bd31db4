"""SupervisedGraphSage head (graphsage.py:7-26) with the reference's undefined names repaired
(`init` is never imported there, `nn.softmax` does not exist — SURVEY.md Appendix B D10):
scores = enc(nodes) @ W;  loss = CrossEntropy(scores, labels)."""
import torch
import torch.nn as nn
from torch.nn import init


class SupervisedGraphSage(nn.Module):
    def __init__(self, num_classes, enc):
        super().__init__()
        self.enc = enc
        self.xent = nn.CrossEntropyLoss()
        self.weight = nn.Parameter(torch.empty(enc.embed_dim, num_classes))
        init.xavier_uniform_(self.weight)

    def forward(self, nodes):
        embeds = self.enc(nodes)
        return embeds.mm(self.weight)

    def loss(self, nodes, labels):
        scores = self.forward(nodes)
        return self.xent(scores, labels.squeeze())
