"""GCN model surface of the reference (gcn/layers.py:8-41, gcn/models.py:8-46) on the HIP kernels.

Same class names, constructor arguments, parameter names/shapes (``gc1.weight [F,H]``,
``gc1.bias [H]``, ``gc2.weight [H,C]``, ``gc2.bias [C]``) and init law, so a ``state_dict``
trained with the reference loads unchanged (gcn_trainer.py:102-105).  ``forward(x, adj)`` accepts
what the reference passes (a dense float tensor and a torch sparse COO adjacency) or a
``HipGraph``; inference runs through liblinkteller_hip.  Training (autograd through these layers)
is outside the hot path ("OUT OF SCOPE GCNTrainer.train") and is refused
rather than silently served by another backend.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import engine


def _refuse_training(module):
    if module.training and torch.is_grad_enabled():
        raise NotImplementedError(
            "linkteller_amd implements the inference/attack hot path only; call model.eval() and/or "
            "torch.no_grad().  Train with the reference implementation and load its state_dict.")


class GraphConvolution(nn.Module):
    """support = input @ W; output = adj @ support + bias  (reference gcn/layers.py:30-36)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.weight.size(1))     # reference gcn/layers.py:24-28
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, input, adj, relu=False):
        _refuse_training(self)
        support = engine.gemm(input, self.weight.detach())
        return engine.spmm(adj, support, None if self.bias is None else self.bias.detach(), relu=relu)

    def extra_repr(self):
        return f"{self.in_features} -> {self.out_features}"


class GCN(nn.Module):
    """relu(gc1) -> dropout (identity in eval) -> gc2, raw logits (reference gcn/models.py:8-25)."""

    def __init__(self, nfeat, nhid, nclass, dropout):
        super().__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = GraphConvolution(nhid, nclass)
        self.dropout = dropout

    def forward(self, x, adj):
        _refuse_training(self)
        g1, g2 = self.gc1, self.gc2
        fused = (g1.bias is not None and g2.bias is not None and g1.out_features <= 256 and g2.out_features <= 8)
        if fused:   # one GEMM + two fused sparse kernels, H1 never materialised
            return engine.gcn2_forward(adj, x, g1.weight.detach(), g1.bias.detach(),
                                       g2.weight.detach(), g2.bias.detach())
        h = g1(x, adj, relu=True)
        return g2(h, adj)


class GCN3(nn.Module):
    """Three-layer variant reachable with ``--n-layer 3`` (reference gcn/models.py:28-46)."""

    def __init__(self, nfeat, nhid1, nhid2, nclass, dropout):
        super().__init__()
        self.gc1 = GraphConvolution(nfeat, nhid1)
        self.gc2 = GraphConvolution(nhid1, nhid2)
        self.gc3 = GraphConvolution(nhid2, nclass)
        self.dropout = dropout

    def forward(self, x, adj):
        _refuse_training(self)
        h = self.gc1(x, adj, relu=True)
        h = self.gc2(h, adj, relu=True)
        return self.gc3(h, adj)
