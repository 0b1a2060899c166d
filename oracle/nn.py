"""ORACLE (test infrastructure): PyG <=1.7 ``GCNConv`` / ``SAGEConv``.

Call sites restated: /root/reference/arxiv_pyg/gnn.py:13,28-35,61-67,92 (SparseTensor input,
``cached=True``) and /root/reference/ppi_pyg/gnn.py:125-132,158-164 (``edge_index`` LongTensor input,
``cached=False``).  Parameter layout is the PyG <=1.7 one: ``GCNConv.weight [in,out]``, ``bias [out]``;
``SAGEConv.lin_l`` (bias) / ``lin_r`` (no bias).
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from .sparse import SparseTensor, gcn_norm_edge_index, gcn_norm_sparse, matmul, ind2ptr


def _edge_index_to_adj_t(edge_index: Tensor, n: int, value=None) -> SparseTensor:
    """Rows = targets (edge_index[1]), columns = sources (edge_index[0]); stable in edge order."""
    src, dst = edge_index[0], edge_index[1]
    perm = torch.argsort(dst, stable=True)
    return SparseTensor(rowptr=ind2ptr(dst[perm], n), col=src[perm],
                        value=None if value is None else value[perm], sparse_sizes=(n, n))


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, cached: bool = False, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.cached = in_channels, out_channels, cached
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._cached_adj_t = None
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))  # glorot
        with torch.no_grad():
            self.weight.uniform_(-a, a)
            if self.bias is not None:
                self.bias.zero_()
        self._cached_adj_t = None

    def forward(self, x: Tensor, adj: "SparseTensor | Tensor") -> Tensor:
        norm = self._cached_adj_t
        if norm is None:
            if isinstance(adj, SparseTensor):
                norm = gcn_norm_sparse(adj)
            else:
                ei, ew = gcn_norm_edge_index(adj, x.shape[0], x.dtype)
                norm = _edge_index_to_adj_t(ei, x.shape[0], ew)
            if self.cached:
                self._cached_adj_t = norm
        out = matmul(norm, x @ self.weight, "sum")
        if self.bias is not None:
            out = out + self.bias
        return out


class SAGEConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean"):
        super().__init__()
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x: Tensor, adj: "SparseTensor | Tensor") -> Tensor:
        if not isinstance(adj, SparseTensor):
            adj = _edge_index_to_adj_t(adj, x.shape[0])
        agg = matmul(adj.set_value(None), x, self.aggr)
        return self.lin_l(agg) + self.lin_r(x)
