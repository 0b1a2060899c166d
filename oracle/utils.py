"""ORACLE (test infrastructure): ``torch_geometric.utils.{softmax, subgraph}``.

Call sites restated: /root/reference/arxiv_pyg/criterion.py:5,103-113 (softmax),
/root/reference/arxiv_pyg/gnn.py:14,249 (subgraph).
"""
from __future__ import annotations

import torch
from torch import Tensor


def softmax(src: Tensor, index: Tensor, num_nodes: int | None = None) -> Tensor:
    """exp(src - groupmax) / (groupsum + 1e-16), groups given by ``index`` (PyG <=1.7)."""
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    gmax = torch.full((n,), float("-inf"), dtype=src.dtype)
    gmax = gmax.scatter_reduce(0, index, src.detach(), reduce="amax", include_self=True)
    e = (src - gmax[index]).exp()
    gsum = torch.zeros(n, dtype=src.dtype).index_add(0, index, e)
    return e / (gsum[index] + 1e-16)


def subgraph(subset: Tensor, edge_index: Tensor, edge_attr=None, relabel_nodes: bool = False,
             num_nodes: int | None = None):
    """Keep edges with both ends in ``subset`` (original order); relabel id -> position in subset."""
    n = num_nodes if num_nodes is not None else int(max(int(edge_index.max()) + 1 if edge_index.numel() else 0,
                                                       int(subset.max()) + 1 if subset.numel() else 0))
    if subset.dtype == torch.bool:
        in_set = subset
        idx = torch.nonzero(subset).view(-1)
    else:
        in_set = torch.zeros(n, dtype=torch.bool)
        in_set[subset] = True
        idx = subset
    mask = in_set[edge_index[0]] & in_set[edge_index[1]]
    ei = edge_index[:, mask]
    ea = edge_attr[mask] if edge_attr is not None else None
    if relabel_nodes:
        relabel = torch.zeros(n, dtype=torch.int64)
        relabel[idx] = torch.arange(idx.numel(), dtype=torch.int64)
        ei = relabel[ei]
    return ei, ea
