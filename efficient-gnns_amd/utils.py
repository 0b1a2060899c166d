"""``torch_geometric.utils.{softmax, subgraph}`` stand-ins (/root/reference/arxiv_pyg/criterion.py:5,103-113;
/root/reference/arxiv_pyg/gnn.py:14,249)."""
from __future__ import annotations

import torch
from torch import Tensor


def subgraph(subset: Tensor, edge_index: Tensor, edge_attr=None, relabel_nodes: bool = False, num_nodes: int | None = None):
    """Edges with both endpoints in ``subset`` (original order); ids relabelled to positions in ``subset``.
    Integer preprocessing (once per process in the reference); runs on the device the indices live on."""
    dev = edge_index.device
    if num_nodes is None:
        num_nodes = int(max(int(edge_index.max()) + 1 if edge_index.numel() else 0,
                            (int(subset.max()) + 1 if subset.dtype != torch.bool else subset.numel()) if subset.numel() else 0))
    if subset.dtype == torch.bool:
        in_set, idx = subset, torch.nonzero(subset).view(-1)
    else:
        in_set = torch.zeros(num_nodes, dtype=torch.bool, device=dev)
        in_set[subset] = True
        idx = subset
    mask = in_set[edge_index[0]] & in_set[edge_index[1]]
    ei = edge_index[:, mask]
    ea = edge_attr[mask] if edge_attr is not None else None
    if relabel_nodes:
        relabel = torch.zeros(num_nodes, dtype=torch.int64, device=dev)
        relabel[idx] = torch.arange(idx.numel(), dtype=torch.int64, device=dev)
        ei = relabel[ei]
    return ei, ea


def softmax(src: Tensor, index: Tensor, ptr=None, num_nodes: int | None = None) -> Tensor:
    """Segment softmax over entries grouped by ``index``: exp(src - max) / (sum + 1e-16)."""
    from . import _lib
    from .ops_edge import segment_softmax
    return segment_softmax(_lib.real(src), index, num_nodes)


def dgl_bidirected_with_self_loops(adj_t):
    """The message graph of the arxiv GAT teacher (/root/reference/arxiv_dgl/gat.py:56-71 ``preprocess``):
    ``dgl.to_bidirected`` (union with the reversed edges, duplicates merged), ``remove_self_loop().add_self_loop()`` (exactly
    one loop per node) -- as a ``SparseTensor`` whose row i lists the sources of the edges into i, columns ascending."""
    import torch
    from .sparse import SparseTensor
    sym = adj_t.to_symmetric()
    rowptr, col, _ = sym.csr()
    n = sym.sparse_size(0)
    row = sym.storage.row()
    keep = row != col
    loops = torch.arange(n, dtype=col.dtype, device=col.device)
    r, c = torch.cat([row[keep], loops]), torch.cat([col[keep], loops])
    return SparseTensor(row=r, col=c, sparse_sizes=(n, n))   # sorted by (row, col) in the constructor
