"""``torch_geometric.transforms.ToSparseTensor`` stand-in (/root/reference/arxiv_pyg/gnn.py:237)."""
from __future__ import annotations

import torch

from .sparse import SparseTensor


def to_sparse_tensor(edge_index: torch.Tensor, num_nodes: int) -> SparseTensor:
    """(source, target) edge list -> ``adj_t``: row i lists the sources j of edges j->i, ascending; no dedupe."""
    src, dst = edge_index[0], edge_index[1]
    if edge_index.is_cuda and edge_index.shape[1] < 2 ** 31 - 1:
        from .sparse import csr_from_coo
        rowptr, col = csr_from_coo(dst, src, num_nodes, symmetric=False)   # egnn_csr_from_coo_i64: sort on the device
        out = SparseTensor(rowptr=rowptr, col=col, value=None, sparse_sizes=(num_nodes, num_nodes))
        out._cols_sorted = True
        return out
    perm = torch.argsort(dst * num_nodes + src, stable=True)
    return SparseTensor(row=dst[perm], col=src[perm], value=None, sparse_sizes=(num_nodes, num_nodes), is_sorted=True)


class ToSparseTensor:
    def __call__(self, data):
        n = getattr(data, "num_nodes", None)
        if n is None:
            n = data.x.shape[0]
        data.adj_t = to_sparse_tensor(data.edge_index, int(n))
        data.edge_index = None
        return data


def neighbor_average_features(adj_t: SparseTensor, x: torch.Tensor, R: int):
    """SIGN preprocessing (/root/reference/arxiv_dgl/sign.py:175-186): ``[x, A x, A^2 x, ..., A^R x]`` with A = mean over
    the in-neighbours (``fn.copy_u`` + ``fn.mean``; nodes without in-edges get 0) -- R mean-SpMMs on the GPU."""
    from .ops import spmm_raw
    feats = [x]
    adj = adj_t.set_value(None) if adj_t.has_value() else adj_t
    for _ in range(R):
        feats.append(spmm_raw(adj, feats[-1], "mean")[0])
    return feats
