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


def reorder_nodes(data, perm: torch.Tensor):
    """Apply a node permutation (new node i = old node perm[i]) to a whole node-classification problem in place: ``adj_t``,
    the per-node tensors (x, y, teacher artefacts) and the split index sets.  Done once per dataset; every later epoch runs on
    the reordered problem, predictions map back through ``perm``."""
    n = perm.numel()
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, dtype=perm.dtype, device=perm.device)
    data.adj_t = data.adj_t.permute(perm.to(data.adj_t.device))
    for name in ("x", "y", "teacher_out_feat", "teacher_logits", "community"):
        t = getattr(data, name, None)
        if t is not None:
            setattr(data, name, t[perm.to(t.device)])
    if getattr(data, "split_idx", None) is not None:
        data.split_idx = {k: inv.to(v.device)[v] for k, v in data.split_idx.items()}
    data.perm = perm
    return data


class ToSparseTensor:
    """``reorder='community'`` (extension, off by default like PyG's): after building ``adj_t``, relabel the nodes in a
    locality-aware order (``sparse.community_order``) and permute the per-node tensors of ``data`` with it."""

    def __init__(self, reorder: str | None = None):
        if reorder not in (None, "community"):
            raise ValueError(f"unknown reorder '{reorder}'")
        self.reorder = reorder

    def __call__(self, data):
        n = getattr(data, "num_nodes", None)
        if n is None:
            n = data.x.shape[0]
        data.adj_t = to_sparse_tensor(data.edge_index, int(n))
        data.edge_index = None
        if self.reorder == "community":
            from .sparse import community_order
            reorder_nodes(data, community_order(data.adj_t.to_symmetric()))
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
