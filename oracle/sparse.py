"""ORACLE (test infrastructure): torch-sparse ``SparseTensor`` semantics used by the reference.

Restates, in plain CPU PyTorch, the subset of torch-sparse 0.6.8/0.6.9 the reference's hot path
touches:

  ``T.ToSparseTensor()``               /root/reference/arxiv_pyg/gnn.py:236-237
  ``adj_t.to_symmetric()``             /root/reference/arxiv_pyg/gnn.py:240
  ``adj_t.coo()``                      /root/reference/arxiv_pyg/gnn.py:248
  ``SparseTensor(row=col, col=row)``   /root/reference/mag_pyg/gnn.py:151
  ``adj_t.matmul(x, reduce='mean')``   /root/reference/mag_pyg/gnn.py:162
  ``gcn_norm`` (inside ``GCNConv``)    /root/reference/arxiv_pyg/gnn.py:28-35 (first forward, cached)

Index tensors are int64, values fp32.  Nothing here is on any product path.
"""
from __future__ import annotations

import torch
from torch import Tensor


def ind2ptr(row: Tensor, n_rows: int) -> Tensor:
    """rowptr[i] = #entries with row < i  (torch-sparse ``ind2ptr``)."""
    counts = torch.bincount(row, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr


def ptr2ind(rowptr: Tensor, nnz: int) -> Tensor:
    n_rows = rowptr.numel() - 1
    return torch.repeat_interleave(torch.arange(n_rows, dtype=torch.int64, device=rowptr.device),
                                   rowptr[1:] - rowptr[:-1], output_size=nnz)


class _Storage:
    """Mirror of ``SparseTensor.storage`` accessors used by callers."""

    def __init__(self, owner: "SparseTensor"):
        self._o = owner

    def rowptr(self):
        return self._o._rowptr

    def row(self):
        return self._o._row()

    def col(self):
        return self._o._col

    def value(self):
        return self._o._value

    def csr2csc(self):
        return self._o._csr2csc()

    def colptr(self):
        return self._o._colptr()

    def rowcount(self):
        return self._o._rowptr[1:] - self._o._rowptr[:-1]


class SparseTensor:
    """CSR sparse matrix, rows sorted, columns ascending within a row, duplicates kept."""

    def __init__(self, row: Tensor | None = None, rowptr: Tensor | None = None, col: Tensor | None = None,
                 value: Tensor | None = None, sparse_sizes=None, is_sorted: bool = False):
        assert col is not None and (row is not None or rowptr is not None)
        if sparse_sizes is None:
            m = int(row.max()) + 1 if row is not None and row.numel() else (rowptr.numel() - 1 if rowptr is not None else 0)
            n = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (m, n)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if row is not None and not is_sorted:
            # torch-sparse: argsort of row * N + col (keys of duplicates are equal)
            key = row * self._sizes[1] + col
            perm = torch.argsort(key, stable=True)
            row, col = row[perm], col[perm]
            if value is not None:
                value = value[perm]
        self._rowptr = rowptr if rowptr is not None else ind2ptr(row, self._sizes[0])
        self._row_cache = row
        self._col = col
        self._value = value
        self._t_cache = None  # (colptr, csr2csc)
        self.storage = _Storage(self)

    # ---- basic accessors -------------------------------------------------------------------
    def _row(self) -> Tensor:
        if self._row_cache is None:
            self._row_cache = ptr2ind(self._rowptr, self._col.numel())
        return self._row_cache

    def _transpose_meta(self):
        if self._t_cache is None:
            n_cols = self._sizes[1]
            key = self._col * self._sizes[0] + self._row()
            perm = torch.argsort(key, stable=True)  # csr2csc
            colptr = ind2ptr(self._col[perm], n_cols)
            self._t_cache = (colptr, perm)
        return self._t_cache

    def _csr2csc(self) -> Tensor:
        return self._transpose_meta()[1]

    def _colptr(self) -> Tensor:
        return self._transpose_meta()[0]

    def sparse_sizes(self):
        return self._sizes

    def sparse_size(self, dim: int) -> int:
        return self._sizes[dim]

    def size(self, dim: int) -> int:
        return self._sizes[dim]

    def nnz(self) -> int:
        return self._col.numel()

    def has_value(self) -> bool:
        return self._value is not None

    def coo(self):
        return self._row(), self._col, self._value

    def csr(self):
        return self._rowptr, self._col, self._value

    def set_value(self, value, layout=None) -> "SparseTensor":
        out = SparseTensor(rowptr=self._rowptr, col=self._col, value=value, sparse_sizes=self._sizes)
        out._row_cache = self._row_cache
        out._t_cache = self._t_cache
        return out

    def fill_value(self, fill: float, dtype=torch.float32) -> "SparseTensor":
        return self.set_value(torch.full((self.nnz(),), fill, dtype=dtype, device=self._col.device))

    def to(self, device) -> "SparseTensor":
        return self  # the oracle is CPU-only

    def t(self) -> "SparseTensor":
        colptr, perm = self._transpose_meta()
        val = None if self._value is None else self._value[perm]
        return SparseTensor(rowptr=colptr, col=self._row()[perm], value=val,
                            sparse_sizes=(self._sizes[1], self._sizes[0]))

    # ---- structure ops ---------------------------------------------------------------------
    def to_symmetric(self) -> "SparseTensor":
        """Set-union of (r,c) and (c,r), sorted by (r,c), duplicates merged."""
        assert self._value is None, "the reference only symmetrises value-less adjacency"
        n = max(self._sizes)
        row, col = self._row(), self._col
        key = torch.unique(torch.cat([row * n + col, col * n + row]))  # sorted + deduped
        return SparseTensor(row=key // n, col=key % n, sparse_sizes=(n, n), is_sorted=True)

    def fill_diag(self, fill: float) -> "SparseTensor":
        """Drop existing diagonal, insert (i,i,fill) for every i at its sorted slot."""
        m, n = self._sizes
        k = min(m, n)
        row, col = self._row(), self._col
        keep = row != col
        val = self._value if self._value is not None else None
        diag = torch.arange(k, dtype=torch.int64)
        new_row = torch.cat([row[keep], diag])
        new_col = torch.cat([col[keep], diag])
        new_val = None
        if val is not None:
            new_val = torch.cat([val[keep], torch.full((k,), fill, dtype=val.dtype)])
        return SparseTensor(row=new_row, col=new_col, value=new_val, sparse_sizes=(m, n))

    def sum(self, dim: int = 1) -> Tensor:
        assert dim == 1
        m = self._sizes[0]
        val = self._value if self._value is not None else torch.ones(self.nnz())
        return torch.zeros(m, dtype=val.dtype).index_add_(0, self._row(), val)

    # ---- matmul ----------------------------------------------------------------------------
    def matmul(self, x: Tensor, reduce: str = "sum") -> Tensor:
        return matmul(self, x, reduce)

    def __matmul__(self, x: Tensor) -> Tensor:
        return matmul(self, x, "sum")


# --------------------------------------------------------------------------------------------
# torch_sparse.matmul(A, X, reduce)  
# --------------------------------------------------------------------------------------------
def _csr_mm(rowptr: Tensor, col: Tensor, val: Tensor, x: Tensor, n_rows: int, n_cols: int) -> Tensor:
    """Row-parallel CSR x dense on the host cores (ATen sparse-CSR kernel)."""
    a = torch.sparse_csr_tensor(rowptr, col, val, size=(n_rows, n_cols))
    return torch.sparse.mm(a, x)


class _SpmmSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj: SparseTensor, mean: bool):
        rowptr, col, val = adj.csr()
        m, n = adj.sparse_sizes()
        cnt = (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(x.dtype)
        v = val if val is not None else torch.ones(col.numel(), dtype=x.dtype)
        out = _csr_mm(rowptr, col, v, x, m, n)
        if mean:
            out = out / cnt.unsqueeze(1)
        ctx.adj, ctx.mean, ctx.cnt = adj, mean, cnt
        return out

    @staticmethod
    def backward(ctx, gout):
        adj, mean = ctx.adj, ctx.mean
        rowptr, col, val = adj.csr()
        m, n = adj.sparse_sizes()
        colptr, perm = adj._transpose_meta()
        v = val if val is not None else torch.ones(col.numel(), dtype=gout.dtype)
        row = adj._row()
        if mean:  # val := val / max(rowcount[row], 1)
            v = v / ctx.cnt[row]
        gx = _csr_mm(colptr, row[perm], v[perm], gout.contiguous(), n, m)
        return gx, None, None


class _SpmmMax(torch.autograd.Function):
    """max with argmax = first maximal stored entry in CSR order; empty rows -> 0, argmax -1."""

    @staticmethod
    def forward(ctx, x, adj: SparseTensor):
        rowptr, col, val = adj.csr()
        m, _ = adj.sparse_sizes()
        row = adj._row()
        k = x.shape[1]
        src = x[col] if val is None else x[col] * val.unsqueeze(1)
        out = torch.full((m, k), float("-inf"), dtype=x.dtype)
        out.scatter_reduce_(0, row.unsqueeze(1).expand(-1, k), src, reduce="amax", include_self=True)
        e = torch.arange(col.numel(), dtype=torch.int64).unsqueeze(1).expand(-1, k)
        big = torch.full((m, k), col.numel(), dtype=torch.int64)
        cand = torch.where(src == out[row], e, torch.full_like(e, col.numel()))
        big.scatter_reduce_(0, row.unsqueeze(1).expand(-1, k), cand, reduce="amin", include_self=True)
        empty = (rowptr[1:] == rowptr[:-1]).unsqueeze(1)
        out = torch.where(empty, torch.zeros_like(out), out)
        arg = torch.where(empty | (big >= col.numel()), torch.full_like(big, -1), big)
        ctx.save_for_backward(arg)
        ctx.adj = adj
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, gout, _garg):
        (arg,) = ctx.saved_tensors
        adj = ctx.adj
        _, col, val = adj.csr()
        n = adj.sparse_sizes()[1]
        k = gout.shape[1]
        gx = torch.zeros(n, k, dtype=gout.dtype)
        if col.numel() == 0:   # no stored entry: nothing receives gradient
            return gx, None
        valid = arg >= 0
        e = arg.clamp(min=0)
        g = gout if val is None else gout * val[e]
        g = torch.where(valid, g, torch.zeros_like(g))
        gx.scatter_add_(0, col[e], g)
        return gx, None


def matmul(adj: SparseTensor, x: Tensor, reduce: str = "sum") -> Tensor:
    if reduce in ("sum", "add"):
        return _SpmmSumMean.apply(x, adj, False)
    if reduce == "mean":
        return _SpmmSumMean.apply(x, adj, True)
    if reduce == "max":
        return _SpmmMax.apply(x, adj)[0]
    raise ValueError(reduce)


def matmul_max_with_arg(adj: SparseTensor, x: Tensor):
    return _SpmmMax.apply(x, adj)


def spmm_loops(adj: SparseTensor, x: Tensor, reduce: str = "sum") -> Tensor:
    """Second, independent statement of SURVEY 9.6 (entry-order accumulation via index_add_)."""
    rowptr, col, val = adj.csr()
    row = adj._row()
    src = x[col] if val is None else x[col] * val.unsqueeze(1)
    out = torch.zeros(adj.sparse_sizes()[0], x.shape[1], dtype=x.dtype).index_add_(0, row, src)
    if reduce == "mean":
        out = out / (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(x.dtype).unsqueeze(1)
    return out


# --------------------------------------------------------------------------------------------
# T.ToSparseTensor   and gcn_norm
# --------------------------------------------------------------------------------------------
def to_sparse_tensor(edge_index: Tensor, num_nodes: int) -> SparseTensor:
    """(row, col) = edge_index = (source, target); adj_t rows are targets, columns the sources."""
    src, dst = edge_index[0], edge_index[1]
    perm = torch.argsort(dst * num_nodes + src, stable=True)
    return SparseTensor(row=dst[perm], col=src[perm], value=None,
                        sparse_sizes=(num_nodes, num_nodes), is_sorted=True)


class ToSparseTensor:
    """``data.edge_index`` -> ``data.adj_t``; removes ``edge_index`` (arxiv_pyg/gnn.py:237)."""

    def __call__(self, data):
        n = data.num_nodes if getattr(data, "num_nodes", None) is not None else data.x.shape[0]
        data.adj_t = to_sparse_tensor(data.edge_index, n)
        data.edge_index = None
        return data


def gcn_norm_sparse(adj_t: SparseTensor, add_self_loops: bool = True) -> SparseTensor:
    """A^ = D^-1/2 (A + I) D^-1/2 on a SparseTensor (SparseTensor branch)."""
    if not adj_t.has_value():
        adj_t = adj_t.fill_value(1.0)
    if add_self_loops:
        adj_t = adj_t.fill_diag(1.0)
    deg = adj_t.sum(dim=1)
    dinv = deg.pow(-0.5)
    dinv.masked_fill_(dinv == float("inf"), 0.0)
    row, col, val = adj_t.coo()
    val = (val * dinv[row]) * dinv[col]
    return adj_t.set_value(val)


def add_remaining_self_loops(edge_index: Tensor, edge_weight: Tensor, fill: float, n: int):
    """PyG <=1.7: drop loops from their slots, append all N loops (existing weights preserved)."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_w = torch.full((n,), fill, dtype=edge_weight.dtype)
    inv = ~mask
    if int(inv.sum()) > 0:
        loop_w[row[inv]] = edge_weight[inv]
    loops = torch.arange(n, dtype=torch.int64)
    ei = torch.cat([edge_index[:, mask], torch.stack([loops, loops])], dim=1)
    ew = torch.cat([edge_weight[mask], loop_w])
    return ei, ew


def gcn_norm_edge_index(edge_index: Tensor, num_nodes: int, dtype=torch.float32):
    """SURVEY 9.3 edge-index branch (PPI path, /root/reference/ppi_pyg/gnn.py:125-132,201)."""
    ew = torch.ones(edge_index.shape[1], dtype=dtype)
    edge_index, ew = add_remaining_self_loops(edge_index, ew, 1.0, num_nodes)
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(num_nodes, dtype=dtype).index_add_(0, col, ew)
    dinv = deg.pow(-0.5)
    dinv.masked_fill_(dinv == float("inf"), 0.0)
    return edge_index, dinv[row] * ew * dinv[col]


def neighbor_average_features(adj_t: "SparseTensor", x, R: int):
    """SIGN hop features (/root/reference/arxiv_dgl/sign.py:175-186): repeated mean over the in-neighbours."""
    feats = [x]
    adj = adj_t.set_value(None) if adj_t.has_value() else adj_t
    for _ in range(R):
        feats.append(matmul(adj, feats[-1], "mean"))
    return feats
