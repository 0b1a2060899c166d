"""Independent second opinion on the [ext] operator layer of the oracle.

The golden generator shims the reference's third-party operators (torch-sparse ``SparseTensor``, PyG ``gcn_norm`` /
``softmax`` / ``subgraph``) with ``oracle.*`` because those packages cannot be installed here, so the goldens cannot
catch a systematic error in that layer.  This file restates the same semantics a SECOND time with ``scipy.sparse`` and
NumPy only -- no ``torch.sparse``, no oracle helper on the checking side -- straight from the written spec (SURVEY.md
section 9), and compares the oracle with it at the size of the headline workload (N = 169 343, 1 166 243 directed
edges, the synthetic graph ``bench.py`` runs).  Integer arrays bit-equal; values rtol 1e-6 against float64 (sums over
long rows / groups: the fp32 summation bar of SURVEY 8c, 1e-5 of the largest value).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import efficient_gnns_amd.data as D
import oracle.sparse as OS
import oracle.utils as OU


@pytest.fixture(scope="module")
def arxiv_edges():
    n = D.ARXIV["num_nodes"]
    ei = D.powerlaw_edges(n, D.ARXIV["num_edges"], max_degree=D.ARXIV["max_degree"], seed=0)
    return n, ei


def _csr_of_targets(ei, n):
    """9.1: row i lists the sources j of the edges j -> i, ascending, duplicates kept."""
    src, dst = ei
    order = np.lexsort((src, dst))                       # by target, then source
    rows, cols = dst[order], src[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    return rowptr, cols.astype(np.int64)


def _symmetric(ei, n):
    """9.2: set-union of (r,c) and (c,r), sorted by (r,c), duplicates merged -- through a scipy boolean pattern."""
    src, dst = ei
    a = sp.coo_matrix((np.ones(src.size, dtype=np.int8), (dst, src)), shape=(n, n)).tocsr()
    s = ((a + a.T) > 0).tocsr()
    s.sort_indices()
    return s


def test_to_sparse_tensor_and_to_symmetric_bit_exact_at_arxiv_scale(arxiv_edges):
    n, ei = arxiv_edges
    o = OS.to_sparse_tensor(torch.from_numpy(ei), n)
    rowptr, col, val = o.csr()
    rp_ref, col_ref = _csr_of_targets(ei, n)
    assert val is None
    assert np.array_equal(rowptr.numpy(), rp_ref) and np.array_equal(col.numpy(), col_ref)
    s_ref = _symmetric(ei, n)
    rowptr_s, col_s, _ = o.to_symmetric().csr()
    assert np.array_equal(rowptr_s.numpy(), s_ref.indptr.astype(np.int64))
    assert np.array_equal(col_s.numpy(), s_ref.indices.astype(np.int64))
    # the cached transpose meta (csr2csc / colptr) against scipy's own CSR -> CSC conversion of an entry-numbered matrix
    st = o.to_symmetric()
    ids = sp.csr_matrix((np.arange(1, s_ref.nnz + 1, dtype=np.int64), s_ref.indices, s_ref.indptr), shape=(n, n)).tocsc()
    ids.sort_indices()
    assert np.array_equal(st._colptr().numpy(), ids.indptr.astype(np.int64))
    assert np.array_equal(st._csr2csc().numpy(), ids.data - 1)


def _gcn_norm_scipy(s):
    """9.3 in float64: value := 1, diagonal replaced by 1, deg = row sums, dinv = deg^-1/2 (inf -> 0)."""
    n = s.shape[0]
    a = s.astype(np.float64).tolil()
    a.setdiag(1.0)
    a = a.tocsr()
    a.sort_indices()
    deg = np.asarray(a.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = deg ** -0.5
    dinv[~np.isfinite(dinv)] = 0.0
    return (sp.diags(dinv) @ a @ sp.diags(dinv)).tocsr(), n


def test_gcn_norm_structure_bit_exact_and_values_at_arxiv_scale(arxiv_edges):
    n, ei = arxiv_edges
    s_ref = _symmetric(ei, n)
    # existing self loops must be REPLACED (weight 1, once): plant a few before normalising, on both sides
    loops = np.array([0, 17, n - 1])
    with_loops = (s_ref + sp.coo_matrix((np.ones(3, dtype=bool), (loops, loops)), shape=(n, n)).tocsr() > 0).tocsr()
    with_loops.sort_indices()
    ref, _ = _gcn_norm_scipy(with_loops)
    ref.sort_indices()
    o = OS.SparseTensor(rowptr=torch.from_numpy(with_loops.indptr.astype(np.int64)),
                        col=torch.from_numpy(with_loops.indices.astype(np.int64)), sparse_sizes=(n, n))
    rowptr, col, val = OS.gcn_norm_sparse(o).csr()
    assert np.array_equal(rowptr.numpy(), ref.indptr.astype(np.int64))
    assert np.array_equal(col.numpy(), ref.indices.astype(np.int64))
    np.testing.assert_allclose(val.numpy().astype(np.float64), ref.data, rtol=1e-6)
    assert val.dtype == torch.float32


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_spmm_sum_and_mean_forward_backward_at_arxiv_scale(arxiv_edges, reduce):
    """9.6: Y = A X (sum) or (A X) / max(count, 1) (mean, count = stored entries); dX = A^T dY with the same scaling."""
    n, ei = arxiv_edges
    s_ref = _symmetric(ei, n)
    K = 32
    g = np.random.default_rng(3)
    x = g.standard_normal((n, K)).astype(np.float32)
    gy = g.standard_normal((n, K)).astype(np.float32)
    if reduce == "sum":      # GCN-normalised values
        a, _ = _gcn_norm_scipy(s_ref)
        a.sort_indices()
        o = OS.SparseTensor(rowptr=torch.from_numpy(a.indptr.astype(np.int64)), col=torch.from_numpy(a.indices.astype(np.int64)),
                            value=torch.from_numpy(a.data.astype(np.float32)), sparse_sizes=(n, n))
        a = sp.csr_matrix((a.data.astype(np.float32).astype(np.float64), a.indices, a.indptr), shape=(n, n))   # the same fp32 values
        scale = np.ones(n)
    else:                    # value-less adjacency, mean over the stored entries; empty rows give 0
        a = s_ref.astype(np.float64)
        o = OS.SparseTensor(rowptr=torch.from_numpy(s_ref.indptr.astype(np.int64)), col=torch.from_numpy(s_ref.indices.astype(np.int64)),
                            sparse_sizes=(n, n))
        scale = 1.0 / np.maximum(np.diff(s_ref.indptr), 1)
    y_ref = (a @ x.astype(np.float64)) * scale[:, None]
    dx_ref = a.T @ (gy.astype(np.float64) * scale[:, None])
    xt = torch.from_numpy(x).requires_grad_(True)
    y = OS.matmul(o, xt, reduce)
    y.backward(torch.from_numpy(gy))
    tol = dict(rtol=1e-5, atol=1e-5 * np.abs(y_ref).max())
    np.testing.assert_allclose(y.detach().numpy().astype(np.float64), y_ref, **tol)
    np.testing.assert_allclose(xt.grad.numpy().astype(np.float64), dx_ref, rtol=1e-5, atol=1e-5 * np.abs(dx_ref).max())
    assert np.diff(s_ref.indptr).min() == 0 or reduce == "sum"   # the synthetic graph has isolated nodes: the mean's max(count, 1) is exercised


def test_segment_softmax_and_subgraph_at_arxiv_scale(arxiv_edges):
    """9.7 (softmax over groups given by an unsorted index, +1e-16 in the denominator) and 9.8 (train-induced subgraph
    with relabelling by position in the subset), on the train subgraph of the headline workload."""
    n, ei = arxiv_edges
    s_ref = _symmetric(ei, n).tocoo()
    row, col = s_ref.row.astype(np.int64), s_ref.col.astype(np.int64)
    g = np.random.default_rng(4)
    subset = g.permutation(n)[: D.ARXIV["split"][0]]
    # 9.8 with NumPy: keep the edges with both ends in the subset, in the original order; relabel id -> position
    pos = np.full(n, -1, dtype=np.int64)
    pos[subset] = np.arange(subset.size)
    keep = (pos[row] >= 0) & (pos[col] >= 0)
    ref_ei = np.stack([pos[row[keep]], pos[col[keep]]])
    out_ei, attr = OU.subgraph(torch.from_numpy(subset), torch.from_numpy(np.stack([row, col])), relabel_nodes=True, num_nodes=n)
    assert attr is None and np.array_equal(out_ei.numpy(), ref_ei)
    # 9.7 with NumPy ufunc.at in float64
    index = ref_ei[1]
    src = (g.standard_normal(index.size) * 4).astype(np.float32)
    m = np.full(subset.size, -np.inf)
    np.maximum.at(m, index, src.astype(np.float64))
    e = np.exp(src.astype(np.float64) - m[index])
    ssum = np.zeros(subset.size)
    np.add.at(ssum, index, e)
    ref = e / (ssum[index] + 1e-16)
    out = OU.softmax(torch.from_numpy(src), torch.from_numpy(index), subset.size)
    # fp32 against float64: the largest groups (hub nodes) add up thousands of fp32 terms one after the other
    np.testing.assert_allclose(out.numpy().astype(np.float64), ref, rtol=2e-5, atol=1e-12)
    small = np.bincount(index, minlength=subset.size)[index] <= 32     # short groups: only the rounding of exp and one division
    np.testing.assert_allclose(out.numpy().astype(np.float64)[small], ref[small], rtol=2e-6, atol=1e-12)
