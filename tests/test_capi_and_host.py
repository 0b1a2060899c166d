"""CPU-side checks: the C-ABI library loads and exports every declared symbol; host-side integer logic
is bit-exact vs the oracle; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import efficient_gnns_amd as E
import efficient_gnns_amd.data as D
import oracle.sparse as OS
import oracle.utils as OU
from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "egnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(egnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(E._lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/egnn_hip.h but not exported"
    assert set(syms) == set(E._lib.SIGNATURES), "ctypes table and header disagree"
    assert E._lib.load().egnn_abi_version() == 1
    assert "gfx950" in E._lib.build_info()
    assert E._lib.load().egnn_error_string(-3) == b"workspace too small"


def test_algorithmic_bytes_formula():
    # K=256 GCN layer on ogbn-arxiv, int32 indices + fp32 values = 367.4 MB
    b = E._lib.load().egnn_spmm_algorithmic_bytes(169343, 169343, 256, 2484941, 32, 1)
    assert b == 4 * 169343 * 256 * 2 + 2484941 * 8 + 169344 * 4
    assert abs(b / 1e6 - 367.4) < 0.5


def test_no_cpu_fallback():
    adj = E.to_sparse_tensor(torch.tensor([[0, 1], [1, 0]]), 2)
    with pytest.raises(E._lib.HipExtensionError):
        adj.matmul(torch.randn(2, 4))
    with pytest.raises(E._lib.HipExtensionError):
        E.GCNConv(4, 4)(torch.randn(2, 4), adj)
    with pytest.raises(E._lib.HipExtensionError):
        E.kd_criterion(torch.randn(3, 4), torch.tensor([0, 1, 2]), torch.randn(3, 4))
    with pytest.raises(E._lib.HipExtensionError):
        E.nce_criterion(torch.randn(3, 4), torch.tensor([0, 1, 2]), torch.randn(3, 8), torch.randn(3, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "efficient-gnns_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


def random_edges(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])


@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (7, 0, 1), (5, 6, 2), (50, 400, 3), (300, 2000, 4)])
def test_structure_bit_exact_vs_oracle_on_host(n, e, seed):
    ei = random_edges(n, e, seed) if e else torch.zeros(2, 0, dtype=torch.int64)
    a, o = E.to_sparse_tensor(ei, n), OS.to_sparse_tensor(ei, n)
    for x, y in zip(a.csr()[:2], o.csr()[:2]):
        assert torch.equal(x, y)
    s, so = a.to_symmetric(), o.to_symmetric()
    for x, y in zip(s.csr()[:2], so.csr()[:2]):
        assert torch.equal(x, y)
    assert torch.equal(s.storage.colptr(), so._colptr()) and torch.equal(s.storage.csr2csc(), so._csr2csc())
    assert torch.equal(s.t().csr()[1], so.t().csr()[1])
    if e:
        subset = torch.randperm(n)[: max(1, n // 2)]
        eidx = torch.stack(s.coo()[:2])
        assert torch.equal(E.subgraph(subset, eidx, relabel_nodes=True, num_nodes=n)[0],
                           OU.subgraph(subset, eidx, relabel_nodes=True, num_nodes=n)[0])


def test_mag_style_constructor_sorts():
    row = torch.tensor([3, 0, 2, 0, 3])
    col = torch.tensor([1, 2, 2, 0, 0])
    a, o = E.SparseTensor(row=col, col=row), OS.SparseTensor(row=col, col=row)  # mag_pyg/gnn.py:151
    assert a.sparse_sizes() == o.sparse_sizes()
    for x, y in zip(a.csr()[:2], o.csr()[:2]):
        assert torch.equal(x, y)


def test_synthetic_arxiv_generator_properties():
    d = D.arxiv_like(scale=0.02, seed=3)
    n = d.num_nodes
    rowptr, col, _ = d.adj_t.csr()
    row = d.adj_t.storage.row()
    assert rowptr[-1] == col.numel() and (row != col).all()            # no self loops
    key = row * n + col
    assert torch.unique(key).numel() == key.numel()                     # no multi-edges
    assert torch.equal(torch.sort(col * n + row)[0], key)               # symmetric
    sizes = [d.split_idx[k].numel() for k in ("train", "valid", "test")]
    assert sum(sizes) == n and torch.unique(torch.cat(list(d.split_idx.values()))).numel() == n
    assert d.teacher_out_feat.min() >= 0 and d.teacher_out_feat.shape == (n, 750)
    d2 = D.arxiv_like(scale=0.02, seed=3)
    assert torch.equal(d2.adj_t.csr()[1], col) and torch.equal(d2.x, d.x)  # seeded


def test_powerlaw_edges_exact_count_and_hub():
    ei = D.powerlaw_edges(5000, 40000, max_degree=800, seed=1)
    assert ei.shape == (2, 40000) and (ei[0] != ei[1]).all()
    indeg = np.bincount(ei[1], minlength=5000)
    assert 400 < indeg.max() < 1600 and np.median(indeg) <= 8


def test_error_behaviour_matches_reference():
    """criterion.py:86,115,122 raise NotImplementedError for unknown kernels / criteria; gnn.py:260 ValueError."""
    x = torch.randn(4, 3)
    y = torch.tensor([0, 1, 2, 0])
    f = torch.randn(4, 8)
    with pytest.raises(NotImplementedError):
        E.gpw_criterion(x, y, f, f, kernel="laplace")
    with pytest.raises(NotImplementedError):
        E.lpw_criterion(x, y, f, f, torch.tensor([[0, 1], [1, 0]]), kernel="laplace")
    with pytest.raises(NotImplementedError):
        E.lpw_criterion(x, y, f, f, torch.tensor([[0, 1], [1, 0]]), kernel="rbf", criterion="huber")
    with pytest.raises(ValueError):
        E.SAGEConv(4, 4, aggr="median")
    with pytest.raises(ValueError):
        E.SparseTensor(col=torch.tensor([0]))


def test_conv_parameter_layouts_match_pyg_1_7():
    g = E.GCNConv(128, 256, cached=True)
    assert tuple(g.weight.shape) == (128, 256) and tuple(g.bias.shape) == (256,) and float(g.bias.abs().sum()) == 0.0
    assert sorted(g.state_dict()) == ["bias", "weight"]
    s = E.SAGEConv(128, 256)
    assert sorted(s.state_dict()) == ["lin_l.bias", "lin_l.weight", "lin_r.weight"]
    import efficient_gnns_amd.models as PM
    n = lambda m: sum(p.numel() for p in m.parameters())  # noqa: E731
    assert n(PM.GCN(128, 256, 40, 2, 0.5)) == 43816 and n(PM.SAGE(128, 256, 40, 2, 0.5)) == 86824
    assert n(PM.GCN(128, 256, 40, 3, 0.5)) == 110120
    assert sorted(PM.make_projection(256, 128).state_dict())[:2] == ["0.bias", "0.weight"]
