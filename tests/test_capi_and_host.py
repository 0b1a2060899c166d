"""CPU-side checks: the C-ABI library loads and exports every declared symbol; host-side integer logic
is bit-exact vs the oracle; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import efficient_gnns_amd as E
import efficient_gnns_amd.data as D
from efficient_gnns_amd import _lib
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
    assert E._lib.load().egnn_abi_version() == 6
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


def test_c_abi_argument_errors_are_reported_before_any_launch():
    """Every entry point validates its arguments on the host and returns a negative EGNN_E* code without enqueueing
    anything (include/egnn_hip.h conventions) -- callable without a GPU."""
    lib = _lib.load()
    EINVAL, EWORKSPACE = -1, None
    import ctypes
    buf = (ctypes.c_float * 64)()
    ibuf = (ctypes.c_int64 * 16)()
    p, ip = ctypes.addressof(buf), ctypes.addressof(ibuf)
    # segment SpMM: max is not offered; missing combine arrays; negative sizes
    assert lib.egnn_spmm_csr_seg_f32(4, 4, 4, ip, ip, 64, None, None, None, p, 4, p, 4, 2, ip, 1, None, None, 0, None, 0, None) < 0
    assert lib.egnn_spmm_csr_seg_f32(4, 4, 4, ip, ip, 64, None, None, None, p, 4, p, 4, 0, ip, 1, None, None, 2, None, 0, None) < 0
    assert lib.egnn_spmm_csr_seg_f32(4, 4, 4, ip, ip, 16, None, None, None, p, 4, p, 4, 0, ip, 1, None, None, 0, None, 0, None) < 0
    assert lib.egnn_spmm_csr_seg_f32(0, 0, 4, None, None, 64, None, None, None, None, 4, None, 4, 0, None, 0, None, None, 0, None, 0, None) == 0
    # unaligned leading dimension -> EGNN_EALIGN (-4): the host falls back to egnn_spmm_csr_f32
    assert lib.egnn_spmm_csr_seg_f32(4, 4, 4, ip, ip, 64, None, None, None, p, 5, p, 5, 0, ip, 1, None, None, 0, None, 0, None) == -4
    # fused-gather GEMM: at most one gather, and only on an untransposed operand
    assert lib.egnn_gemm_rows_f32(0, 1, 4, 4, 4, 1.0, p, 4, ip, p, 4, ip, None, p, 4, 1, None, 0, None) < 0
    assert lib.egnn_gemm_rows_f32(1, 1, 4, 4, 4, 1.0, p, 4, ip, p, 4, None, None, p, 4, 1, None, 0, None) < 0
    assert lib.egnn_gemm_rows_f32(0, 0, 4, 4, 4, 1.0, p, 4, None, p, 4, ip, None, p, 2, 1, None, 0, None) < 0       # ldc < N
    assert lib.egnn_gemm_f32(0, 0, 4, 4, 64, 1.0, p, 64, p, 4, None, p, 4, 2, None, 0, None) < 0                      # split-K without a workspace
    # G-CRD: tau must be positive; the backward workspace, when given, must be large enough
    assert lib.egnn_nce_fwd_f32(p, p, 4, 4, 4, 0.0, 1, p, p, p, p, 1 << 20, None) < 0
    assert lib.egnn_nce_bwd_f32(p, p, 4, 4, 4, 0.1, 1, p, p, None, p, p, p, 1, None) < 0
    assert lib.egnn_nce_bwd_ws_floats(0, 4, 4) == 0 and lib.egnn_nce_bwd_ws_floats(128, 128, 16) >= 128 * 16
    assert lib.egnn_nce_saves_exp(0.075, 1) == 1 and lib.egnn_nce_saves_exp(0.075, 0) == 0 and lib.egnn_nce_saves_exp(0.01, 1) == 0
    # BatchNorm halves: null pointers
    assert lib.egnn_bn_act_bwd_reduce_f32(None, 4, p, 4, 4, 4, p, p, 1e-5, p, p, 1, 0.0, 0, None, p, p, p, 1 << 20, None) < 0
    assert lib.egnn_bn_act_bwd_apply_f32(p, 4, p, 4, 4, 4, p, p, 1e-5, p, p, 1, 0.0, 0, None, None, p, 1.0, p, 4, None) < 0


def test_segment_plan_covers_every_entry_exactly_once():
    """SparseTensor._seg_plan (integer preprocessing behind egnn_spmm_csr_seg_f32) on the host: every stored entry is in
    exactly one range, ranges hold at most SEG_MAX entries, single-range rows write Y directly, the others own
    consecutive partial slots in row order."""
    import efficient_gnns_amd.sparse as SP
    d = D.arxiv_like(scale=0.05, seed=11, with_teacher=False)
    adj = d.adj_t
    rowptr, col, _ = adj.csr()
    n = adj.sparse_size(0)
    seg, crow, cptr, slots = adj._seg_plan()
    cnt = rowptr[1:] - rowptr[:-1]
    assert int((cnt > SP.SEG_MAX).sum()) == crow.numel() > 0
    length = seg[:, 1] - seg[:, 0]
    assert int(length.max()) <= SP.SEG_MAX and int(length.min()) >= 0
    covered = torch.zeros(int(rowptr[-1]) + 1, dtype=torch.int64)
    covered.index_add_(0, seg[:, 0], torch.ones(seg.shape[0], dtype=torch.int64))
    covered.index_add_(0, seg[:, 1], -torch.ones(seg.shape[0], dtype=torch.int64))
    assert torch.equal(torch.cumsum(covered, 0)[:-1], torch.ones(int(rowptr[-1]), dtype=torch.int64))
    direct = seg[seg[:, 2] < n]
    assert torch.equal(torch.sort(direct[:, 2]).values, torch.nonzero(cnt <= SP.SEG_MAX).view(-1))
    assert torch.equal(direct[:, 0], rowptr[direct[:, 2]]) and torch.equal(direct[:, 1], rowptr[direct[:, 2] + 1])
    part = seg[seg[:, 2] >= n]
    assert part.shape[0] == slots == int(cptr[-1]) and torch.equal(part[:, 2], n + torch.arange(slots))
    for i in (0, crow.numel() // 2, crow.numel() - 1):     # slots of a row tile its entry range, in order
        r = int(crow[i])
        ps = part[int(cptr[i]):int(cptr[i + 1])]
        assert int(ps[0, 0]) == int(rowptr[r]) and int(ps[-1, 1]) == int(rowptr[r + 1])
        assert torch.equal(ps[1:, 0], ps[:-1, 1])


# ------------------------------------------------------------------------------------------------
# property-based: host structure logic vs the oracle on arbitrary edge lists
# ------------------------------------------------------------------------------------------------
from hypothesis import given, settings, strategies as st  # noqa: E402


@st.composite
def _edge_lists(draw):
    n = draw(st.integers(1, 40))
    e = draw(st.integers(0, 120))
    src = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    dst = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    if e and draw(st.booleans()):                      # a hub target and duplicate edges
        dst = [dst[0]] * (e // 2) + dst[e // 2:]
        src = src[:e // 2] + src[:e - e // 2]
    return n, torch.tensor([src, dst], dtype=torch.int64).reshape(2, e)


@settings(max_examples=40, deadline=None, derandomize=True)
@given(_edge_lists())
def test_host_structure_logic_property(case):
    """ToSparseTensor / to_symmetric / csr2csc / segment plan on arbitrary edge lists: empty graphs, isolated nodes, self
    loops, duplicates, hubs -- bit-exact against the oracle; the segment plan always tiles the entry range."""
    import efficient_gnns_amd.sparse as SP
    n, ei = case
    o, p = OS.to_sparse_tensor(ei, n), E.to_sparse_tensor(ei, n)
    for a, b in zip(p.csr()[:2], o.csr()[:2]):
        assert torch.equal(a, b)
    so, sp = o.to_symmetric(), p.to_symmetric()
    for a, b in zip(sp.csr()[:2], so.csr()[:2]):
        assert torch.equal(a, b)
    assert torch.equal(sp.storage.colptr(), so._colptr()) and torch.equal(sp.storage.csr2csc(), so._csr2csc())
    seg, crow, cptr, slots = p._seg_plan()
    rowptr = p.csr()[0]
    assert int((seg[:, 1] - seg[:, 0]).sum()) == p.nnz() and seg.shape[0] == n - crow.numel() + slots
    assert max((seg[:, 1] - seg[:, 0]).tolist(), default=0) <= SP.SEG_MAX
    assert sorted(seg[seg[:, 2] < n][:, 2].tolist() + crow.tolist()) == list(range(n))
    for i, r in enumerate(crow.tolist()):
        ps = seg[(seg[:, 2] >= n + int(cptr[i])) & (seg[:, 2] < n + int(cptr[i + 1]))]
        assert int(ps[:, 0].min()) == int(rowptr[r]) and int(ps[:, 1].max()) == int(rowptr[r + 1])


@settings(max_examples=40, deadline=None, derandomize=True)
@given(_edge_lists(), st.integers(0, 2 ** 16), st.booleans())
def test_subgraph_property(case, seed, as_mask):
    """utils.subgraph (gnn.py:246-249: the train-induced subgraph handed to the LSP loss): kept edges, their order and the
    relabelling equal the oracle's for index and boolean-mask subsets, with and without relabelling."""
    import efficient_gnns_amd.utils as PU
    n, ei = case
    g = torch.Generator().manual_seed(seed)
    k = int(torch.randint(0, n + 1, (1,), generator=g))
    subset = torch.randperm(n, generator=g)[:k]
    if as_mask:
        m = torch.zeros(n, dtype=torch.bool)
        m[subset] = True
        subset = m
    for relabel in (False, True):
        a, _ = PU.subgraph(subset, ei, relabel_nodes=relabel, num_nodes=n)
        b, _ = OU.subgraph(subset, ei, relabel_nodes=relabel, num_nodes=n)
        assert torch.equal(a, b)


def test_integration_md_ctypes_snippet_matches_the_header():
    """The binding shown in INTEGRATION.md must have exactly the parameters include/egnn_hip.h declares (doc rot guard)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    block = md[md.index("```python\nimport ctypes, torch"):]
    block = block[len("```python\n"):block.index("```", 10)]
    stmt = block[block.index("lib.egnn_spmm_csr_f32.argtypes"):block.index("def spmm_sum")]

    class _Fn:
        pass

    class _Lib:
        egnn_spmm_csr_f32 = _Fn()
    exec(stmt, {"ctypes": ctypes, "lib": _Lib})
    shown = _Lib.egnn_spmm_csr_f32.argtypes
    table = _lib.SIGNATURES["egnn_spmm_csr_f32"][1]
    assert len(shown) == len(table)
    assert [ctypes.sizeof(a) for a in shown] == [ctypes.sizeof(a) for a in table]
    call = block[block.index("rc = lib.egnn_spmm_csr_f32("):block.index("assert rc == 0")]
    assert call.count(",") + 1 == len(table), "the example call passes a different number of arguments"


def test_teacher_artifact_round_trip(tmp_path):
    """The on-disk hand-over between teacher and student runs (arxiv_dgl/gat.py:245-251 -> arxiv_pyg/gnn.py:278-279):
    plain torch.save tensors under features/<expt>/<seed>.pt and logits/<expt>/<seed>.pt."""
    g = torch.Generator().manual_seed(0)
    feat, logits = torch.relu(torch.randn(37, 750, generator=g)), torch.randn(37, 40, generator=g)
    D.save_teacher_artifacts(str(tmp_path), "gat-3L250x3h", 3, feat, logits)
    f_path, l_path = D.teacher_artifact_paths(str(tmp_path), "gat-3L250x3h", 3)
    assert f_path.endswith(os.path.join("features", "gat-3L250x3h", "3.pt")) and os.path.exists(l_path)
    assert torch.equal(torch.load(f_path), feat)                      # exactly what gnn.py:278 would read
    f2, l2 = D.load_teacher_artifacts(str(tmp_path), "gat-3L250x3h", 3, num_nodes=37)
    assert torch.equal(f2, feat) and torch.equal(l2, logits)
    with pytest.raises(ValueError):
        D.load_teacher_artifacts(str(tmp_path), "gat-3L250x3h", 3, num_nodes=38)


def test_dropin_launcher_path_order(tmp_path):
    """dropin/launch.py: a script directory's own criterion.py must NOT shadow the drop-in one (sys.path[0] is the script
    directory under ``python gnn.py``); a multi-label (BCE) script directory gets the ppi_pyg shim instead of the arxiv one;
    --keep-criterion keeps the script's own file."""
    import subprocess
    import sys as _sys
    launch = os.path.join(ROOT, "efficient-gnns_amd", "dropin", "launch.py")
    for flavour, body in (("arxiv_like", "import torch.nn.functional as F\ndef kd_criterion(*a): return F.cross_entropy\n"),
                          ("ppi_like", "import torch.nn.functional as F\ndef kd_criterion(*a): return F.binary_cross_entropy_with_logits\n")):
        sdir = tmp_path / flavour
        sdir.mkdir()
        (sdir / "criterion.py").write_text(body)
        (sdir / "logger.py").write_text("NAME = 'script-local module'\n")
        (sdir / "gnn.py").write_text(
            "import sys, criterion, logger\nfrom criterion import *\nimport inspect\n"
            "print('CRIT', criterion.__file__)\nprint('KD', inspect.signature(kd_criterion))\nprint('LOG', logger.NAME, sys.argv[1:])\n")
        out = subprocess.run([_sys.executable, launch, str(sdir / "gnn.py"), "--gnn", "gcn"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        crit = [l for l in out.stdout.splitlines() if l.startswith("CRIT")][0]
        kd = [l for l in out.stdout.splitlines() if l.startswith("KD")][0]
        assert "LOG script-local module ['--gnn', 'gcn']" in out.stdout
        if flavour == "arxiv_like":
            assert crit.endswith(os.path.join("dropin", "criterion.py")) and "alpha=0.9, T=4" in kd
        else:
            assert crit.endswith(os.path.join("dropin", "ppi_pyg", "criterion.py")) and "alpha=0.5, T=1" in kd
        keep = subprocess.run([_sys.executable, launch, "--keep-criterion", str(sdir / "gnn.py")], capture_output=True, text=True, timeout=300)
        assert keep.returncode == 0, keep.stderr[-2000:]
        assert str(sdir / "criterion.py") in keep.stdout


def test_community_order_and_permute_host_logic():
    """sparse.community_order / SparseTensor.permute / transforms.reorder_nodes (integer host logic, any device): a valid
    permutation, the permuted matrix is the same graph (P A P^T), the problem's tensors and index sets move with it, and on
    the community graph with shuffled ids the order recovers most of the true communities' locality."""
    from efficient_gnns_amd.sparse import community_order
    from efficient_gnns_amd.transforms import reorder_nodes
    d = D.arxiv_like(scale=0.05, seed=5, with_teacher=True, graph="local")
    n = d.num_nodes
    adj0, x0, y0, tr0 = d.adj_t, d.x.clone(), d.y.clone(), d.split_idx["train"].clone()
    perm = community_order(adj0)
    assert torch.equal(torch.sort(perm).values, torch.arange(n))
    dense0 = torch.zeros(n, n)
    dense0[adj0.storage.row(), adj0.storage.col()] = 1

    def near(adj, w):
        rowptr, col, _ = adj.csr()
        row = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
        return float(((row - col).abs() < w).float().mean())
    before = near(adj0, n // 40)
    reorder_nodes(d, perm)
    dense1 = torch.zeros(n, n)
    dense1[d.adj_t.storage.row(), d.adj_t.storage.col()] = 1
    assert torch.equal(dense1, dense0[perm][:, perm]), "P A P^T"
    rowptr, col, _ = d.adj_t.csr()
    assert all(bool((col[int(rowptr[i]):int(rowptr[i + 1])][1:] > col[int(rowptr[i]):int(rowptr[i + 1])][:-1]).all()) for i in range(0, n, 97))
    assert torch.equal(d.x, x0[perm]) and torch.equal(d.y, y0[perm])
    assert torch.equal(perm[d.split_idx["train"]], tr0), "index sets keep pointing at the same nodes"
    after = near(d.adj_t, n // 40)
    assert after > 5 * before and after > 0.3, (before, after)


def test_traffic_files_are_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.measured_traffic reports a PMC file only when it carries the stamp of the kernel sources in the tree (build.source_stamp:
    compile flags + csrc + header; independent of the binary's embedded build time), and the committed files carry it."""
    import importlib.util
    import json
    import shutil
    import bench
    spec = importlib.util.spec_from_file_location("egnn_build_t", os.path.join(ROOT, "efficient-gnns_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    stamp = b.source_stamp()
    assert re.fullmatch(r"[0-9a-f]{16}", stamp) and stamp == bench.lib_sha16() == b.source_stamp()
    for name in ("spmm_traffic.json", "spmm_traffic_local.json"):
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            j = json.load(fh)
        val, why = bench.measured_traffic(name)
        if j["lib_sha16"] == stamp:
            assert val == float(j["hbm_bytes_per_call"]) and stamp in why
        else:   # kernels edited since the last evidence run: the line will say traffic = null until tools/evidence.sh <tag> traffic is re-run
            import warnings
            warnings.warn(f"profiles/{name} was measured on other kernel sources ({j['lib_sha16']} != {stamp}): bench.py reports traffic = null")
            assert val is None and "stale" in why
    # a file from other sources is refused; one changed source byte changes the stamp
    fake_root = tmp_path / "repo"
    os.makedirs(fake_root / "profiles")
    j["lib_sha16"] = "0" * 16
    with open(fake_root / "profiles" / "spmm_traffic.json", "w") as fh:
        json.dump(j, fh)
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    monkeypatch.setattr(bench, "lib_sha16", lambda: stamp)
    val, why = bench.measured_traffic("spmm_traffic.json")
    assert val is None and "stale" in why
    csrc2 = tmp_path / "pkg" / "csrc"
    shutil.copytree(os.path.join(ROOT, "efficient-gnns_amd", "csrc"), csrc2)
    os.makedirs(tmp_path / "include")
    shutil.copy(os.path.join(ROOT, "include", "egnn_hip.h"), tmp_path / "include" / "egnn_hip.h")
    monkeypatch.setattr(b, "CSRC", str(csrc2))
    monkeypatch.setattr(b, "HERE", str(tmp_path / "pkg"))
    assert b.source_stamp() == stamp
    with open(csrc2 / "common.h", "a") as f:
        f.write("\n// one more byte\n")
    assert b.source_stamp() != stamp


def test_index_helpers_of_the_fused_paths_on_host():
    """Pure index logic behind two fused paths: ops._inverse_rows (row id -> position in the unique train index, -1 elsewhere; what the
    fused last-layer backward uses to find the projection head's row) and models._global_edges (the relabelled edge list of the
    train-induced subgraph, gnn.py:274, composed with train_idx); both cached on the identity AND version of their inputs."""
    import efficient_gnns_amd.models as PM
    import efficient_gnns_amd.ops as ops
    from efficient_gnns_amd.utils import subgraph
    g = torch.Generator().manual_seed(0)
    n = 500
    idx = torch.randperm(n, generator=g)[:200]
    inv = ops._inverse_rows(idx, n)
    assert inv.dtype == torch.int32 and inv.shape == (n,)
    assert torch.equal(inv[idx].long(), torch.arange(200)) and int((inv >= 0).sum()) == 200 and int(inv.min()) == -1
    assert ops._inverse_rows(idx, n) is inv                      # cached
    idx[0], idx[1] = idx[1].clone(), idx[0].clone()             # in-place edit bumps the version: a new inverse
    inv2 = ops._inverse_rows(idx, n)
    assert inv2 is not inv and torch.equal(inv2[idx].long(), torch.arange(200))
    # edges of the induced subgraph, relabelled -> back in full-graph ids they are exactly the edges with both ends in the index set
    ei = torch.randint(0, n, (2, 4000), generator=g)
    sub, _ = subgraph(idx, ei, relabel_nodes=True, num_nodes=n)
    glob = PM._global_edges(sub, idx)
    member = torch.zeros(n, dtype=torch.bool)
    member[idx] = True
    keep = member[ei[0]] & member[ei[1]]
    assert glob.shape == sub.shape and torch.equal(glob, ei[:, keep])
    assert PM._global_edges(sub, idx) is glob


def test_ctypes_table_matches_the_header_argument_by_argument():
    """Every prototype of include/egnn_hip.h parsed and compared with efficient-gnns_amd/_lib.py::SIGNATURES: same number of arguments,
    same kind per argument (pointer / int64 / int / float / size_t / uint64) and same return type -- a mismatch would not fail at
    load time, it would hand the kernels shifted arguments."""
    import ctypes as C
    import re
    from efficient_gnns_amd import _lib
    src = open(os.path.join(ROOT, "include", "egnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = re.findall(r"\b(int64_t|size_t|int|const\s+char\s*\*)\s+(egnn_\w+)\s*\(([^;{]*)\)\s*;", src)
    assert len(protos) >= 80
    kinds = {"p": C.c_void_p, "i64": C.c_int64, "i32": C.c_int, "f32": C.c_float, "sz": C.c_size_t, "u64": C.c_uint64, "str": C.c_char_p}

    def kind(decl):
        d = decl.strip()
        if d in ("void", ""):
            return None
        if "*" in d:
            return "str" if re.match(r"(const\s+)?char\s*\*", d) else "p"
        t = d.rsplit(" ", 1)[0].replace("const", "").strip() if " " in d else d
        return {"int64_t": "i64", "int": "i32", "float": "f32", "size_t": "sz", "uint64_t": "u64", "hipStream_t": "p"}[t]
    seen = set()
    for ret, name, args in protos:
        seen.add(name)
        assert name in _lib.SIGNATURES, f"{name} is declared in the header but missing from the ctypes table"
        restype, argtypes = _lib.SIGNATURES[name]
        want = [k for k in (kind(a) for a in args.split(",")) if k is not None]
        got = []
        for t in argtypes:
            got.append(next(k for k, v in kinds.items() if v is t))
        # char* buffers are bound as c_char_p or c_void_p alike
        # (c_size_t and c_uint64 are one ctypes class on this platform: one kind)
        norm = lambda ks: ["p" if k == "str" else ("u64" if k == "sz" else k) for k in ks]
        assert norm(got) == norm(want), f"{name}: ctypes {got} vs header {want}"
        rk = {"int": C.c_int, "size_t": C.c_size_t, "int64_t": C.c_int64}.get(ret.strip())
        if rk is not None:
            assert restype is rk, f"{name}: return type {restype} vs header {ret}"
        else:
            assert restype is C.c_char_p
    assert seen == set(_lib.SIGNATURES)


def test_student_layer_form_table():
    """models._Student._layer_form: WHICH kernels run a hidden layer (conv -> BatchNorm -> ReLU -> dropout, gnn.py:47-50) is a function of
    mode, grad mode, device, adjacency kind and module types only.  The decision table, checked on the host with a stand-in for a GPU
    tensor (the forms themselves are compared with their composed counterparts on the GPU: tests/test_gpu_parity.py)."""
    import types
    import efficient_gnns_amd.dist as DD
    import efficient_gnns_amd.models as PM
    gpu = types.SimpleNamespace(is_cuda=True, requires_grad=False)          # what _layer_form reads of the layer input
    cpu = types.SimpleNamespace(is_cuda=False, requires_grad=False)
    adj = E.SparseTensor(rowptr=torch.tensor([0, 1, 2]), col=torch.tensor([1, 0]), sparse_sizes=(2, 2))
    sharded = types.SimpleNamespace(gcn_normalized=lambda: None, aggregate=None)

    def forms(model, x, a):
        return [model._layer_form(li, x, a) for li in range(len(model.bns))]

    gcn = PM.GCN(128, 256, 40, 3, 0.5)
    gcn.train()
    assert forms(gcn, gpu, adj) == [("stats", "bn_act"), ("stats", "tail")]            # training on one GPU: statistics epilogue, fused tail
    assert forms(gcn, cpu, adj) == [("plain", "torch"), ("plain", "torch")]            # (the gloo tests' stand-in route)
    with torch.no_grad():
        assert forms(gcn, gpu, adj) == [("stats", "bn_act"), ("stats", "bn_act")]      # no autograd: no fused tail
    gcn.eval()
    with torch.no_grad():
        assert forms(gcn, gpu, adj) == [("fold", None), ("fold", None)]                # test(): BatchNorm folded into the conv
    assert forms(gcn, gpu, adj) == [("plain", "bn_act"), ("plain", "bn_act")]          # eval with autograd on: no fold
    prev = PM._FUSED_TAIL
    try:
        PM._FUSED_TAIL = False
        gcn.train()
        assert forms(gcn, gpu, adj) == [("stats", "bn_act"), ("stats", "bn_act")]
    finally:
        PM._FUSED_TAIL = prev
    wide_out = PM.GCN(128, 256, 300, 3, 0.5)                                          # 300 classes > 64: the tail kernels do not apply
    wide_out.train()
    assert forms(wide_out, gpu, adj)[-1] == ("stats", "bn_act")
    sage = PM.SAGE(128, 256, 40, 3, 0.5)
    sage.train()
    assert forms(sage, gpu, adj) == [("plain", "bn_act"), ("plain", "bn_act")]
    # node-range shards: SyncBatchNorm1d, the fused tail with all-rank statistics for a GCN, the eval fold as well
    sgcn = DD.swap_batchnorm(PM.GCN(128, 256, 40, 3, 0.5))
    sgcn.train()
    assert forms(sgcn, gpu, sharded) == [("plain", "sync"), ("plain", "tail_sync")]
    assert forms(sgcn, cpu, sharded) == [("plain", "sync"), ("plain", "sync")]
    sgcn.eval()
    with torch.no_grad():
        assert forms(sgcn, gpu, sharded) == [("fold", None), ("fold", None)]
    ssage = DD.swap_batchnorm(PM.SAGE(128, 256, 40, 3, 0.5))
    ssage.train()
    assert forms(ssage, gpu, sharded) == [("plain", "sync"), ("plain", "sync")]


def test_plain_bench_command_starts_the_ranks_itself():
    """``python bench.py --gpus 2`` with NO launcher in front and WORLD_SIZE unset must start two ranks by itself (it used
    to run the single-GPU path and print n_gpus 1).  Host logic only: tests/bench_cpu_harness.py puts the gloo tests' stand-ins in
    place of the kernels in every rank it is re-launched as; the line must say n_gpus 2 and carry a non-empty halo exchange."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_harness.py"), "--gpus", "2", "--device", "cpu", "--scale", "0.004",
           "--hidden", "32", "--max-samples", "128", "--steps", "2", "--warmup", "1", "--graph", "off"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "starting 2 ranks" in r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["steps"] == 2
    per_rank = out["comm_per_epoch"]["per_rank"]
    assert len(per_rank) == 2 and all(c["halo_all_to_all_bytes_sent"] > 0 for c in per_rank)
    # a launcher's WORLD_SIZE that disagrees with --gpus is an error, never a silently different run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_tensor_keyed_cache_is_bounded_by_bytes_as_well():
    """the plane images of constant operands are hundreds of MB each -- their caches are bounded by bytes, not only by
    entry count; the newest entry always stays."""
    from efficient_gnns_amd._cache import TensorKeyedCache
    c = TensorKeyedCache(capacity=8, max_bytes=1000)
    keys = [torch.zeros(1) for _ in range(4)]
    for k in keys[:3]:
        c.get((k,), (), lambda: torch.zeros(100, dtype=torch.uint8))          # 3 x 100 B
    assert len(c) == 3
    c.get((keys[3],), (), lambda: torch.zeros(900, dtype=torch.uint8))        # 1200 B > bound: the oldest entries go
    assert len(c) == 2 and c.get((keys[3],), (), lambda: None) is not None
    big = torch.zeros(1)
    c.get((big,), (), lambda: torch.zeros(5000, dtype=torch.uint8))           # alone above the bound: kept (never an empty cache)
    assert len(c) == 1


def test_launcher_scopes_the_repointed_torch_modules_to_the_script(tmp_path):
    """``dropin/launch.py`` re-points torch.nn.BatchNorm1d / Linear / Adam for the RUN OF THE SCRIPT only: inside the
    script the package's methods are in place (deferred activations on by default, off with --no-deferred-activations, nothing re-pointed
    with --plain-torch-modules); when ``main()`` returns -- or the script raises -- torch's own methods are back."""
    import importlib.util
    import torch.nn as tnn
    bn0, lin0 = tnn.BatchNorm1d.forward, tnn.Linear.forward
    script = tmp_path / "probe_script.py"
    script.write_text("import sys, torch\n"
                      "open(sys.argv[1], 'w').write(torch.nn.BatchNorm1d.forward.__name__ + ' ' + torch.nn.Linear.forward.__name__ + ' ' +\n"
                      "    str(sys.modules['egnn_dropin_accel'].LAZY if 'egnn_dropin_accel' in sys.modules else None))\n"
                      "if len(sys.argv) > 2: raise RuntimeError('boom')\n")
    spec = importlib.util.spec_from_file_location("egnn_launch_t", os.path.join(ROOT, "efficient-gnns_amd", "dropin", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    saved_path, saved_argv = list(sys.path), list(sys.argv)
    try:
        for flags, want in (([], "fast_bn fast_linear"), (["--no-deferred-activations"], "fast_bn fast_linear"), (["--plain-torch-modules"], "forward forward")):
            out = tmp_path / "out.txt"
            launch.main(flags + [str(script), str(out)])
            assert out.read_text().startswith(want), (flags, out.read_text())
            assert tnn.BatchNorm1d.forward is bn0 and tnn.Linear.forward is lin0, "restored after the script's run"
        with pytest.raises(RuntimeError, match="boom"):
            launch.main([str(script), str(tmp_path / "out2.txt"), "raise"])
        assert tnn.BatchNorm1d.forward is bn0 and tnn.Linear.forward is lin0, "restored after an exception as well"
    finally:
        sys.path[:], sys.argv[:] = saved_path, saved_argv
