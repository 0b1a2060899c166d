"""Hand-computed known answers for the third-party semantics the oracle restates.

The reference holds no fixtures for these ("parity unpinned"), so they are pinned to
values worked out by hand on a 5-node toy graph.
"""
import math

import numpy as np
import torch

import oracle.nn as ON
import oracle.sparse as OS
import oracle.utils as OU

# directed edges source->target; includes a duplicate (0->1 twice), a self loop (2->2) and an
# isolated node (4)
EI = torch.tensor([[0, 0, 1, 2, 3, 2], [1, 1, 2, 2, 0, 3]])
N = 5


def test_to_sparse_tensor_rows_are_targets():
    a = OS.to_sparse_tensor(EI, N)
    rowptr, col, val = a.csr()
    # targets: 0<-{3}, 1<-{0,0}, 2<-{1,2}, 3<-{2}, 4<-{}
    assert rowptr.tolist() == [0, 1, 3, 5, 6, 6]
    assert col.tolist() == [3, 0, 0, 1, 2, 2]
    assert val is None


def test_to_symmetric_union_sorted_deduped():
    s = OS.to_sparse_tensor(EI, N).to_symmetric()
    rowptr, col, _ = s.csr()
    # undirected pairs {0,1},{0,3},{1,2},{2,3} + loop (2,2)
    assert rowptr.tolist() == [0, 2, 4, 7, 9, 9]
    assert col.tolist() == [1, 3, 0, 2, 1, 2, 3, 0, 2]
    # symmetric => transpose meta reproduces the same arrays
    assert s._colptr().tolist() == rowptr.tolist()
    assert s._row()[s._csr2csc()].tolist() == col.tolist()


def test_gcn_norm_hand_values():
    s = OS.to_sparse_tensor(EI, N).to_symmetric()
    g = OS.gcn_norm_sparse(s)
    rowptr, col, val = g.csr()
    # diagonal replaced/inserted in sorted position; degrees incl. self loop: 3,3,3,3,1
    assert rowptr.tolist() == [0, 3, 6, 9, 12, 13]
    assert col.tolist() == [0, 1, 3, 0, 1, 2, 1, 2, 3, 0, 2, 3, 4]
    third = 1.0 / 3.0
    np.testing.assert_allclose(val.numpy(), [third] * 12 + [1.0], rtol=1e-6)


def test_gcn_norm_edge_index_branch():
    ei = torch.tensor([[0, 1, 1, 2, 2], [1, 0, 2, 1, 2]])  # has loop (2,2)
    e2, w = OS.gcn_norm_edge_index(ei, 3)
    # loops removed from their slots, all 3 appended at the end
    assert e2.tolist() == [[0, 1, 1, 2, 0, 1, 2], [1, 0, 2, 1, 0, 1, 2]]
    deg = np.array([2.0, 3.0, 2.0])  # in-degree by target incl. loop
    dinv = deg ** -0.5
    exp = [dinv[r] * dinv[c] for r, c in zip(*e2.tolist())]
    np.testing.assert_allclose(w.numpy(), exp, rtol=1e-6)


def test_matmul_sum_mean_max():
    a = OS.to_sparse_tensor(EI, N)
    x = torch.tensor([[1.0, -1.0], [2.0, 5.0], [3.0, 0.5], [4.0, -2.0], [9.0, 9.0]])
    np.testing.assert_allclose(a.matmul(x, "sum").numpy(), [[4, -2], [2, -2], [5, 5.5], [3, 0.5], [0, 0]])
    np.testing.assert_allclose(a.matmul(x, "mean").numpy(), [[4, -2], [1, -1], [2.5, 2.75], [3, 0.5], [0, 0]])
    out, arg = OS.matmul_max_with_arg(a, x)
    np.testing.assert_allclose(out.numpy(), [[4, -2], [1, -1], [3, 5], [3, 0.5], [0, 0]])
    # ties (duplicate entries 1,2 of row 1) -> first stored entry; empty row -> -1
    assert arg.tolist() == [[0, 0], [1, 1], [4, 3], [5, 5], [-1, -1]]
    np.testing.assert_allclose(OS.spmm_loops(a, x, "mean").numpy(), a.matmul(x, "mean").numpy())


def test_matmul_backward_is_transpose():
    torch.manual_seed(0)
    a = OS.gcn_norm_sparse(OS.to_sparse_tensor(EI, N))
    x = torch.randn(N, 3, requires_grad=True)
    g = torch.randn(N, 3)
    a.matmul(x, "sum").backward(g)
    dense = torch.zeros(N, N)
    row, col, val = a.coo()
    dense.index_put_((row, col), val, accumulate=True)
    np.testing.assert_allclose(x.grad.numpy(), (dense.t() @ g).numpy(), rtol=1e-5, atol=1e-6)
    x.grad = None
    a.matmul(x, "mean").backward(g)
    cnt = (a.storage.rowcount()).clamp(min=1).float()
    np.testing.assert_allclose(x.grad.numpy(), (dense.t() @ (g / cnt[:, None])).numpy(), rtol=1e-5, atol=1e-6)
    x.grad = None
    out, arg = OS.matmul_max_with_arg(a.set_value(None), x)
    out.backward(g)
    exp = torch.zeros(N, 3)
    for i in range(N):
        for k in range(3):
            if arg[i, k] >= 0:
                exp[col[arg[i, k]], k] += g[i, k]
    np.testing.assert_allclose(x.grad.numpy(), exp.numpy(), rtol=1e-6)


def test_segment_softmax_and_subgraph():
    src = torch.tensor([1.0, 2.0, 3.0, -1.0])
    idx = torch.tensor([0, 0, 2, 2])
    p = OU.softmax(src, idx)
    e = math.exp(-1.0)
    np.testing.assert_allclose(p.numpy(), [e / (1 + e), 1 / (1 + e), 1 / (1 + math.exp(-4)), math.exp(-4) / (1 + math.exp(-4))], rtol=1e-6)
    ei = torch.tensor([[0, 1, 2, 3, 4], [1, 2, 3, 4, 0]])
    sub, _ = OU.subgraph(torch.tensor([3, 1, 2]), ei, relabel_nodes=True)
    assert sub.tolist() == [[1, 2], [2, 0]]  # edges 1->2, 2->3 relabelled by position in subset


def test_conv_parameter_counts_match_paper():
    """SURVEY section 6 cross-check: GCN-2L-256 = 43 816 ("44K"), SAGE-2L-256 = 86 824 ("87K")."""
    import oracle.models as OM
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(OM.GCN(128, 256, 40, 2, 0.5)) == 43816
    assert n(OM.SAGE(128, 256, 40, 2, 0.5)) == 86824
    assert n(OM.GCN(128, 256, 40, 3, 0.5)) == 110120
    c = ON.GCNConv(3, 4, cached=True)
    assert tuple(c.weight.shape) == (3, 4) and float(c.bias.abs().sum()) == 0.0
    s = ON.SAGEConv(3, 4)
    assert s.lin_l.bias is not None and s.lin_r.bias is None


def test_gatconv_hand_values():
    """PyG <=1.7 GATConv on a 3-node graph, one head, identity-like weights: scores, softmax and aggregation by hand."""
    import math
    import oracle.nn as ON
    conv = ON.GATConv(2, 2, heads=1, bias=True)
    with torch.no_grad():
        conv.lin_l.weight.copy_(torch.eye(2))
        conv.att_l.copy_(torch.tensor([[[1.0, 0.0]]]))     # alpha_l[i] = x[i,0]
        conv.att_r.copy_(torch.tensor([[[0.0, 1.0]]]))     # alpha_r[i] = x[i,1]
        conv.bias.copy_(torch.tensor([0.5, -0.5]))
    x = torch.tensor([[1.0, 2.0], [3.0, -1.0], [-2.0, 0.5]])
    edge_index = torch.tensor([[1, 2, 0, 0], [0, 0, 1, 0]])   # 1->0, 2->0, 0->1, and a self loop 0->0 that gets replaced
    out = conv(x, edge_index)
    lrelu = lambda v: v if v > 0 else 0.2 * v  # noqa: E731
    # target 0: sources {1, 2, 0(loop)}; score = lrelu(x[src,0] + x[0,1])
    s0 = [lrelu(3.0 + 2.0), lrelu(-2.0 + 2.0), lrelu(1.0 + 2.0)]
    e0 = [math.exp(v - max(s0)) for v in s0]
    a0 = [v / (sum(e0) + 1e-16) for v in e0]
    row0 = [a0[0] * 3.0 + a0[1] * -2.0 + a0[2] * 1.0 + 0.5, a0[0] * -1.0 + a0[1] * 0.5 + a0[2] * 2.0 - 0.5]
    # target 1: sources {0, 1(loop)}
    s1 = [lrelu(1.0 - 1.0), lrelu(3.0 - 1.0)]
    e1 = [math.exp(v - max(s1)) for v in s1]
    a1 = [v / (sum(e1) + 1e-16) for v in e1]
    row1 = [a1[0] * 1.0 + a1[1] * 3.0 + 0.5, a1[0] * 2.0 + a1[1] * -1.0 - 0.5]
    # target 2: only its self loop -> coefficient 1
    row2 = [-2.0 + 0.5, 0.5 - 0.5]
    torch.testing.assert_close(out, torch.tensor([row0, row1, row2]), rtol=1e-6, atol=1e-6)
    assert sorted(conv.state_dict()) == ["att_l", "att_r", "bias", "lin_l.weight", "lin_r.weight"]
    # two heads averaged (concat=False) = mean of the per-head results; parameter shapes of the PPI teacher's last layer
    c2 = ON.GATConv(4, 3, heads=6, concat=False)
    assert tuple(c2.lin_l.weight.shape) == (18, 4) and tuple(c2.att_l.shape) == (1, 6, 3) and tuple(c2.bias.shape) == (3,)
    assert tuple(c2(torch.randn(5, 4), torch.tensor([[0, 1], [1, 2]])).shape) == (5, 3)
