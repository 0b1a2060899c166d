"""ORACLE (test infrastructure): PyG <=1.7 ``GCNConv`` / ``SAGEConv``.

Call sites restated: /root/reference/arxiv_pyg/gnn.py:13,28-35,61-67,92 (SparseTensor input,
``cached=True``) and /root/reference/ppi_pyg/gnn.py:125-132,158-164 (``edge_index`` LongTensor input,
``cached=False``).  Parameter layout is the PyG <=1.7 one: ``GCNConv.weight [in,out]``, ``bias [out]``;
``SAGEConv.lin_l`` (bias) / ``lin_r`` (no bias).
``GATConv`` (the PPI teacher, /root/reference/ppi_pyg/gnn.py:86-117): PyG 1.6/1.7 semantics restated from memory of
that release (parity unpinned, like the other third-party operators): ``lin_l`` shared with ``lin_r`` (no bias),
``att_l`` / ``att_r`` [1,H,C] glorot, self loops replaced (remove_self_loops + add_self_loops), LeakyReLU(0.2) scores,
``utils.softmax`` over the edges of a target (+1e-16), dropout on the coefficients, sum aggregation, head concat or
mean, bias zeros.
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


class GATConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True, negative_slope: float = 0.2,
                 dropout: float = 0.0, add_self_loops: bool = True, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, heads, concat
        self.negative_slope, self.dropout, self.add_self_loops = negative_slope, dropout, add_self_loops
        self.lin_l = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.lin_r = self.lin_l
        self.att_l = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_r = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.empty(heads * out_channels if concat else out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        def glorot(t):
            a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
            with torch.no_grad():
                t.uniform_(-a, a)
        glorot(self.lin_l.weight)
        glorot(self.att_l)
        glorot(self.att_r)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    @staticmethod
    def _edges(adj, n: int, add_self_loops: bool):
        """(src, dst) in the order PyG would see them: loops removed, then (i,i) appended for every node."""
        if isinstance(adj, SparseTensor):
            rowptr, col, _ = adj.csr()
            dst = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
            src = col
        else:
            src, dst = adj[0], adj[1]
        if add_self_loops:
            keep = src != dst
            loops = torch.arange(n, dtype=src.dtype)
            src, dst = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
        return src, dst

    def forward(self, x: Tensor, adj) -> Tensor:
        from .utils import softmax
        n, H, C = x.shape[0], self.heads, self.out_channels
        xl = self.lin_l(x).view(n, H, C)
        alpha_l = (xl * self.att_l).sum(-1)
        alpha_r = (xl * self.att_r).sum(-1)
        src, dst = self._edges(adj, n, self.add_self_loops)
        e = torch.nn.functional.leaky_relu(alpha_l[src] + alpha_r[dst], self.negative_slope)   # [E,H]
        a = torch.stack([softmax(e[:, h], dst, num_nodes=n) for h in range(H)], dim=1)
        a = torch.nn.functional.dropout(a, p=self.dropout, training=self.training)
        out = torch.zeros(n, H, C, dtype=x.dtype).index_add_(0, dst, xl[src] * a.unsqueeze(-1))
        out = out.reshape(n, H * C) if self.concat else out.mean(dim=1)
        if self.bias is not None:
            out = out + self.bias
        return out


class MessagePassing(nn.Module):
    """The slice of PyG's ``MessagePassing`` the reference's own ``RGCNConv`` subclass uses (mag_pyg/gnn.py:25-68):
    ``propagate(edge_index, x=..., **kw)`` = gather source rows, ``self.message(x_j, **kw)``, aggregate at the targets
    (``aggr`` in {'mean', 'add'}; targets without messages get 0).  flow = source_to_target."""

    def __init__(self, aggr: str = "add"):
        super().__init__()
        self.aggr = aggr

    def propagate(self, edge_index: Tensor, x: Tensor, **kw) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        msg = self.message(x[src], **kw)
        out = torch.zeros(x.shape[0], msg.shape[1], dtype=msg.dtype).index_add_(0, dst, msg)
        if self.aggr == "mean":
            cnt = torch.zeros(x.shape[0], dtype=msg.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=msg.dtype))
            out = out / cnt.clamp(min=1).unsqueeze(1)
        return out


class RGCNConv(MessagePassing):
    """/root/reference/mag_pyg/gnn.py:25-68 restated: per edge type a mean over the incoming edges of that type of
    ``rel_lins[type](x_j)`` (no bias), plus ``root_lins[node type](x_i)`` (bias)."""

    def __init__(self, in_channels, out_channels, num_node_types, num_edge_types):
        super().__init__(aggr="mean")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_node_types, self.num_edge_types = num_node_types, num_edge_types
        self.rel_lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(num_edge_types)])
        self.root_lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=True) for _ in range(num_node_types)])

    def reset_parameters(self):
        for lin in list(self.rel_lins) + list(self.root_lins):
            lin.reset_parameters()

    def forward(self, x, edge_index, edge_type, node_type):
        out = x.new_zeros(x.size(0), self.out_channels)
        for i in range(self.num_edge_types):
            mask = edge_type == i
            out = out + self.propagate(edge_index[:, mask], x=x, edge_type=i)
        for i in range(self.num_node_types):
            mask = node_type == i
            out = out.index_add(0, torch.nonzero(mask).view(-1), self.root_lins[i](x[mask]))
        return out

    def message(self, x_j, edge_type: int):
        return self.rel_lins[edge_type](x_j)


class DGLGATConv(nn.Module):
    """The arxiv GAT teacher's layer, /root/reference/arxiv_dgl/models.py:95-236 (the reference's OWN module on top of DGL
    message passing), restated on a ``SparseTensor`` whose row i lists the sources j of the edges j -> i (the DGL graph
    after ``gat.py:56-71``: bidirected, self loops replaced).  Same parameter names (``fc``, ``attn_l``, ``attn_r``,
    ``res_fc``) and initialisation (xavier_normal, relu gain) as the reference; returns [N, H, F].

    DGL semantics restated (third party, parity unpinned): ``u_add_v('el','er','e')`` = el[src] + er[dst] per edge,
    ``copy_u``; ``edge_softmax`` = softmax over the incoming edges of each destination, per head; ``update_all(u_mul_e,
    sum)`` = sum over incoming edges of ft[src] * a; ``out_degrees`` / ``in_degrees`` = edges leaving / entering a node.
    Inference semantics: ``edge_drop`` acts in training mode only (models.py:205-210) and teacher training is out of scope."""

    def __init__(self, in_feats, out_feats, num_heads=1, feat_drop=0.0, attn_drop=0.0, edge_drop=0.0, negative_slope=0.2,
                 use_attn_dst=True, residual=False, activation=None, allow_zero_in_degree=False, use_symmetric_norm=False):
        super().__init__()
        self._num_heads, self._in_feats, self._out_feats = num_heads, in_feats, out_feats
        self._allow_zero_in_degree, self._use_symmetric_norm = allow_zero_in_degree, use_symmetric_norm
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        if use_attn_dst:
            self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        else:
            self.register_buffer("attn_r", None)
        self.feat_drop, self.attn_drop, self.edge_drop = nn.Dropout(feat_drop), nn.Dropout(attn_drop), edge_drop
        self.negative_slope = negative_slope
        if residual:
            self.res_fc = nn.Linear(in_feats, num_heads * out_feats, bias=False)
        else:
            self.register_buffer("res_fc", None)
        self._activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        if isinstance(self.attn_r, nn.Parameter):
            nn.init.xavier_normal_(self.attn_r, gain=gain)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def forward(self, adj: SparseTensor, feat: Tensor) -> Tensor:
        if self.training and (self.edge_drop > 0 or self.attn_drop.p > 0 or self.feat_drop.p > 0):
            raise NotImplementedError("DGLGATConv restates the teacher's inference (eval-mode) forward")
        n, H, F_ = feat.shape[0], self._num_heads, self._out_feats
        rowptr, col, _ = adj.csr()
        in_deg = rowptr[1:] - rowptr[:-1]
        if not self._allow_zero_in_degree and bool((in_deg == 0).any()):
            raise AssertionError("zero in-degree node (models.py:167-169)")
        dst = torch.repeat_interleave(torch.arange(n), in_deg)
        src = col
        h = self.feat_drop(feat)
        feat_src = self.fc(h).view(n, H, F_)
        feat_dst = feat_src                                           # non-block graph (models.py:178-180): bound BEFORE the scaling below
        if self._use_symmetric_norm:
            out_deg = torch.bincount(src, minlength=n).float().clamp(min=1)
            feat_src = feat_src * torch.pow(out_deg, -0.5).view(n, 1, 1)
        el = (feat_src * self.attn_l).sum(dim=-1)                     # [n, H]: from the SCALED source features
        e = el[src]
        if self.attn_r is not None:
            e = e + (feat_dst * self.attn_r).sum(dim=-1)[dst]         # from the UNSCALED ones (models.py:182-200, pinned by the golden)
        e = torch.nn.functional.leaky_relu(e, self.negative_slope)    # [E, H]
        m = torch.full((n, H), float("-inf")).scatter_reduce(0, dst.view(-1, 1).expand(-1, H), e, "amax", include_self=True)
        ex = torch.exp(e - m[dst])
        a = ex / torch.zeros(n, H).index_add_(0, dst, ex)[dst]        # dgl.ops.edge_softmax
        a = self.attn_drop(a)
        rst = torch.zeros(n, H, F_).index_add_(0, dst, feat_src[src] * a.unsqueeze(-1))
        if self._use_symmetric_norm:
            rst = rst * torch.pow(in_deg.float().clamp(min=1), 0.5).view(n, 1, 1)
        if self.res_fc is not None:
            rst = rst + self.res_fc(h).view(n, -1, F_)
        if self._activation is not None:
            rst = self._activation(rst)
        return rst
