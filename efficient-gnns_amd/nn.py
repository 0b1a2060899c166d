"""``torch_geometric.nn.{GCNConv, SAGEConv}`` stand-ins on the gfx950 kernels.

Same constructor / forward signatures, parameter names and layouts as PyG <=1.7, which is what the
reference's ``state_dict`` files hold (/root/reference/arxiv_pyg/gnn.py:13,28-35,61-67,92;
/root/reference/ppi_pyg/gnn.py:125-132,158-164).
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from . import ops
import os

from . import _lib
from .sparse import SparseTensor, _ind2ptr, csr_from_coo, gcn_norm

# opt-in: the headline bench keeps the reference's per-step work (aggregate every layer every step)
_MEMOISE_AX = os.environ.get("EGNN_GCN_MEMOISE_AX", "0") == "1"
_SAGE_FUSED = True          # SAGEConv as one autograd node with accumulating stores (ops._SageLayer); tests compare with the composed form
_SAGE_NARROW_FIRST = True   # SAGEConv aggregates lin_l(x) when out < in (r03: SAGE + LSP 124.6 -> 136.1 epochs/s)


def _adj_from_edge_index(edge_index: Tensor, n: int, value: Tensor | None = None) -> SparseTensor:
    """edge_index = (source, target) -> CSR with rows = targets, stable in edge order (PPI path)."""
    src, dst = edge_index[0], edge_index[1]
    perm = torch.argsort(dst, stable=True)
    return SparseTensor(rowptr=_ind2ptr(dst[perm].contiguous(), n), col=src[perm],
                        value=None if value is None else value[perm], sparse_sizes=(n, n))


def _gcn_norm_edge_index(edge_index: Tensor, n: int) -> SparseTensor:
    """PyG <=1.7 ``gcn_norm`` on an edge_index (ppi_pyg/gnn.py:125-132): add_remaining_self_loops, then
    D^-1/2 A D^-1/2 with in-degrees by target.  Every loop carries weight 1, so the result equals the
    SparseTensor branch applied to the target-major CSR with duplicates kept."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loops = torch.arange(n, dtype=torch.int64, device=edge_index.device)
    src2, dst2 = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
    w = torch.ones(src2.numel(), dtype=torch.float32, device=edge_index.device)
    deg = torch.zeros(n, dtype=torch.float32, device=edge_index.device).index_add_(0, dst2, w)
    dinv = deg.pow(-0.5)
    dinv.masked_fill_(dinv == float("inf"), 0.0)
    val = dinv[src2] * w * dinv[dst2]
    return _adj_from_edge_index(torch.stack([src2, dst2]), n, val)


from ._cache import TensorKeyedCache as _TensorKeyedCache  # noqa: E402  (LRU, pins entries a captured graph reads: _cache.py)


_GCN_NORM_CACHE = _TensorKeyedCache(capacity=64)


class GCNConv(nn.Module):
    """out = A^ (x W) + b;  W [in,out] glorot, b zeros;  ``cached=True`` keeps A^ until reset_parameters().

    The aggregation runs on the narrower side of W: A^ (x W) when in >= out (the reference's order, gnn.py:47) and
    (A^ x) W when in < out -- the same product, re-associated so that the HBM-bound gather moves `in` instead of `out`
    floats per neighbour (ogbn-arxiv layer 1: 128 instead of 256), and a constant input needs no aggregation at all in
    the backward (dW = (A^ x)^T dOut)."""

    def __init__(self, in_channels: int, out_channels: int, cached: bool = False, bias: bool = True, **_):
        super().__init__()
        self.in_channels, self.out_channels, self.cached = in_channels, out_channels, cached
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._cached_adj_t = None
        self._cached_ax = None  # (key, A^ x) for a constant input tensor, see forward()
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        with torch.no_grad():
            self.weight.uniform_(-a, a)
            if self.bias is not None:
                self.bias.zero_()
        self._cached_adj_t = None
        self._cached_ax = None

    def forward(self, x: Tensor, edge_index, edge_weight=None, bn_stats_shift: Tensor | None = None, want_bn_stats: bool = False,
                eval_bn=None, xw: Tensor | None = None) -> Tensor:
        """``bn_stats_shift`` / ``want_bn_stats`` (extension, off by default): the caller applies a BatchNorm to the result next
        and wants its column statistics formed in the aggregation's epilogue (``ops.spmm``).
        ``eval_bn`` (extension, no autograd): an eval-mode ``BatchNorm1d`` that follows, with a ReLU behind it (gnn.py:47-49 under
        ``model.eval()``): returns ``relu(eval_bn(conv(x)))`` with the BatchNorm folded into W / b (``ops.bn_fold``) and the ReLU
        in the last kernel's store -- the separate normalisation pass disappears from ``test()``.
        ``xw`` (extension): ``x @ self.weight`` when the caller has it already (``ops.bn_act_linear`` forms it together with x); only
        taken on the transform-first order of a plain SparseTensor adjacency."""
        if edge_weight is not None:
            raise NotImplementedError("GCNConv with edge_weight is not used by the reference")
        agg_first = self.in_channels < self.out_channels
        if eval_bn is not None:
            sharded = hasattr(edge_index, "gcn_normalized")
            if torch.is_grad_enabled() or eval_bn.training or not (isinstance(edge_index, SparseTensor) or sharded) or not x.is_cuda:
                raise ValueError("eval_bn: inference only (no_grad, BatchNorm in eval mode, SparseTensor / sharded adjacency on the GPU)")
            if sharded:   # node-range shard: the same fold, bias + ReLU in the store of the aggregation's last piece
                w, b = ops.bn_fold(self.weight, self.bias, eval_bn)
                if agg_first:
                    return ops.gemm_raw(edge_index.gcn_normalized().aggregate(x, "sum"), w, False, False, b, relu=True)
                return edge_index.gcn_normalized().aggregate(ops.gemm_raw(x, w), "sum", bias=b, relu=True)
            norm = self._cached_adj_t
            if norm is None:
                norm = gcn_norm(edge_index)
                if self.cached:
                    self._cached_adj_t = norm
            w, b = ops.bn_fold(self.weight, self.bias, eval_bn)
            if agg_first:
                return ops.gemm_raw(ops.spmm_raw(norm, x, "sum")[0], w, False, False, b, relu=True)
            return ops.spmm_raw(norm, ops.gemm_raw(x, w), "sum", bias=b, relu=True)[0]
        if hasattr(edge_index, "gcn_normalized"):  # node-range shard (dist.ShardedAdj): halo exchange + local rows of A^
            if agg_first:
                return ops.matmul(edge_index.gcn_normalized().aggregate(x, "sum"), self.weight, self.bias)
            return edge_index.gcn_normalized().aggregate(ops.matmul(x, self.weight) if xw is None else xw, "sum", bias=self.bias)
        norm = self._cached_adj_t
        if norm is None:
            if isinstance(edge_index, SparseTensor):
                norm = gcn_norm(edge_index)
            else:
                # ``cached=False`` (PPI: one batch graph per step, ppi_pyg/gnn.py:125) re-normalises on every call in PyG.
                # The result only depends on the integer edge list, so it is memoised on that tensor's identity: the 20
                # training graphs come back every epoch, in every layer, and building A^ costs a device->host read.
                n_nodes = x.shape[0]
                norm = _GCN_NORM_CACHE.get((edge_index,), (n_nodes,), lambda: _gcn_norm_edge_index(edge_index, n_nodes))
            if self.cached:
                self._cached_adj_t = norm
        if (self.cached and _MEMOISE_AX and not x.requires_grad and self.in_channels <= self.out_channels
                and isinstance(edge_index, SparseTensor)):
            # constant input (the first layer's node features): A^ (x W) == (A^ x) W, and A^ x does not change between
            # steps, so it is kept next to the cached A^ (same lifetime) and the per-step aggregation disappears from
            # forward, backward (dW = (A^ x)^T dOut, no dX needed) and eval.  Keyed on the tensor's identity + version.
            key = (x.data_ptr(), x._version, tuple(x.shape), id(norm))
            if self._cached_ax is None or self._cached_ax[0] != key:
                with torch.no_grad():
                    self._cached_ax = (key, ops.spmm_raw(norm, x, "sum")[0], x)   # holds x: its address cannot be reused meanwhile
            out = ops.matmul(self._cached_ax[1], self.weight)
            return ops.add_bias(out, self.bias)
        if agg_first:
            return ops.matmul(ops.spmm(norm, x, "sum"), self.weight, self.bias)
        return ops.spmm(norm, ops.matmul(x, self.weight) if xw is None else xw, "sum", bias=self.bias,  # bias added in the kernel's store
                        bn_stats_shift=bn_stats_shift, want_bn_stats=want_bn_stats)

    def _uses_memoised_input(self, x: Tensor) -> bool:
        """True when forward() would take the opt-in memoised ``A^ x`` route for this input (EGNN_GCN_MEMOISE_AX)."""
        return bool(self.cached and _MEMOISE_AX and not x.requires_grad and self.in_channels <= self.out_channels)

    def __repr__(self):
        return f"GCNConv({self.in_channels}, {self.out_channels})"


class SAGEConv(nn.Module):
    """out = lin_l(aggr_{j in N(i)} x_j) + lin_r(x_i); ``aggr`` in {mean (reference), sum, max}."""

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean", **_):
        super().__init__()
        if aggr == "add":
            aggr = "sum"
        if aggr not in ("mean", "sum", "max"):
            raise ValueError(f"unsupported aggregation '{aggr}'")
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x: Tensor, edge_index) -> Tensor:
        if hasattr(edge_index, "aggregate"):  # node-range shard (dist.ShardedAdj)
            if _SAGE_FUSED and hasattr(edge_index, "sage_layer"):
                # one autograd node with accumulating stores and (out < in) the narrow-first order, as on one GPU (ops._SageLayer)
                out = edge_index.sage_layer(x, self.lin_l, self.lin_r, self.aggr, _SAGE_NARROW_FIRST and self.out_channels < self.in_channels)
                if out is not None:
                    return out
            agg = edge_index.aggregate(x, self.aggr, valueless=True)
            # lin_l(agg) + lin_r(x): the second product is added in the first GEMM's store (ops.linear_add)
            return ops.linear_add(agg, self.lin_l.weight, self.lin_l.bias, ops.linear(x, self.lin_r.weight, None))
        adj = edge_index if isinstance(edge_index, SparseTensor) else _adj_from_edge_index(edge_index, x.shape[0])
        if (_SAGE_FUSED and self.aggr in ("mean", "sum") and x.is_cuda and x.dim() == 2 and self.lin_r.bias is None
                and adj.nnz() > 0 and adj.sparse_size(0) == adj.sparse_size(1)):
            # lin_l(aggr(x)) + lin_r(x) with both sums formed in kernel stores (ops._SageLayer): no element-wise pass forward or backward
            narrow = _SAGE_NARROW_FIRST and self.out_channels < self.in_channels
            return ops.sage_layer(x, adj.set_value(None), self.lin_l, self.lin_r, self.aggr, narrow)
        if _SAGE_NARROW_FIRST and self.out_channels < self.in_channels and self.aggr in ("mean", "sum") and x.is_cuda:
            # mean / sum are linear: aggr_j(x_j) W^T == aggr_j(x_j W^T).  On the output layer (256 -> 40 classes, gnn.py:84) the HBM-bound
            # gather then moves `out` instead of `in` floats per neighbour, forward and backward -- the re-association GCNConv makes.
            # The bias goes in after the aggregation (a row without neighbours gets lin_l(0) = bias either way).
            return ops.spmm(adj.set_value(None), ops.linear(x, self.lin_l.weight, None), self.aggr, bias=self.lin_l.bias) \
                + ops.linear(x, self.lin_r.weight, None)
        agg = ops.spmm(adj.set_value(None), x, self.aggr)
        return ops.linear(agg, self.lin_l.weight, self.lin_l.bias) + ops.linear(x, self.lin_r.weight, None)

    def __repr__(self):
        return f"SAGEConv({self.in_channels}, {self.out_channels}, aggr={self.aggr})"


_GAT_STRUCT_CACHE = _TensorKeyedCache(capacity=64)


class GATConv(nn.Module):
    """PyG <=1.7 ``GATConv`` for INFERENCE: the frozen GAT teacher that the PPI / MAG train loops run inside every student
    step (/root/reference/ppi_pyg/gnn.py:86-117,208-209).  Same parameters and ``state_dict`` keys as PyG 1.6/1.7
    (``lin_l.weight`` shared with ``lin_r``, ``att_l`` / ``att_r`` [1,H,C], ``bias``).

    Forward on the gfx950 kernels: x W on the fp32 MFMA, the 2H attention logits per node as one more small GEMM
    (block-diagonal ``att``), scores + LeakyReLU + per-target softmax fused in ``egnn_gat_attention_fwd_f32`` (no [E,H]
    gathers or scatter-softmax temporaries), one valued SpMM per head written straight into its column block.
    Training the teacher is out of scope: a call that would need gradients raises."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True, negative_slope: float = 0.2,
                 dropout: float = 0.0, add_self_loops: bool = True, bias: bool = True, **_):
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
        with torch.no_grad():
            for t in (self.lin_l.weight, self.att_l, self.att_r):
                a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))  # glorot
                t.uniform_(-a, a)
            if self.bias is not None:
                self.bias.zero_()

    def _structure(self, adj, n: int) -> SparseTensor:
        """CSR by target with the self loops replaced (remove_self_loops + add_self_loops), built on the device.
        Integer preprocessing, memoised on the identity of the index tensor (the PPI teacher meets the same 20 batch
        graphs every epoch, in each of its three layers; PyG rebuilds the loops per call)."""
        keys = (adj._col, adj._rowptr) if isinstance(adj, SparseTensor) else (adj,)
        return _GAT_STRUCT_CACHE.get(keys, (n, self.add_self_loops), lambda: self._build_structure(adj, n))

    def _build_structure(self, adj, n: int) -> SparseTensor:
        if isinstance(adj, SparseTensor):
            src, dst = adj._col, adj._row()
        else:
            src, dst = adj[0], adj[1]
        if self.add_self_loops:
            keep = src != dst
            loops = torch.arange(n, dtype=src.dtype, device=src.device)
            src, dst = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
        rowptr, col = csr_from_coo(dst.contiguous(), src.contiguous(), n, symmetric=False)
        return SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(n, n))

    def forward(self, x: Tensor, edge_index) -> Tensor:
        x = _lib.real(x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("GATConv runs the frozen teacher (torch.no_grad() / requires_grad_(False)); "
                                      "teacher training is out of scope")
        _lib.require_gpu(x)
        n, H, C = x.shape[0], self.heads, self.out_channels
        xl = ops.linear(x, self.lin_l.weight)                                   # [n, H*C]
        # alpha_l[i,h] = <xl[i,h,:], att_l[h,:]> (and alpha_r) as ONE GEMM with a block-diagonal [H*C, 2H] matrix
        blk = torch.zeros(H * C, 2 * H, dtype=torch.float32, device=x.device)
        rows = torch.arange(H * C, device=x.device)
        blk[rows, rows // C] = self.att_l.detach().reshape(-1)
        blk[rows, H + rows // C] = self.att_r.detach().reshape(-1)
        alpha = ops.matmul(xl, blk)                                             # [n, 2H]
        a_src, a_dst = alpha[:, :H].contiguous(), alpha[:, H:].contiguous()
        adj = self._structure(edge_index, n)
        rowptr, col, _ = adj.csr()
        nnz = adj.nnz()
        att = torch.empty(H, nnz, dtype=torch.float32, device=x.device)         # head-major: att[h] is a value array
        _lib.check(_lib.load().egnn_gat_attention_fwd_f32(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(a_src), _lib.ptr(a_dst), n, nnz, H,
                                                          float(self.negative_slope), _lib.ptr(att), _lib.stream()),
                   "egnn_gat_attention_fwd_f32")
        if self.training and self.dropout > 0:
            att = torch.nn.functional.dropout(att, p=self.dropout, training=True)
        out = torch.empty(n, H * C, dtype=torch.float32, device=x.device)
        for h in range(H):
            ops.spmm_raw(adj.set_value(att[h]), xl[:, h * C:(h + 1) * C], "sum", out=out[:, h * C:(h + 1) * C])
        if not self.concat:
            out = out.view(n, H, C).mean(dim=1)
        return ops.add_bias(out, self.bias)

    def __repr__(self):
        return f"GATConv({self.in_channels}, {self.out_channels}, heads={self.heads})"


class DGLGATConv(nn.Module):
    """The arxiv GAT teacher's layer (/root/reference/arxiv_dgl/models.py:95-236 -- the reference's own module over DGL
    message passing) for INFERENCE on the gfx950 kernels: the producer of the ``features/`` / ``logits/`` artefacts the
    student path reads (gat.py:243-258).  Same parameter names and initialisation as the reference
    (``fc``, ``attn_l``, ``attn_r`` (absent with ``use_attn_dst=False``), ``res_fc``), so its checkpoints load.

    ``forward(adj, feat)``: ``adj`` = ``SparseTensor`` whose row i lists the sources of the edges j -> i (the DGL graph after
    gat.py:56-71 ``preprocess``: ``utils.dgl_bidirected_with_self_loops``).  x W on the fp32 MFMA, ``el`` / ``er`` as one more
    small GEMM, ``u_add_v`` + LeakyReLU + ``edge_softmax`` fused in ``egnn_gat_attention_fwd_f32``, ``u_mul_e`` + ``sum`` as
    one valued SpMM per head; symmetric normalisation (out-degree^-1/2 on the sources, in-degree^1/2 on the result) and the
    residual projection as in the reference.  Returns [N, H, F].  Training the teacher is out of scope: training-mode drops
    (``edge_drop`` / ``attn_drop`` / ``feat_drop``) raise."""

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
        self.feat_drop_p, self.attn_drop_p, self.edge_drop = feat_drop, attn_drop, edge_drop
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

    @staticmethod
    def _degrees(adj: SparseTensor):
        """(in-degree^1/2, out-degree^-1/2) of the message graph, both clamped at 1 (models.py:183-186,219-223); cached."""
        st = adj._struct
        if "dgl_norm" not in st:
            rowptr, col, _ = adj.csr()
            n = adj.sparse_size(0)
            in_deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32).clamp(min=1)
            out_deg = torch.bincount(col, minlength=adj.sparse_size(1)).to(torch.float32).clamp(min=1)
            st["dgl_norm"] = (in_deg.pow(0.5).view(n, 1), out_deg.pow(-0.5).view(-1, 1), bool((rowptr[1:] == rowptr[:-1]).any()))
        return st["dgl_norm"]

    def forward(self, adj: SparseTensor, feat: Tensor) -> Tensor:
        feat = _lib.real(feat)
        if torch.is_grad_enabled() and (feat.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("DGLGATConv runs the teacher's inference forward (torch.no_grad()); teacher training is out of scope")
        if self.training and (self.edge_drop > 0 or self.attn_drop_p > 0 or self.feat_drop_p > 0):
            raise NotImplementedError("training-mode edge / attention / feature drop: teacher training is out of scope (use .eval())")
        _lib.require_gpu(feat)
        n, H, F_ = feat.shape[0], self._num_heads, self._out_feats
        in_sqrt, out_rsqrt, has_isolated = self._degrees(adj)
        if has_isolated and not self._allow_zero_in_degree:
            raise AssertionError("zero in-degree node (arxiv_dgl/models.py:167-169)")
        feat_src = ops.linear(feat, self.fc.weight)                               # [n, H*F]
        # el[i,h] = <feat_src[i,h,:], attn_l[h,:]> (and er) as ONE GEMM with a block-diagonal [H*F, 2H] matrix
        blk = torch.zeros(H * F_, 2 * H, dtype=torch.float32, device=feat.device)
        rows = torch.arange(H * F_, device=feat.device)
        blk[rows, rows // F_] = self.attn_l.detach().reshape(-1)
        if self.attn_r is not None:
            blk[rows, H + rows // F_] = self.attn_r.detach().reshape(-1)
        alpha = ops.matmul(feat_src, blk)                                         # [n, 2H]; er = 0 without attn_r (copy_u)
        el, er = alpha[:, :H].contiguous(), alpha[:, H:].contiguous()
        if self._use_symmetric_norm:
            # the reference scales the SOURCE features (and with them el) by out-degree^-1/2 but forms er from the unscaled
            # destination features (models.py:178-200: feat_dst is bound before the scaling)
            feat_src = feat_src * out_rsqrt
            el = el * out_rsqrt
        rowptr, col, _ = adj.csr()
        nnz = adj.nnz()
        att = torch.empty(H, nnz, dtype=torch.float32, device=feat.device)        # head-major: att[h] is a value array
        _lib.check(_lib.load().egnn_gat_attention_fwd_f32(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(el), _lib.ptr(er), n, nnz, H,
                                                          float(self.negative_slope), _lib.ptr(att), _lib.stream()),
                   "egnn_gat_attention_fwd_f32")
        Fp = (F_ + 3) // 4 * 4                                                      # head blocks on 16-byte boundaries (F = 250 -> 252)
        src_heads = feat_src if Fp == F_ else torch.nn.functional.pad(feat_src.view(n, H, F_), (0, Fp - F_)).reshape(n, H * Fp)
        out = torch.empty(n, H * Fp, dtype=torch.float32, device=feat.device)
        plain = adj.set_value(None) if adj.has_value() else adj
        for h in range(H):
            ops.spmm_raw(plain.set_value(att[h]), src_heads[:, h * Fp:(h + 1) * Fp], "sum", out=out[:, h * Fp:(h + 1) * Fp])
        rst = out.view(n, H, Fp)[:, :, :F_]
        if self._use_symmetric_norm:
            rst = rst * in_sqrt.view(n, 1, 1)
        if self.res_fc is not None:
            rst = rst + ops.linear(feat, self.res_fc.weight).view(n, -1, F_)
        if self._activation is not None:
            rst = self._activation(rst)
        return rst

    def __repr__(self):
        return f"DGLGATConv({self._in_feats}, {self._out_feats}, heads={self._num_heads})"


_RGCN_REL_CACHE = _TensorKeyedCache(capacity=8)


class RGCNConv(nn.Module):
    """The reference's own R-GCN layer (/root/reference/mag_pyg/gnn.py:25-68, a ``MessagePassing(aggr='mean')`` subclass) on
    the kernels: per edge type i, ``mean_{j -> t, type i} rel_lins[i](x_j)`` is computed as ``rel_lins[i](mean x_j)`` (the
    linear map has no bias, so it commutes with the mean) -- one mean-SpMM over the type's edges and one MFMA GEMM
    instead of a GEMM over every message; ``root_lins[node type]`` through the fused row-gather GEMM.  Same parameter
    names as the reference (``rel_lins.*``, ``root_lins.*``); differentiable (SpMM / GEMM autograd)."""

    def __init__(self, in_channels, out_channels, num_node_types, num_edge_types):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_node_types, self.num_edge_types = num_node_types, num_edge_types
        self.rel_lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(num_edge_types)])
        self.root_lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=True) for _ in range(num_node_types)])

    def reset_parameters(self):
        for lin in list(self.rel_lins) + list(self.root_lins):
            lin.reset_parameters()

    def _relations(self, edge_index: Tensor, edge_type: Tensor, node_type: Tensor, n: int):
        """Per-type CSR (by target) and per-node-type row lists; integer preprocessing, cached on the tensors' identity."""
        def build():
            adjs = []
            for i in range(self.num_edge_types):
                sel = torch.nonzero(edge_type == i).view(-1)
                if sel.numel() == 0:
                    adjs.append(None)
                    continue
                src, dst = edge_index[0, sel].contiguous(), edge_index[1, sel].contiguous()
                rowptr, col = csr_from_coo(dst, src, n, symmetric=False)
                adjs.append(SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(n, n)))
            rows = [torch.nonzero(node_type == i).view(-1) for i in range(self.num_node_types)]
            return adjs, rows
        return _RGCN_REL_CACHE.get((edge_index, edge_type, node_type), (n, self.num_edge_types, self.num_node_types), build)

    def forward(self, x: Tensor, edge_index: Tensor, edge_type: Tensor, node_type: Tensor) -> Tensor:
        x = _lib.real(x)
        _lib.require_gpu(x, edge_index)
        n = x.shape[0]
        adjs, rows = self._relations(edge_index, edge_type, node_type, n)
        out = torch.zeros(n, self.out_channels, dtype=torch.float32, device=x.device)
        for i, adj in enumerate(adjs):
            if adj is not None:
                out = out + ops.linear(ops.spmm(adj, x, "mean"), self.rel_lins[i].weight)
        for i, idx in enumerate(rows):
            if idx.numel():
                lin = self.root_lins[i]
                out = out.index_add(0, idx, ops.linear_rows(x, idx, lin.weight, lin.bias))
        return out

    def __repr__(self):
        return f"RGCNConv({self.in_channels}, {self.out_channels}, node_types={self.num_node_types}, edge_types={self.num_edge_types})"
