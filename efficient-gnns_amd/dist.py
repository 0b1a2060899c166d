"""Node-range sharded full-graph training over RCCL / xGMI.

The reference has no multi-GPU code (replicas with different seeds only, run_gcn.sh:24-27).  Here rank r owns the
contiguous node range [lo_r, hi_r): its rows of the adjacency, of every activation, of the labels and of the
teacher artefacts.  One process per GPU, ``torch.distributed`` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU
tests).  Collectives on the data path, all with autograd:

  * halo exchange per layer  -- ``all_to_all_single`` of exactly the boundary rows a peer's rows reference
    (send lists are integer preprocessing, built once from the global CSR); backward = the reverse exchange and a
    fixed-order accumulation into the owner's rows;
  * BatchNorm statistics     -- ``all_reduce`` of per-column (sum, centred sum of squares) so that BN matches the
    reference's full-graph batch statistics (SyncBN semantics), forward and backward;
  * G-CRD                    -- every rank draws the same NumPy sample; the sampled unit teacher rows are
    all-gathered, each rank evaluates its own row block of Z on the MFMA kernels and the teacher-side gradient
    is all-reduced;
  * parameters               -- one flat ``all_reduce`` of all gradients per step (~0.5 MB: latency-bound).

Everything else (local aggregation, GEMMs, losses) is the single-GPU kernel path on the local shard.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import _lib, ops
from .sparse import SparseTensor


class CommTrace:
    """Records every torch.distributed collective this rank issues inside the ``with`` block: (op, payload bytes in, payload
    bytes out, per-peer element counts of an all_to_all).  Two uses: (1) the CPU (gloo) test asserts that all ranks issue the
    SAME sequence with matching sizes, step after step, including shards without train rows -- a mismatch is a hang on real
    hardware; (2) ``bench_main`` reports the per-rank communication volume of one epoch in its JSON line.  Also collects the
    overlap probes of ``_OverlapAggregate`` (HIP events: compute window the exchange is hidden under / time the stream then
    still waits for it) when ``probe_overlap`` is set."""

    _OPS = ("all_reduce", "all_to_all_single", "all_gather_into_tensor", "all_gather", "broadcast", "reduce_scatter_tensor", "barrier")
    active = None

    def __init__(self, probe_overlap: bool = False):
        self.records, self.overlap, self.probe_overlap = [], [], probe_overlap

    @staticmethod
    def _bytes(t):
        return 0 if t is None else t.numel() * t.element_size()

    def __enter__(self):
        self._orig = {n: getattr(dist, n) for n in self._OPS}
        rec = self.records

        def wrap(name, orig):
            def f(*a, **kw):
                if name == "all_to_all_single":
                    out, inp = a[0], a[1]
                    osz = a[2] if len(a) > 2 else kw.get("output_split_sizes")
                    isz = a[3] if len(a) > 3 else kw.get("input_split_sizes")
                    width = 1
                    for v in inp.shape[1:]:
                        width *= int(v)
                    rec.append((name, self._bytes(inp), self._bytes(out), width,
                                None if isz is None else tuple(int(v) for v in isz), None if osz is None else tuple(int(v) for v in osz),
                                _A2A_KIND[0], inp.element_size()))
                elif name == "all_gather_into_tensor":
                    rec.append((name, self._bytes(a[1]), self._bytes(a[0]), a[1].numel()))
                elif name == "barrier":
                    rec.append((name, 0, 0, 0))
                else:
                    t = a[0] if torch.is_tensor(a[0]) else a[1]
                    rec.append((name, self._bytes(t), self._bytes(t), t.numel()))
                return orig(*a, **kw)
            return f
        for n, o in self._orig.items():
            setattr(dist, n, wrap(n, o))
        CommTrace.active = self
        return self

    def __exit__(self, *exc):
        for n, o in self._orig.items():
            setattr(dist, n, o)
        CommTrace.active = None

    def summary(self) -> dict:
        """Bytes this rank put on / took off the wire, by kind (an all_reduce counted as 2 (w-1)/w of its payload each way is the
        RING volume; here the plain payload is reported and the ring factor left to the reader)."""
        halo = [r for r in self.records if r[0] == "all_to_all_single" and r[6] != "sliced"]
        a2a_out, a2a_in = sum(r[1] for r in halo), sum(r[2] for r in halo)
        red = sum(r[1] for r in self.records if r[0] == "all_reduce")
        gat = sum(r[2] for r in self.records if r[0].startswith("all_gather"))
        out = dict(collectives=len(self.records), halo_all_to_all_bytes_sent=a2a_out, halo_all_to_all_bytes_received=a2a_in,
                   all_reduce_payload_bytes=red, all_gather_bytes_received=gat)
        sliced = [r for r in self.records if r[0] == "all_to_all_single" and r[6] == "sliced"]
        if sliced:
            # column-sliced aggregations (_SlicedAggregate): what leaves this rank = the payload minus the block it keeps for itself
            me = dist.get_rank() if dist.is_initialized() else 0
            wire = lambda r, sizes: (sum(sizes) - sizes[me]) * r[3] * r[7]   # noqa: E731
            out.update(sliced_exchanges=len(sliced), sliced_all_to_all_bytes_sent=sum(wire(r, r[4]) for r in sliced),
                       sliced_all_to_all_bytes_received=sum(wire(r, r[5]) for r in sliced))
        if self.overlap:
            win = sum(e0.elapsed_time(e1) for e0, e1, _ in self.overlap) * 1e3
            exp = sum(e1.elapsed_time(e2) for _, e1, e2 in self.overlap) * 1e3
            out.update(overlapped_exchanges=len(self.overlap), overlap_window_us=round(win, 1), exposed_comm_us=round(exp, 1))
        return out


def consistent_collectives(per_rank_records) -> str | None:
    """None when the ranks' CommTrace records describe one consistent collective program, else a description of the first
    mismatch: same number and order of operations; equal payload for all_reduce / all_gather; for every all_to_all_single
    what rank r sends to q is what q expects from r."""
    world = len(per_rank_records)
    n = len(per_rank_records[0])
    for r, rec in enumerate(per_rank_records):
        if len(rec) != n:
            return f"rank {r} issued {len(rec)} collectives, rank 0 issued {n}"
    for i in range(n):
        ops_i = [rec[i] for rec in per_rank_records]
        names = {o[0] for o in ops_i}
        if len(names) != 1:
            return f"collective #{i}: ranks disagree on the operation: {sorted(names)}"
        name = ops_i[0][0]
        if name in ("all_reduce", "all_gather_into_tensor", "broadcast", "reduce_scatter_tensor"):
            if len({o[3] for o in ops_i}) != 1:
                return f"collective #{i} ({name}): element counts differ across ranks: {[o[3] for o in ops_i]}"
        elif name == "all_to_all_single":
            if len({o[3] for o in ops_i}) != 1:
                return f"collective #{i} (all_to_all_single): row widths differ: {[o[3] for o in ops_i]}"
            for r in range(world):
                for q in range(world):
                    send = ops_i[r][4][q] if ops_i[r][4] is not None else None
                    recv = ops_i[q][5][r] if ops_i[q][5] is not None else None
                    if send is not None and recv is not None and send != recv:
                        return f"collective #{i} (all_to_all_single): rank {r} sends {send} rows to rank {q}, which expects {recv}"
    return None


# ------------------------------------------------------------------------------------------------
# partition plan (integer, deterministic, identical on every rank)
# ------------------------------------------------------------------------------------------------
def node_range(n: int, world: int, rank: int):
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


_OVERLAP = os.environ.get("EGNN_DIST_OVERLAP", "1") != "0"
# How a sharded aggregation gets its remote operand rows (ShardedAdj.aggregate):
#   halo    the rows a shard's entries reference travel to it (one all_to_all of [halo rows, K] per aggregation): bytes grow with the halo
#   sliced  the feature COLUMNS are re-sharded around the aggregation (all_to_all to [N, K / world] column slices, aggregation of all N
#           rows on the full adjacency, all_to_all back): 2 N K 4 (world - 1) / world^2 bytes per rank, whatever the graph looks like
#   auto    sliced where it moves fewer bytes than the halo (decided once per adjacency from the all-rank mean halo, same on every rank)
_AGG_MODE = os.environ.get("EGNN_DIST_AGG", "auto")
_A2A_KIND = ["halo"]      # what CommTrace files an all_to_all_single under (set around the sliced exchanges)


def _agg(adj, x, addend=None, bias=None, relu=False):
    """Local sum-aggregation of a shard piece (the single-GPU kernels; tests swap in the oracle through ``ops.spmm``).
    ``bias`` / ``relu``: added / applied in the kernel's store (the LAST piece of a row's sum only); ``relu`` is inference only."""
    if relu:
        return ops.spmm_raw(adj, x, "sum", bias=bias, addend=addend, relu=True)[0]
    return ops.spmm(adj, x, "sum", bias=bias, addend=addend)


def halo_rows_per_rank(rowptr: Tensor, col: Tensor, n: int, world: int) -> list:
    """Distinct remote source rows every rank needs per halo exchange (= rows it receives per layer and direction) when the nodes
    are cut into ``world`` contiguous ranges.  Integer, host or device; what locality-aware node orders reduce."""
    out = []
    for r in range(world):
        lo, hi, _ = node_range(n, world, r)
        c = col[int(rowptr[lo]):int(rowptr[hi])]
        out.append(int(torch.unique(c[(c < lo) | (c >= hi)]).numel()))
    return out


def reorder_nodes(data, perm: Tensor):
    """The same problem with its nodes relabelled: new node i is old node ``perm[i]`` -- adjacency (rows and columns), features,
    labels, teacher artefacts and the split index lists (which keep their ORDER: position k of the train list still names the same
    node, so the sampled criteria draw the same rows).  Every loss is invariant under the relabelling (up to summation order)."""
    import types
    n = data.num_nodes
    perm = perm.to("cpu", torch.int64)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, dtype=torch.int64)
    d = types.SimpleNamespace(**vars(data))
    d.adj_t = data.adj_t.permute(perm.to(data.adj_t.device))
    for k in ("x", "y", "teacher_out_feat", "teacher_logits"):
        v = getattr(data, k, None)
        if v is not None:
            setattr(d, k, v[perm])
    d.split_idx = {k: inv[v] for k, v in data.split_idx.items()}
    if getattr(data, "community", None) is not None:
        d.community = data.community[perm]
    d.node_perm = perm
    return d


def locality_order(data, world: int, device=None, **kw):
    """``sparse.community_order`` of the problem's graph (size-capped label propagation, communities contiguous) as the node order
    to cut the ranges from, when it pays: returns (perm | None, halo rows per rank in the given order, ... in the community order).
    perm is None when the community order does not lower the total halo (a graph without locality: the Chung-Lu headline workload).
    Deterministic (seeded), identical on every rank; integer work on ``device`` when given."""
    from .sparse import community_order
    adj = data.adj_t if device is None else data.adj_t.to(device)
    rowptr, col, _ = adj.csr()
    n = data.num_nodes
    before = halo_rows_per_rank(rowptr, col, n, world)
    # communities well below a rank's range, so that ranges hold whole communities
    cap = kw.pop("max_community", max(64, min(4096, n // (4 * max(world, 1)))))
    perm = community_order(adj, max_community=cap, **kw)
    padj = adj.permute(perm)
    prow, pcol, _ = padj.csr()
    after = halo_rows_per_rank(prow, pcol, n, world)
    return (perm.cpu() if sum(after) < sum(before) else None), before, after


class ShardPlan:
    """Local view of one rank's rows of a global CSR: remapped columns + halo send / receive lists.

    Extended column ids: [0, n_local) = the rank's own nodes, n_local + j = halo node ``halo_ids[j]`` (ascending global id, so
    grouped by owner).  ``send_idx``: local row ids the peers need, grouped by peer (each group ascending);
    ``send_counts[p]`` / ``recv_counts[p]``: rows sent to / received from peer p in one halo exchange."""

    def __init__(self, n, world, rank, rowptr_local, col_global, value_local, send_idx, send_counts):
        self.n, self.world, self.rank = n, world, rank
        lo, hi, per = node_range(n, world, rank)
        self.lo, self.hi, self.per, self.n_local = lo, hi, per, hi - lo
        c = col_global
        remote = (c < lo) | (c >= hi)
        self.halo_ids = torch.unique(c[remote])                       # ascending => grouped by owner
        owner = torch.div(self.halo_ids, per, rounding_mode="floor")
        self.recv_counts = torch.bincount(owner, minlength=world).tolist()
        self.col_ext = torch.where(remote, self.n_local + torch.searchsorted(self.halo_ids, c), c - lo).contiguous()
        self.remote_mask = remote
        self.rowptr_local = rowptr_local.contiguous()
        self.value_local = None if value_local is None else value_local.contiguous()
        self.send_idx, self.send_counts = send_idx, list(send_counts)
        self.n_halo = self.halo_ids.numel()

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def from_global(cls, rowptr: Tensor, col: Tensor, value: Tensor | None, n: int, world: int, rank: int) -> "ShardPlan":
        """From the GLOBAL CSR, without communication (single-process tools and tests): what a peer needs from this rank
        is read off the peer's rows directly."""
        lo, hi, per = node_range(n, world, rank)
        e0, e1 = int(rowptr[lo]), int(rowptr[hi])
        row_owner = torch.div(torch.repeat_interleave(torch.arange(n, device=col.device), rowptr[1:] - rowptr[:-1]), per,
                              rounding_mode="floor")
        mine = (col >= lo) & (col < hi) & (row_owner != rank)
        keys = torch.unique(row_owner[mine] * n + col[mine])          # (peer, my node) pairs, sorted by peer then node
        send_idx = keys % n - lo
        send_counts = torch.bincount(torch.div(keys, n, rounding_mode="floor"), minlength=world).tolist()
        return cls(n, world, rank, rowptr[lo:hi + 1] - e0, col[e0:e1], None if value is None else value[e0:e1], send_idx, send_counts)

    @classmethod
    def from_local(cls, rowptr_local: Tensor, col_global: Tensor, value_local: Tensor | None, n: int, world: int, rank: int,
                   group=None) -> "ShardPlan":
        """COLLECTIVE: every rank passes only ITS rows (rowptr starting at 0, global column ids).  The halo lists are agreed
        on by exchanging index lists (one all_to_all of counts, one of ids): no rank ever holds the global structure --
        what the MAG-scale graph needs."""
        lo, hi, per = node_range(n, world, rank)
        plan = cls(n, world, rank, rowptr_local, col_global, value_local, torch.zeros(0, dtype=torch.int64, device=col_global.device),
                   [0] * world)
        if world > 1:
            dev = col_global.device
            need_counts = torch.tensor(plan.recv_counts, dtype=torch.int64, device=dev)      # rows I need from each peer
            give_counts = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_to_all_single(give_counts, need_counts, group=group)
            send_counts = give_counts.tolist()
            wanted = torch.empty(int(sum(send_counts)), dtype=torch.int64, device=dev)
            dist.all_to_all_single(wanted, plan.halo_ids.contiguous(), send_counts, plan.recv_counts, group=group)
            plan.send_idx, plan.send_counts = wanted - lo, send_counts
        return plan

    # ---- the pieces the aggregation uses -----------------------------------------------------------
    def local_adj(self, device) -> SparseTensor:
        """All entries over the extended columns [own | halo] (one aggregation after the exchange)."""
        return SparseTensor(rowptr=self.rowptr_local.to(device), col=self.col_ext.to(device),
                            value=None if self.value_local is None else self.value_local.to(device),
                            sparse_sizes=(self.n_local, self.n_local + self.n_halo))

    def split_adj(self, device, value: Tensor | None = None):
        """(entries with an own column, entries with a halo column): the second one can only run after the exchange, the
        first one overlaps it.  ``value``: per-entry values to use instead of ``value_local``."""
        val = self.value_local if value is None else value
        rows = torch.repeat_interleave(torch.arange(self.n_local, device=self.col_ext.device),
                                       self.rowptr_local[1:] - self.rowptr_local[:-1])
        out = []
        for mask, ncol, off in ((~self.remote_mask, self.n_local, 0), (self.remote_mask, self.n_halo, self.n_local)):
            r, c = rows[mask], self.col_ext[mask] - off
            rp = torch.zeros(self.n_local + 1, dtype=torch.int64, device=r.device)
            torch.cumsum(torch.bincount(r, minlength=self.n_local), 0, out=rp[1:])
            out.append(SparseTensor(rowptr=rp.to(device), col=c.contiguous().to(device),
                                    value=None if val is None else val[mask].contiguous().to(device), sparse_sizes=(self.n_local, ncol)))
        return out

    def scatter_adj(self, device) -> SparseTensor:
        """[n_local, n_send] 0/1 matrix: row send_idx[j] holds entry j.  Aggregating the rows received by the REVERSE
        exchange with it adds every peer's contribution into the owner's rows in a fixed order (columns ascending = peer
        order) -- one kernel, no atomics, instead of one indexed add per peer."""
        order = torch.argsort(self.send_idx, stable=True)
        rp = torch.zeros(self.n_local + 1, dtype=torch.int64, device=self.send_idx.device)
        torch.cumsum(torch.bincount(self.send_idx, minlength=self.n_local), 0, out=rp[1:])
        return SparseTensor(rowptr=rp.to(device), col=order.to(device), sparse_sizes=(self.n_local, self.send_idx.numel()))


class _HaloExchange(torch.autograd.Function):
    """x_local [n_local,K] -> x_ext [n_local + n_halo, K] (own rows first, then halo rows ordered by global id)."""

    @staticmethod
    def forward(ctx, x_local, sadj):
        plan, group = sadj.plan, sadj.group
        K = x_local.shape[1]
        send_buf = x_local.index_select(0, sadj.send_idx_dev).contiguous()
        # the halo rows are received straight into the tail of the extended matrix (no concatenation copy of the halo,
        # which is 7/8 of the matrix at 8 ranks on a graph without locality)
        x_ext = torch.empty(plan.n_local + plan.n_halo, K, dtype=x_local.dtype, device=x_local.device)
        x_ext[:plan.n_local].copy_(x_local)
        dist.all_to_all_single(x_ext[plan.n_local:], send_buf, plan.recv_counts, plan.send_counts, group=group)
        ctx.sadj = sadj
        return x_ext

    @staticmethod
    def backward(ctx, g_ext):
        sadj = ctx.sadj
        plan, group = sadj.plan, sadj.group
        K = g_ext.shape[1]
        g_halo = g_ext[plan.n_local:].contiguous()
        back = torch.empty(plan.send_idx.numel(), K, dtype=g_ext.dtype, device=g_ext.device)
        dist.all_to_all_single(back, g_halo, plan.send_counts, plan.recv_counts, group=group)
        g_local = g_ext[:plan.n_local].contiguous()
        if back.shape[0]:
            g_local = _agg(sadj.scatter, back, addend=g_local)        # fixed-order accumulation into the owners' rows
        return g_local, None


def _overlap_forward(x_local, sadj, a_own, a_halo, static_halo, bias=None, relu=False, addend=None):
    """y = A_own x_own + A_halo x_halo (+ addend) (+ bias, ReLU in the LAST piece's store) with the halo exchange IN FLIGHT under the
    first product.  Plain function (no autograd): the body of ``_OverlapAggregate`` and of ``_ShardedSageLayer``."""
    plan, group = sadj.plan, sadj.group
    K = x_local.shape[1]
    work = None
    if static_halo is not None:
        x_halo = static_halo
    else:
        send_buf = x_local.index_select(0, sadj.send_idx_dev).contiguous()
        x_halo = torch.empty(plan.n_halo, K, dtype=x_local.dtype, device=x_local.device)
        work = dist.all_to_all_single(x_halo, send_buf, plan.recv_counts, plan.send_counts, group=group, async_op=True)
    probe = CommTrace.active is not None and CommTrace.active.probe_overlap and work is not None and x_local.is_cuda
    if probe:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
    last_is_own = not plan.n_halo                              # bias (+ ReLU) go into the store of the row sums' LAST piece
    y = _agg(a_own, x_local, addend=addend, bias=bias if last_is_own else None, relu=relu and last_is_own)   # runs while the halo rows travel
    if probe:
        ev[1].record()
    if work is not None:
        work.wait()
    if probe:   # ev0..ev1: compute the exchange is hidden under; ev1..ev2: what the stream still waits for it afterwards
        ev[2].record()
        CommTrace.active.overlap.append(tuple(ev))
    if plan.n_halo:
        y = _agg(a_halo, x_halo, addend=y, bias=bias, relu=relu)
    return y


def _overlap_backward(g_y, sadj, a_own, a_halo, addend=None):
    """g_x = A_own^T g_y (+ addend) + (the peers' A_halo^T g_y rows, returned by the reverse exchange and added into their owner rows
    in a fixed order), the exchange in flight under the first product."""
    plan, group = sadj.plan, sadj.group
    g_y = g_y.contiguous()
    K = g_y.shape[1]
    back = torch.empty(plan.send_idx.numel(), K, dtype=g_y.dtype, device=g_y.device)
    g_halo = _agg(a_halo.t(), g_y) if plan.n_halo else torch.zeros(0, K, dtype=g_y.dtype, device=g_y.device)
    work = dist.all_to_all_single(back, g_halo, plan.send_counts, plan.recv_counts, group=group, async_op=True)
    g_x = _agg(a_own.t(), g_y, addend=addend)                    # runs while the halo gradients travel back
    work.wait()
    if back.shape[0]:
        g_x = _agg(sadj.scatter, back, addend=g_x)
    return g_x


class _OverlapAggregate(torch.autograd.Function):
    """y = A_own x_own + A_halo x_halo with the halo exchange IN FLIGHT under the first product (and, in the backward, the
    reverse exchange under A_own^T g): the entries of a shard are split by column ownership, so the overlap does not
    depend on the graph having interior rows (on a graph without locality there are none)."""

    @staticmethod
    def forward(ctx, x_local, sadj, a_own, a_halo, static_halo, bias=None, relu=False):
        y = _overlap_forward(x_local, sadj, a_own, a_halo, static_halo, bias, relu)
        ctx.sadj, ctx.pieces, ctx.has_bias = sadj, (a_own, a_halo), bias is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        a_own, a_halo = ctx.pieces
        g_y = g_y.contiguous()
        g_x = _overlap_backward(g_y, ctx.sadj, a_own, a_halo)
        g_b = ops.colsum(g_y) if (ctx.has_bias and ctx.needs_input_grad[5]) else None
        return g_x, None, None, None, None, g_b, None


def _sliced_product(x_local, sadj, full, bias=None, relu=False):
    """y_local = (full @ X)[own rows] for the row-sharded X whose local rows are ``x_local`` -- by re-sharding the COLUMNS around the
    aggregation: every rank receives the column slice [N, K / world] of X that belongs to it (all_to_all), aggregates ALL N rows of that
    slice on the full matrix ``full`` (the local kernels; ``bias`` / ``relu`` of its slice in the store), and the row blocks travel back
    (all_to_all).  Payload per rank and direction: N K 4 (world - 1) / world^2 bytes, independent of the halo.  Plain function (no
    autograd): the body of ``_SlicedAggregate`` forward and, on ``full.t()``, backward."""
    plan, group = sadj.plan, sadj.group
    G, n_local, K = plan.world, plan.n_local, x_local.shape[1]
    Kg = K // G
    rows = sadj.rows_per_rank
    # [n_local, G, Kg] -> [G, n_local, Kg]: the block for peer p is contiguous
    send = x_local.reshape(n_local, G, Kg).permute(1, 0, 2).contiguous().view(G * n_local, Kg)
    xs = torch.empty(plan.n, Kg, dtype=x_local.dtype, device=x_local.device)
    _A2A_KIND[0] = "sliced"
    try:
        dist.all_to_all_single(xs, send, rows, [n_local] * G, group=group)
        b = None if bias is None else bias[plan.rank * Kg:(plan.rank + 1) * Kg].contiguous()
        ys = _agg(full, xs, bias=b, relu=relu)                                 # all N rows of this rank's column slice
        back = torch.empty(G * n_local, Kg, dtype=x_local.dtype, device=x_local.device)
        dist.all_to_all_single(back, ys.contiguous(), [n_local] * G, rows, group=group)
    finally:
        _A2A_KIND[0] = "halo"
    return back.view(G, n_local, Kg).permute(1, 0, 2).reshape(n_local, K)


class _SlicedAggregate(torch.autograd.Function):
    """``ShardedAdj.aggregate`` in column-sliced form (see ``_sliced_product``); the backward is the same exchange pattern on the
    transposed full matrix."""

    @staticmethod
    def forward(ctx, x_local, sadj, full, bias=None, relu=False):
        ctx.sadj, ctx.full, ctx.has_bias = sadj, full, bias is not None
        return _sliced_product(x_local.contiguous(), sadj, full, bias, relu)

    @staticmethod
    def backward(ctx, g_y):
        g_y = g_y.contiguous()
        g_x = _sliced_product(g_y, ctx.sadj, ctx.full.t())
        g_b = ops.colsum(g_y) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        return g_x, None, None, g_b, None


class _ShardedSageLayer(torch.autograd.Function):
    """``ops._SageLayer`` on a node-range shard: SAGEConv (``lin_l(aggr_j x_j) + lin_r(x_i)``, gnn.py:79-84) as ONE autograd node whose
    aggregation is the overlapped halo-exchange form above -- ``lin_r(x)`` is the addend of the GEMM / aggregation store, the input
    gradient's second path ``g W_r`` the addend of the backward aggregation's first piece (no element-wise pass over [n, C] in either
    direction); ``narrow`` (out < in): ``x W_l^T`` is aggregated instead of x, the halo then carries `out` floats per row."""

    @staticmethod
    def forward(ctx, x, sadj, a_own, a_halo, static_halo, wl, bl, wr, narrow):
        ctx.tap_box = getattr(x, "_egnn_tap", None)
        x = x.contiguous()
        r = ops.gemm_raw(x, wr, False, True)                                   # lin_r(x)
        if narrow:
            t = ops.gemm_raw(x, wl, False, True)                               # x W_l^T: aggregated with bias + lin_r(x) in the stores
            out = _overlap_forward(t, sadj, a_own, a_halo, None, bias=bl, addend=r)
            agg = None
        else:
            agg = _overlap_forward(x, sadj, a_own, a_halo, static_halo)
            out = ops.gemm_raw(agg, wl, False, True, bias=bl, addend=r)        # lin_l(agg) + lin_r(x) in one store
        ctx.save_for_backward(x, wl, wr, *([] if agg is None else [agg]))
        ctx.sadj, ctx.pieces, ctx.narrow, ctx.has_bias = sadj, (a_own, a_halo), narrow, bl is not None
        ctx.static_input = static_halo is not None      # its forward made no exchange: neither does its backward, on any rank
        return out

    @staticmethod
    def backward(ctx, g):
        x, wl, wr = ctx.saved_tensors[:3]
        agg = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        a_own, a_halo = ctx.pieces
        g = g.contiguous()
        need_x = ctx.needs_input_grad[0]
        gx = gwl = gbl = gwr = None
        if ctx.needs_input_grad[7]:
            gwr = ops.gemm_raw(g, x, True, False)                              # dW_r = g^T x
        if ctx.has_bias and ctx.needs_input_grad[6]:
            gbl = ops.colsum(g)
        # EVERY rank runs the backward exchange (a collective), whether or not its own input needs a gradient
        if ctx.narrow:
            dt = _overlap_backward(g, ctx.sadj, a_own, a_halo)
            if ctx.needs_input_grad[5]:
                gwl = ops.gemm_raw(dt, x, True, False)                         # dW_l = dt^T x
            if need_x:
                gx = ops.gemm_raw(dt, wl, False, False, addend=ops.gemm_raw(g, wr, False, False))     # dt W_l + g W_r
        else:
            if ctx.needs_input_grad[5]:
                gwl = ops.gemm_raw(g, agg, True, False)                        # dW_l = g^T agg
            # a non-static input exchanged its halo in the forward on EVERY rank, so every rank runs the reverse exchange here -- also one
            # whose own input needs no gradient (sage_layer() only hands such inputs over when they are static; kept collective-safe anyway)
            if need_x or not ctx.static_input:
                gx = _overlap_backward(ops.gemm_raw(g, wl, False, False), ctx.sadj, a_own, a_halo, addend=ops.gemm_raw(g, wr, False, False))
                gx = gx if need_x else None
        return ops._fresh(gx, ctx.tap_box), None, None, None, None, gwl, gbl, gwr, None


class ShardedAdj:
    """What the convs receive as ``adj_t`` on a sharded run: the rank's rows of A (and of A^ = gcn_norm(A)).

    Built from the rank's OWN rows only (``rowptr_local`` starting at 0, global column ids); the halo lists come from
    ``ShardPlan.from_local`` (a collective) and, for A^, the degrees of the halo nodes from one exchange of a vector."""

    def __init__(self, rowptr_local: Tensor, col_global: Tensor, n: int, world: int, rank: int, device, group=None,
                 with_gcn: bool = True, _plan: ShardPlan | None = None):
        self.group, self.device = group, torch.device(device)
        collective = world > 1 and dist.is_available() and dist.is_initialized()
        make = (lambda rp, c, v: ShardPlan.from_local(rp, c, v, n, world, rank, group)) if collective or world == 1 else None
        if _plan is not None:
            self.plan = _plan
        elif make is not None:
            self.plan = make(rowptr_local.to(self.device), col_global.to(self.device), None)
        else:
            raise RuntimeError("ShardedAdj with world > 1 needs an initialised process group (or a plan built by ShardPlan.from_global)")
        self._finish()
        self._gcn = None
        if with_gcn and make is None:
            # A^ needs deg^-1/2 of the halo nodes from their owners (one exchange) and its own send lists: both are collectives
            raise RuntimeError("ShardedAdj(_plan=..., with_gcn=True) with world > 1 needs an initialised process group for the "
                               "normalised adjacency; pass with_gcn=False for the raw adjacency only")
        if with_gcn:
            self._gcn = ShardedAdj.__new__(ShardedAdj)
            self._gcn.group, self._gcn.device = group, self.device
            self._gcn.plan = self._gcn_plan(rowptr_local.to(self.device), col_global.to(self.device), n, world, rank, group, make)
            self._gcn._finish()
            self._gcn._gcn = None

    def _finish(self):
        dev = self.device
        self.send_idx_dev = self.plan.send_idx.to(dev)
        self.raw = self.plan.local_adj(dev)
        self.scatter = self.plan.scatter_adj(dev)
        self._pieces = {}
        self._static = None
        self._full = {}
        plan = self.plan
        self.rows_per_rank = [node_range(plan.n, plan.world, r)[1] - node_range(plan.n, plan.world, r)[0] for r in range(plan.world)]
        # the mode decision must be the same on every rank: the all-rank MEAN halo (one small all_reduce at construction)
        self.mean_halo = float(plan.n_halo)
        if plan.world > 1 and dist.is_available() and dist.is_initialized():
            t = torch.tensor([float(plan.n_halo)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=self.group)
            self.mean_halo = float(t.item()) / plan.world
        self.agg_mode = _AGG_MODE

    def sliced_pays(self, K: int) -> bool:
        """True when the column-sliced exchange moves fewer bytes than the halo exchange for a K-wide aggregation:
        mean halo rows x K  >  2 N K (world - 1) / world^2 (there and back), and the slices are kernel-aligned (K % (4 world) == 0)."""
        plan = self.plan
        G = plan.world
        if G < 2 or K % (4 * G):
            return False
        if self.agg_mode == "sliced":
            return True
        return self.agg_mode == "auto" and self.mean_halo * G * G > 2.0 * plan.n * (G - 1)

    def full_adj(self, mean: bool, valueless: bool) -> SparseTensor:
        """The FULL [N, N] aggregation matrix (every rank's rows, global column ids, the values the pieces of ``_split`` carry), built
        once per kind by all-gathering the shards' CSR pieces -- what the column-sliced form aggregates on.  COLLECTIVE (padded
        all_gather_into_tensor of row counts, columns and values); the first sliced aggregation of a kind triggers it on every rank
        at the same program point because the mode decision is global."""
        key = (mean, valueless)
        if key in self._full:
            return self._full[key]
        plan, dev, G = self.plan, self.device, self.plan.world
        rp = plan.rowptr_local.to(dev)
        cnt = (rp[1:] - rp[:-1])
        halo_ids = plan.halo_ids.to(dev)
        ce = plan.col_ext.to(dev)
        col_global = torch.where(plan.remote_mask.to(dev), halo_ids[(ce - plan.n_local).clamp(min=0)] if plan.n_halo else ce, ce + plan.lo)
        val = None if valueless else (None if plan.value_local is None else plan.value_local.to(dev))
        if mean:
            inv = torch.repeat_interleave(1.0 / cnt.clamp(min=1).to(torch.float32), cnt)
            val = inv if val is None else val * inv

        def gather_padded(t, width):
            pad = torch.zeros(width, dtype=t.dtype, device=dev)
            pad[:t.numel()] = t
            out = torch.empty(G * width, dtype=t.dtype, device=dev)
            dist.all_gather_into_tensor(out, pad, group=self.group)
            return out.view(G, width)
        per = plan.per
        counts_all = gather_padded(cnt, per)                                       # [G, per] row lengths (padding rows: 0)
        nnz = torch.tensor([int(cnt.sum())], dtype=torch.int64, device=dev)
        nnz_all = torch.empty(G, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(nnz_all, nnz, group=self.group)
        nnz_list = [int(v) for v in nnz_all.tolist()]
        width = max(max(nnz_list), 1)
        cols = gather_padded(col_global, width)
        vals = None if val is None else gather_padded(val.to(torch.float32), width)
        counts = torch.cat([counts_all[r, :self.rows_per_rank[r]] for r in range(G)])
        rowptr = torch.zeros(plan.n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=rowptr[1:])
        col = torch.cat([cols[r, :nnz_list[r]] for r in range(G)]).contiguous()
        value = None if vals is None else torch.cat([vals[r, :nnz_list[r]] for r in range(G)]).contiguous()
        full = SparseTensor(rowptr=rowptr, col=col, value=value, sparse_sizes=(plan.n, plan.n))
        self._full[key] = full
        return full

    @staticmethod
    def _gcn_plan(rowptr_local, col_global, n, world, rank, group, make):
        """Rows of A^ = D^-1/2 (A + I) D^-1/2 for this shard (PyG gcn_norm, SparseTensor branch: existing diagonal removed, one
        loop per node, deg = row sums of ones).  Self loops are own columns, so the halo set is that of A; the only remote
        quantity is deg^-1/2 of the halo nodes: one exchange of a [n_local] vector."""
        lo, hi, _ = node_range(n, world, rank)
        n_local = hi - lo
        dev = col_global.device
        rows = torch.repeat_interleave(torch.arange(n_local, device=dev), rowptr_local[1:] - rowptr_local[:-1])
        keep = col_global != rows + lo
        loops = torch.arange(n_local, device=dev)
        r2, c2 = torch.cat([rows[keep], loops]), torch.cat([col_global[keep], loops + lo])
        order = torch.argsort(r2 * n + c2, stable=True)
        r2, c2 = r2[order], c2[order]
        rp = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(r2, minlength=n_local), 0, out=rp[1:])
        plan = make(rp, c2, None)
        dinv = (rp[1:] - rp[:-1]).to(torch.float32).pow(-0.5)
        dinv.masked_fill_(dinv == float("inf"), 0.0)
        dinv_ext = torch.empty(n_local + plan.n_halo, dtype=torch.float32, device=dev)
        dinv_ext[:n_local] = dinv
        if plan.world > 1 and dist.is_initialized():
            dist.all_to_all_single(dinv_ext[n_local:], dinv.index_select(0, plan.send_idx).contiguous(), plan.recv_counts, plan.send_counts,
                                   group=group)
        plan.value_local = ((1.0 * dinv[r2]) * dinv_ext[plan.col_ext]).contiguous()      # (1 * dinv[row]) * dinv[col], as PyG forms it
        return plan

    def gcn_normalized(self) -> "ShardedAdj":
        if self._gcn is None:
            raise RuntimeError("ShardedAdj was built without the normalised adjacency (with_gcn=False)")
        return self._gcn

    def register_static(self, x_local: Tensor) -> None:
        """Declare ``x_local`` (the input node features: constant for the whole run) static: its halo rows are fetched
        from the peers ONCE and kept next to the local rows, as a partitioned graph store keeps the features of a
        partition's halo nodes.  Every later aggregation of exactly this tensor skips the exchange (2 of the 8 per
        epoch, 22 % of the halo volume of the GCN run).  Collective: all ranks must register."""
        x_ext = _HaloExchange.apply(x_local.detach(), self)
        self._static = (x_local, x_ext, x_local._version)
        if self._gcn is not None:
            self._gcn.register_static(x_local)

    def _split(self, mean: bool, valueless: bool):
        """(own-column piece, halo-column piece) of the aggregation matrix actually applied: A^ values, or 1 / rowcount per
        entry for ``mean`` (sum of the two pieces = the mean over all of the row's entries), or no values (sum)."""
        key = (mean, valueless)
        if key not in self._pieces:
            val = None if valueless else self.plan.value_local
            if mean:
                cnt = (self.plan.rowptr_local[1:] - self.plan.rowptr_local[:-1]).clamp(min=1).to(torch.float32)
                inv = torch.repeat_interleave(1.0 / cnt, self.plan.rowptr_local[1:] - self.plan.rowptr_local[:-1])
                val = inv if val is None else val * inv
                own, halo = self.plan.split_adj(self.device, value=val)
            elif val is None:
                saved, self.plan.value_local = self.plan.value_local, None
                own, halo = self.plan.split_adj(self.device)
                self.plan.value_local = saved
            else:
                own, halo = self.plan.split_adj(self.device)
            self._pieces[key] = (own, halo)
        return self._pieces[key]

    def aggregate(self, x_local: Tensor, reduce: str, valueless: bool = False, bias: Tensor | None = None, relu: bool = False) -> Tensor:
        """``bias`` [K]: added to every row in the aggregation's store (GCNConv's bias: no separate pass over [n, K]); ``relu``
        (inference only): clamp in the same store (the eval-mode BatchNorm fold of GCNConv)."""
        if reduce not in ("sum", "add", "mean"):
            raise NotImplementedError(f"sharded aggregation supports sum / mean, not '{reduce}'")
        st = self._static
        static = st is not None and st[0] is x_local and not x_local.requires_grad and x_local._version == st[2]   # version AT registration
        if not static and self.sliced_pays(x_local.shape[1]):      # (a registered static input travels once per run: nothing to save)
            return _SlicedAggregate.apply(x_local, self, self.full_adj(reduce == "mean", valueless), bias, relu)
        if _OVERLAP:
            own, halo = self._split(reduce == "mean", valueless)
            return _OverlapAggregate.apply(x_local, self, own, halo, st[1][self.plan.n_local:] if static else None, bias, relu)
        x_ext = st[1] if static else _HaloExchange.apply(x_local, self)
        adj = self.raw.set_value(None) if valueless and self.raw.has_value() else self.raw
        if relu:
            return ops.spmm_raw(adj, x_ext, reduce, bias=bias, relu=True)[0]
        return ops.spmm(adj, x_ext, reduce, bias=bias)

    def sage_layer(self, x_local: Tensor, lin_l, lin_r, reduce: str, narrow: bool):
        """SAGEConv on this shard as one autograd node (``_ShardedSageLayer``); None when the fused form does not apply (blocking
        exchange mode, CPU stand-ins, a layer whose input needs no gradient and is not the registered static input -- its backward
        exchange would be skipped by autograd on this rank only)."""
        if not (_OVERLAP and x_local.is_cuda and reduce in ("sum", "add", "mean") and lin_r.bias is None and self.plan.n_local > 0):
            return None
        width = lin_l.weight.shape[0] if narrow else x_local.shape[1]
        if self.sliced_pays(width) and not (self._static is not None and self._static[0] is x_local):
            return None                       # column-sliced mode: SAGEConv composes ``aggregate`` (sliced) with its GEMMs
        st = self._static
        static = st is not None and st[0] is x_local and not x_local.requires_grad and x_local._version == st[2]
        if torch.is_grad_enabled() and not (static or x_local.requires_grad):
            return None
        if static and narrow:
            return None                       # the narrow form aggregates x W_l^T: the static halo of x does not apply
        own, halo = self._split(reduce == "mean", True)
        return _ShardedSageLayer.apply(x_local, self, own, halo, st[1][self.plan.n_local:] if static else None, lin_l.weight, lin_l.bias,
                                       lin_r.weight, narrow)

    def halo_fraction(self) -> float:
        return self.plan.n_halo / max(1, self.plan.n - self.plan.n_local)


# ------------------------------------------------------------------------------------------------
# BatchNorm with global (all-rank) batch statistics
# ------------------------------------------------------------------------------------------------
class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        n_local = torch.tensor([float(x.shape[0])], dtype=torch.float32, device=x.device)
        s = torch.cat([x.sum(0), n_local])
        dist.all_reduce(s, group=group)
        n = s[-1]
        mean = s[:-1] / n
        xc = x - mean
        ss = (xc * xc).sum(0)
        dist.all_reduce(ss, group=group)
        var = ss / n
        rstd = torch.rsqrt(var + eps)
        xhat = xc * rstd
        ctx.save_for_backward(xhat, weight, rstd, n)
        ctx.group = group
        return xhat * weight + bias, mean, var, n

    @staticmethod
    def backward(ctx, g, _gm, _gv, _gn):
        xhat, weight, rstd, n = ctx.saved_tensors
        gw_local = (g * xhat).sum(0)
        gb_local = g.sum(0)
        tot = torch.cat([gb_local, gw_local])
        dist.all_reduce(tot, group=ctx.group)
        C = gb_local.numel()
        sg, sgx = tot[:C], tot[C:]
        gx = (weight * rstd) * (g - sg / n - xhat * (sgx / n))
        return gx, gw_local, gb_local, None, None  # parameter grads stay local: the flat all-reduce sums them


class SyncBatchNorm1d(nn.Module):
    """Drop-in for ``torch.nn.BatchNorm1d`` (same parameter / buffer names) whose training statistics span all ranks."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1, group=None):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.group = num_features, eps, momentum, group
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def reset_parameters(self):
        with torch.no_grad():
            self.weight.fill_(1.0)
            self.bias.zero_()
            self.running_mean.zero_()
            self.running_var.fill_(1.0)
            self.num_batches_tracked.zero_()

    def _update_running(self, mean, var, n):
        with torch.no_grad():
            if mean.is_cuda and self.running_mean.dtype == torch.float32 and self.num_batches_tracked.dtype == torch.int64:
                # one launch (the row count stays on the device) instead of nine element-wise ones
                lib = _lib.load()
                _lib.check(lib.egnn_bn_running_update_dev_f32(_lib.ptr(mean), _lib.ptr(var), mean.numel(), _lib.ptr(n), float(self.momentum),
                                                              _lib.ptr(self.running_mean), _lib.ptr(self.running_var),
                                                              _lib.ptr(self.num_batches_tracked), _lib.stream()), "egnn_bn_running_update_dev_f32")
                return
            m = self.momentum
            self.running_mean.mul_(1 - m).add_(mean.detach(), alpha=m)
            self.running_var.mul_(1 - m).add_(var.detach() * (n / (n - 1).clamp(min=1)), alpha=m)
            self.num_batches_tracked += 1

    def forward(self, x: Tensor) -> Tensor:
        if not self.training:
            return (x - self.running_mean) * torch.rsqrt(self.running_var + self.eps) * self.weight + self.bias
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps, self.group)
        self._update_running(mean, var, n)
        return y

    def fused_act(self, x: Tensor, relu: bool, p: float, training: bool, pick: Tensor | None = None) -> Tensor:
        """dropout(relu(self(x)), p) -- on the GPU through the fused BN + ReLU + dropout kernels (ops.sync_bn_act /
        ops.bn_act), elsewhere (gloo tests) through the torch operators above.  ``pick`` (unique row ids): only those rows of the
        result are formed; the statistics still span every row of every rank."""
        _lib.require_gpu(x)
        if not ops.bn_shape_ok(x):       # a width the fused kernels do not take (C % 4 != 0, C > 1024): the torch operators above
            y = self(x)
            y = torch.relu(y) if relu else y
            y = torch.nn.functional.dropout(y, p, training) if p > 0 else y
            return y if pick is None else y[pick]
        if not training:   # running statistics: the single-GPU kernel applies as it is
            y = ops._BnAct.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, relu, 0.0, 0, False)
            return y if pick is None else y[pick]
        y, mean, var, n = ops.sync_bn_act(x, self, relu, p, training, self.group, pick)
        self._update_running(mean, var, n)
        return y

    def fused_act_linear(self, x: Tensor, w: Tensor, relu: bool, p: float, training: bool):
        """(h, h @ w) for h = dropout(relu(self(x)), p) and a narrow ``w`` (the output conv's weight): the students' last hidden layer
        in one forward pass and a two-half backward around ONE all-reduce (ops.sync_bn_act_linear).  None when not taken."""
        _lib.require_gpu(x)
        if not (training and ops.bn_shape_ok(x)):
            return None
        both = ops.sync_bn_act_linear(x, self, w, relu, p, training, self.group)
        if both is None:
            return None
        h, xw, mean, var, n = both
        self._update_running(mean, var, n)
        return h, xw


# ------------------------------------------------------------------------------------------------
# G-CRD across ranks: row block of Z per rank
# ------------------------------------------------------------------------------------------------
def _all_gather_rows(x: Tensor, counts: list[int], group) -> Tensor:
    """Concatenate per-rank row blocks of differing length (padded all_gather_into_tensor)."""
    world = len(counts)
    mx = max(counts)
    pad = torch.zeros(mx, x.shape[1], dtype=x.dtype, device=x.device)
    pad[:x.shape[0]] = x
    out = torch.empty(world * mx, x.shape[1], dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[p * mx:p * mx + c] for p, c in enumerate(counts)], dim=0)


class _DistNCE(torch.autograd.Function):
    """fhat_local, that_local: this rank's sampled unit rows; returns the GLOBAL mean loss (identical on every rank)."""

    @staticmethod
    def forward(ctx, fhat, that, tau, counts, rank, group):
        fhat, that = fhat.contiguous(), that.contiguous()
        S = int(sum(counts))
        off = int(sum(counts[:rank]))
        t_all = _all_gather_rows(that, counts, group)
        Sr, P = fhat.shape
        loss = torch.zeros(1, dtype=torch.float32, device=fhat.device)
        Z = lse = None
        if Sr > 0:
            Z, lse, loss = ops.nce_block_fwd(fhat, t_all, off, tau, 1.0 / S)
        dist.all_reduce(loss, group=group)
        ctx.save_for_backward(fhat, t_all, *([Z, lse] if Sr > 0 else []))
        ctx.meta = (tau, counts, rank, group, S, off)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        tau, counts, rank, group, S, off = ctx.meta
        saved = ctx.saved_tensors
        fhat, t_all = saved[0], saved[1]
        Sr, P = fhat.shape
        if Sr > 0:
            Z, lse = saved[2], saved[3]
            df, dt_all = ops.nce_block_bwd(fhat, t_all, off, 1.0 / (S * tau), Z, lse, g.contiguous().to(torch.float32), tau)
        else:   # no row block here: zeros that still take part in the collective
            dt_all, df = torch.zeros_like(t_all), torch.zeros_like(fhat)
        dist.all_reduce(dt_all, group=group)  # every row block contributes to every teacher row
        return df, dt_all[off:off + Sr].contiguous(), None, None, None, None


class _GatherSampledRows(torch.autograd.Function):
    """This rank's sampled rows -> the [S, P] matrix of ALL ranks' sampled rows (rank blocks in order).  The consumer (GSP)
    evaluates the full all-pairs loss on every rank -- S <= 4096 rows: milliseconds, no row-block kernel needed -- so every
    rank holds the complete gradient of the GLOBAL loss; the backward hands back the rows of this rank's block, and the
    gradient all-reduce of the parameters adds the ranks' row blocks up to the full gradient (no extra collective)."""

    @staticmethod
    def forward(ctx, x, counts, rank, group):
        ctx.meta = (int(sum(counts[:rank])), x.shape[0])
        return _all_gather_rows(x.contiguous(), counts, group)

    @staticmethod
    def backward(ctx, g):
        off, n = ctx.meta
        return g[off:off + n].contiguous(), None, None, None


# ------------------------------------------------------------------------------------------------
# the sampled criteria (G-CRD / GSP) in STATIC shapes: capturable into a hipGraph at any world size
# ------------------------------------------------------------------------------------------------
class StaticSample:
    """The per-step row sample of criterion.py:62-65,134-137 laid out in shapes that do not depend on the draw.

    The draw itself is unchanged: ONE ``np.random.choice(n_train, S, replace=False)`` over the global train list, the same on
    every rank.  How many of the S rows a rank owns is a hypergeometric count that moves from step to step; here every rank
    contributes a block of ``cap`` rows (its owned sampled rows first, in draw order, then rows that no consumer reads) to an
    all-gather, and ``perm`` [S] picks the S valid rows out of the [world * cap] gathered ones IN DRAW ORDER.  cap = the
    largest rank's expected count + ``sigmas`` standard deviations (never more than that rank owns); a draw that does not fit
    (probability ~1e-9 per step at 6 sigma) is reported by ``fill`` and the caller runs that step with the dynamic shapes.
    ``idx`` [m] (m = min(cap, rows this rank owns), fixed): local train positions to form -- the owned sampled rows, then DISTINCT
    unsampled rows (distinct: the row-gather kernels' backward assumes unique ids; they receive a zero gradient)."""

    def __init__(self, prob: "ShardedProblem", S: int, sigmas: float = 6.0):
        ntr, world = prob.n_train_global, prob.world
        self.S, self.world, self.rank = min(int(S), ntr), world, prob.rank
        per_rank = torch.bincount(prob.train_owner, minlength=world)
        n_max = int(per_rank.max())
        p = n_max / max(ntr, 1)
        # hypergeometric variance <= the binomial's S p (1 - p)
        cap = int(np.ceil(self.S * p + sigmas * np.sqrt(max(self.S * p * (1.0 - p), 0.0)))) + 1
        self.cap = max(1, min(cap, n_max, self.S))
        self.n_mine = int(per_rank[prob.rank])
        self.m = min(self.cap, self.n_mine)
        self._owner, self._localpos = prob.train_owner.numpy(), prob.train_localpos.numpy()
        dev = prob.device
        self.idx_host = torch.zeros(max(self.m, 1), dtype=torch.int64)
        self.perm_host = torch.zeros(self.S, dtype=torch.int64)
        if torch.device(dev).type == "cuda":
            self.idx_host, self.perm_host = self.idx_host.pin_memory(), self.perm_host.pin_memory()
        self.idx_dev = torch.zeros(max(self.m, 1), dtype=torch.int64, device=dev)
        self.perm_dev = torch.zeros(self.S, dtype=torch.int64, device=dev)
        self.counts = [0] * world
        self.external = False     # True: somebody else (ShardedGraphedEpoch) draws / fills / uploads before each step

    def draw(self) -> np.ndarray:
        ntr = self._owner.shape[0]
        return np.random.choice(ntr, self.S, replace=False) if self.S < ntr else np.arange(ntr)

    def fill(self, pick: np.ndarray) -> bool:
        """Host side of one step: the padded id list of this rank and the permutation, from the draw.  False = the draw does not
        fit ``cap`` (nothing is written)."""
        owner = self._owner[pick]
        counts = np.bincount(owner, minlength=self.world)
        if counts.max() > self.cap:
            return False
        order = np.argsort(owner, kind="stable")
        starts = np.cumsum(counts) - counts
        k = np.empty(self.S, dtype=np.int64)
        k[order] = np.arange(self.S) - starts[owner[order]]
        self.perm_host.copy_(torch.from_numpy(owner.astype(np.int64) * self.cap + k))
        if self.m > 0:
            mine = self._localpos[pick[owner == self.rank]]
            taken = np.zeros(self.n_mine, dtype=bool)
            taken[mine] = True
            fillers = np.flatnonzero(~taken)[:self.m - mine.shape[0]]
            self.idx_host.copy_(torch.from_numpy(np.concatenate([mine, fillers]).astype(np.int64)))
        self.counts = counts.tolist()
        return True

    def upload(self):
        self.idx_dev.copy_(self.idx_host, non_blocking=True)
        self.perm_dev.copy_(self.perm_host, non_blocking=True)


class _GatherPadded(torch.autograd.Function):
    """This rank's block x [m, Q] (zero-padded to [cap, Q]) -> the S sampled rows of ALL ranks in draw order (``perm`` into the
    [world * cap, Q] all-gather).  backward: the [S, Q] gradient is spread back over the padded layout and -- ``reduce_grad`` --
    summed over the ranks (each rank evaluated only its row block of the loss: G-CRD), then this rank's block is returned."""

    @staticmethod
    def forward(ctx, x, perm, cap, rank, world, reduce_grad, group):
        m, Q = x.shape
        if m == cap:                       # a full block (one rank: always): nothing to pad
            pad = x.contiguous()
        else:
            pad = x.new_zeros(cap, Q)
            pad[:m].copy_(x)
        if world > 1:
            allx = torch.empty(world * cap, Q, dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(allx, pad, group=group)
        else:
            allx = pad
        ctx.save_for_backward(perm)
        ctx.meta = (m, cap, rank, world, reduce_grad, group)
        return allx.index_select(0, perm)

    @staticmethod
    def backward(ctx, g):
        (perm,) = ctx.saved_tensors
        m, cap, rank, world, reduce_grad, group = ctx.meta
        # perm holds unique positions; when it addresses EVERY padded row (one rank: cap == S) no row is left to zero
        gp = g.new_empty(world * cap, g.shape[1]) if perm.numel() == world * cap else g.new_zeros(world * cap, g.shape[1])
        gp.index_copy_(0, perm, g.contiguous())
        if reduce_grad and world > 1:
            from . import hostcomm
            # every rank only needs ITS block of the sum: reduce_scatter where the transport has one (RCCL; hostcomm for the tensors it
            # stages -- gloo itself has none: host tensors take the all_reduce below)
            if dist.get_backend(group) == "nccl" or (hostcomm.active() and hostcomm._staged(group, gp)):
                mine = torch.empty(cap, g.shape[1], dtype=g.dtype, device=g.device)
                dist.reduce_scatter_tensor(mine, gp, group=group)
                return mine[:m], None, None, None, None, None, None
            dist.all_reduce(gp, group=group)
        return gp[rank * cap:rank * cap + m].contiguous(), None, None, None, None, None, None


class _BalancedNCE(torch.autograd.Function):
    """G-CRD (criterion.py:129-149) on the gathered sample: every rank evaluates rows [r0, r1) = its EVEN share of the S x S
    problem on the MFMA kernels (fixed shapes: ceil(S / world) rows, diagonal offset r0), whoever owns those rows; the loss terms
    are all-reduced, the partial gradients (its rows of dfhat, its contribution to every row of dthat) are summed over the
    ranks by the gather's backward."""

    @staticmethod
    def forward(ctx, fhat_all, t_all, tau, rank, world, group):
        fhat_all, t_all = fhat_all.contiguous(), t_all.contiguous()
        S = fhat_all.shape[0]
        per = (S + world - 1) // world
        r0 = min(rank * per, S)
        r1 = min(r0 + per, S)
        saved = [fhat_all, t_all]
        if r1 > r0:
            Z, lse, loss = ops.nce_block_fwd(fhat_all[r0:r1], t_all, r0, tau, 1.0 / S)
            saved += [Z, lse]
        else:
            loss = torch.zeros(1, dtype=torch.float32, device=fhat_all.device)
        if world > 1:
            dist.all_reduce(loss, group=group)
        ctx.save_for_backward(*saved)
        ctx.meta = (tau, r0, r1, S)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        tau, r0, r1, S = ctx.meta
        saved = ctx.saved_tensors
        fhat_all, t_all = saved[0], saved[1]
        if r1 > r0:
            df, dt_all = ops.nce_block_bwd(fhat_all[r0:r1], t_all, r0, 1.0 / (S * tau), saved[2], saved[3], g.contiguous().to(torch.float32), tau)
            if r0 == 0 and r1 == S:        # one rank: the block is the whole problem
                df_all = df
            else:
                df_all = torch.zeros_like(fhat_all)
                df_all[r0:r1].copy_(df)
        else:
            df_all, dt_all = torch.zeros_like(fhat_all), torch.zeros_like(t_all)
        return df_all, dt_all, None, None, None, None


def _static_sample_ready(prob: "ShardedProblem") -> bool:
    """True when this step's sampled criterion runs in the static shapes.  Eager steps make the step's ONE host draw here (the
    point where the dynamic path makes it) and fall back to the dynamic shapes -- with the same draw -- when it does not fit."""
    ss = prob.static_sample
    if ss is None:
        return False
    if ss.external:
        return True
    pick = ss.draw()
    if ss.fill(pick):
        ss.upload()
        return True
    prob._forced_pick = pick
    return False


def _static_sampled_pair(prob: "ShardedProblem", f: Tensor, t: Tensor, normalize: bool, reduce_grad: bool):
    """(student rows, teacher rows) [S, P] of the global sample in draw order, identical on every rank, from this rank's projected
    rows f / t [m, P] (the rows ``static_sample.idx_dev[:m]`` of the heads' outputs, already picked) -- ONE all-gather of the
    [cap, Ps + Pt] block."""
    ss = prob.static_sample
    if ss.m > 0:
        fs = ops.gather_normalize(f, None) if normalize else f
        ts = ops.gather_normalize(t, None) if normalize else t
        block = torch.cat([fs, ts], dim=1)
    else:   # a rank without train rows: an empty block that stays attached to both heads' graphs
        block = torch.cat([f[:0], t[:0]], dim=1)
    both = _GatherPadded.apply(block, ss.perm_dev, ss.cap, prob.rank, prob.world, reduce_grad, prob.group)
    Ps = f.shape[1]
    return both[:, :Ps], both[:, Ps:]


class _TrainSubgraph:
    """LSP on shards (criterion.py:95-126 over ``subgraph(train_idx, edge_index)``, gnn.py:246-250): this rank's slice of the
    train-node subgraph = the entries (row r, column c) of its own rows with r and c both train nodes.  The softmax groups
    (edges sharing ``dst``) are whole on the owner of ``dst``; the ``src`` features of remote train neighbours come through
    the same halo exchange as the convs' (own plan: only train neighbours travel).  Built once per problem (collective)."""

    def __init__(self, prob: "ShardedProblem", rowptr_local: Tensor, col_global: Tensor, train_global: Tensor):
        n, world, rank, dev = prob.n, prob.world, prob.rank, prob.device
        is_train = torch.zeros(n, dtype=torch.bool)
        is_train[train_global] = True
        n_local = prob.hi - prob.lo
        rows = torch.repeat_interleave(torch.arange(n_local), rowptr_local[1:] - rowptr_local[:-1])
        keep = is_train[rows + prob.lo] & is_train[col_global]
        rows_k, cols_k = rows[keep], col_global[keep]
        rp = torch.zeros(n_local + 1, dtype=torch.int64)
        rp[1:] = torch.cumsum(torch.bincount(rows_k, minlength=n_local), 0)
        self.sadj = ShardedAdj(rp, cols_k, n, world, rank, dev, prob.group, with_gcn=False)
        pl = self.sadj.plan
        # edge (src = neighbour in extended numbering, dst = the local row): groups by dst, as the reference's softmax does
        self.edge_index = torch.stack([pl.col_ext.to(device=dev, dtype=torch.int64), rows_k.to(device=dev, dtype=torch.int64)])
        e = torch.tensor([float(rows_k.numel())], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(e, group=prob.group)
        self.e_local, self.e_global = int(rows_k.numel()), int(e.item())
        self.teacher_ext = None

    def extend(self, x_local: Tensor) -> Tensor:
        return _HaloExchange.apply(x_local, self.sadj)


# ------------------------------------------------------------------------------------------------
# sharded problem + train / eval step
# ------------------------------------------------------------------------------------------------
def swap_batchnorm(module: nn.Module, group=None) -> nn.Module:
    """Replace every BatchNorm1d by SyncBatchNorm1d (state is copied; names unchanged)."""
    for name, child in list(module.named_children()):
        if isinstance(child, nn.BatchNorm1d):
            sb = SyncBatchNorm1d(child.num_features, child.eps, child.momentum if child.momentum is not None else 0.1, group)
            sb.load_state_dict(child.state_dict())
            setattr(module, name, sb.to(child.weight.device))
        else:
            swap_batchnorm(child, group)
    return module


class ShardedProblem:
    """Everything one rank holds: its rows of x / y / teacher artefacts, local train / eval index sets."""

    def __init__(self, data, world: int, rank: int, device, group=None, need_gcn: bool = True):
        n = data.num_nodes
        self.world, self.rank, self.group, self.device = world, rank, group, device
        lo, hi, _ = node_range(n, world, rank)
        self.lo, self.hi, self.n = lo, hi, n
        # this rank's rows only (a data loader would read just these; the synthetic generator makes the whole graph on the
        # host of every rank, deterministically, and the slice is taken here)
        rowptr, col, _ = data.adj_t.csr()
        e0, e1 = int(rowptr[lo]), int(rowptr[hi])
        self.adj = ShardedAdj((rowptr[lo:hi + 1] - e0), col[e0:e1], n, world, rank, device, group, with_gcn=need_gcn)
        self._rows_cpu = ((rowptr[lo:hi + 1] - e0).clone(), col[e0:e1].clone(), data.split_idx["train"].clone())   # for the LSP subgraph plan
        self._train_sub = None
        self.sample_hook = None          # ShardedGraphedEpoch: static device buffer instead of the per-step host draw + upload
        self.static_sample = None        # StaticSample: the sampled criteria run in draw-independent shapes
        self._forced_pick = None
        self._zero = None
        self.x = data.x[lo:hi].to(device)
        self.adj.register_static(self.x)          # input features never change: halo copy fetched once
        self.y = data.y[lo:hi].to(device)
        self.teacher_out_feat = data.teacher_out_feat[lo:hi].to(device) if getattr(data, "teacher_out_feat", None) is not None else None
        if self.teacher_out_feat is not None and self.teacher_out_feat.is_cuda:
            self.teacher_out_feat = ops.pad_pitch(self.teacher_out_feat)   # 750 floats per row -> 16-byte aligned rows
        self.teacher_logits = data.teacher_logits[lo:hi].to(device) if getattr(data, "teacher_logits", None) is not None else None
        # global train order (gnn.py:243-244) restricted to my range; positions keep the global order
        tr = data.split_idx["train"]
        self.n_train_global = tr.numel()
        mine = (tr >= lo) & (tr < hi)
        self.train_pos = torch.nonzero(mine).view(-1)                 # positions in the global train list
        self.train_local = (tr[mine] - lo).to(device)                 # local row ids
        owner = torch.div(tr, (n + world - 1) // world, rounding_mode="floor")
        self.train_owner = owner                                       # CPU, per global train position
        localpos = torch.zeros_like(tr)
        localpos[mine] = torch.arange(int(mine.sum()))
        self.train_localpos = localpos                                 # valid where owner == rank
        self.split_local = {}
        self.split_sizes = {}
        for k, idx in data.split_idx.items():
            m = (idx >= lo) & (idx < hi)
            self.split_local[k] = (idx[m] - lo).to(device)
            self.split_sizes[k] = idx.numel()


def _zero_scalar(prob: "ShardedProblem") -> Tensor:
    if prob._zero is None:
        prob._zero = torch.zeros((), dtype=torch.float32, device=prob.x.device)     # a constant: made once, never written
    return prob._zero


ShardedProblem.zero_scalar = _zero_scalar


def _train_subgraph(prob: ShardedProblem) -> _TrainSubgraph:
    if prob._train_sub is None:
        prob._train_sub = _TrainSubgraph(prob, *prob._rows_cpu)
    return prob._train_sub


def allreduce_grads(params, group=None):
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class FlatGrads:
    """All parameter gradients in ONE buffer: the per-step gradient exchange is a single all-reduce.  ``loss.backward()`` runs with
    the parameters' ``.grad`` unset (each gradient is then assigned, not accumulated: no zero fill, no add per parameter);
    ``gather`` copies them into the buffer with one batched copy, and after the all-reduce ``.grad`` are views of the buffer
    (what the optimizer reads)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self):
        """Before ``backward``: unset the gradients (no kernel)."""
        for p in self.params:
            p.grad = None

    def gather(self):
        """After ``backward``: one batched copy of all gradients into the flat buffer (a parameter without gradient counts as 0)."""
        if not self.params:
            return
        stale = [(p, v) for p, v in zip(self.params, self.views) if p.grad is None or p.grad.data_ptr() != v.data_ptr()]
        if len(stale) == len(self.params):      # the step's usual case (``zero`` unset every gradient): one batched copy
            torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params], out=self.flat)
        else:
            # some gradients ARE the views already (backward without a preceding ``zero``, a second all_reduce, gradient
            # accumulation): autograd accumulated into the buffer in place; copy only the others (cat would overlap its output)
            for p, v in stale:
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, group=None):
        self.gather()
        dist.all_reduce(self.flat, group=group)


def _flat_grads_of(optimizer) -> FlatGrads:
    """The optimizer's FlatGrads, rebuilt when its parameter list changed (``add_param_group`` after the first step)."""
    params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
    fg = getattr(optimizer, "_egnn_flat_grads", None)
    if fg is None or len(fg.params) != len(params) or any(a is not b for a, b in zip(fg.params, params)):
        fg = FlatGrads(params)
        optimizer._egnn_flat_grads = fg
    return fg


def _sampled_rows(prob: ShardedProblem, S: int, dev):
    """criterion.py:62-65,134-137 on shards: ONE np.random.choice over the GLOBAL train list (the same draw on every rank), then
    (local positions of the rows this rank owns, as a device tensor; rows per rank).  ``prob.sample_hook`` (ShardedGraphedEpoch)
    replaces the host draw + upload by a static device buffer refilled before every replay."""
    if prob.sample_hook is not None:
        return prob.sample_hook(S)
    ntr = prob.n_train_global
    pick = prob._forced_pick                      # a draw already made for this step (StaticSample overflow fallback)
    prob._forced_pick = None
    if pick is None:
        pick = np.random.choice(ntr, S, replace=False) if S < ntr else np.arange(ntr)
    pick_t = torch.from_numpy(pick)
    owner = prob.train_owner[pick_t]
    counts = torch.bincount(owner, minlength=prob.world).tolist()
    return prob.train_localpos[pick_t[owner == prob.rank]].to(dev), counts


def finish_losses(vals, mode: str, hp: dict):
    """(loss, loss_cls, loss_aux) of the GLOBAL problem from the three reduced numbers of ``sharded_train_step_tensors``."""
    aux_is_global = mode in ("nce", "gpw")
    loss_cls_g = vals[0]
    loss_aux_g = vals[2] if aux_is_global else vals[1]
    if mode == "kd":
        loss_g = loss_aux_g * (hp["alpha"] * hp["kd_T"] ** 2) + loss_cls_g * (1 - hp["alpha"])
    elif mode in ("nce", "gpw", "lpw"):
        loss_g = loss_cls_g + hp["beta"] * loss_aux_g
    else:
        loss_g = loss_cls_g
    return loss_g, loss_cls_g, loss_aux_g


def sharded_train_step(model, prob: ShardedProblem, optimizer, mode: str, hp: dict, student_proj=None, teacher_proj=None):
    """The reference's ``train()`` (gnn.py:102-195) on one shard; returns the GLOBAL (loss, loss_cls, loss_aux)."""
    rep = sharded_train_step_tensors(model, prob, optimizer, mode, hp, student_proj, teacher_proj)
    return finish_losses(rep.tolist(), mode, hp)                 # one device->host read per step


def sharded_train_step_tensors(model, prob: ShardedProblem, optimizer, mode: str, hp: dict, student_proj=None, teacher_proj=None):
    """The step without its host read: forward, loss, backward, gradient all-reduce, Adam, and the all-reduce of the loss terms;
    returns the device tensor [loss_cls, loss_aux (sum over ranks), loss_aux (already global for nce / gpw)]."""
    from . import criterion as C
    model.train()
    for p in (student_proj, teacher_proj):
        if p is not None:
            p.train()
    group = prob.group
    take = ops.take_rows   # train ids are unique: gather / scatter without a sort
    logits = model(prob.x, prob.adj)
    rows, n_tr = prob.train_local, int(prob.train_local.numel())
    labels = prob.y.view(-1)
    frac = n_tr / prob.n_train_global                      # local mean -> contribution to the global mean
    dev = logits.device
    zero = prob.zero_scalar()
    # A rank that owns no train row still has to run the SAME backward collectives as its peers (halo exchange and
    # SyncBN reductions of every layer): its loss terms are exact zeros that stay attached to the model's graph.
    # (formed only on such a rank, from ONE row: never a long reduction inside a captured step, _audit.py)
    attached_zero = (logits[:1].sum() * 0.0) if n_tr == 0 else None
    # gnn.py:109-110 `out = model(...)[train_idx]`, `y.squeeze(1)[train_idx]`, `teacher_logits[train_idx]`: the row picks happen inside
    # the CE / KD kernels (their backward writes the dense logits gradient) -- no gather / zero-fill + scatter of [n_train, classes]
    ce = lambda: ops.cross_entropy(logits, labels, rows) * frac   # noqa: E731

    def heads(pick):
        """(student rows, teacher rows) of the projection heads: all local train rows enter the Linear and the (all-rank) BatchNorm
        statistics; with ``pick`` only those output rows are normalised and stored (ops.sync_bn_act(..., pick=))."""
        if hasattr(student_proj, "forward_rows"):
            return (student_proj.forward_rows(model.out_feat, rows, pick=pick), teacher_proj.forward_rows(prob.teacher_out_feat, rows, pick=pick, const_input=True))
        f, t = student_proj(take(model.out_feat, rows)), teacher_proj(take(prob.teacher_out_feat, rows))
        return (f, t) if pick is None else (f[pick], t[pick])
    if mode == "supervised":
        loss_cls = ce() if n_tr else attached_zero
        loss_aux = zero
        loss = loss_cls
    elif mode == "kd":
        if n_tr:
            lc, lk = ops.ce_and_kd(logits, labels, prob.teacher_logits, hp["kd_T"], rows)
            loss_cls, loss_aux = lc * frac, lk * frac
        else:
            loss_cls = loss_aux = attached_zero
        loss = loss_aux * (hp["alpha"] * hp["kd_T"] ** 2) + loss_cls * (1 - hp["alpha"])
    elif mode == "nce":
        loss_cls = ce() if n_tr else attached_zero
        if _static_sample_ready(prob):
            ss = prob.static_sample
            f, t = heads(ss.idx_dev[:ss.m])
            fhat_all, t_all = _static_sampled_pair(prob, f, t, normalize=True, reduce_grad=True)
            loss_aux = _BalancedNCE.apply(fhat_all, t_all, hp["nce_T"], prob.rank, prob.world, group)
        else:
            idx, counts = _sampled_rows(prob, hp["max_samples"], dev)
            f, t = heads(idx)
            fhat = ops.gather_normalize(f, None)
            that = ops.gather_normalize(t, None)
            loss_aux = _DistNCE.apply(fhat, that, hp["nce_T"], counts, prob.rank, group)
        # loss_aux is already the global value on every rank: scale its gradient contribution once (1/world per rank
        # would double count the all-reduce of parameter grads), so only the local row block's graph carries grad
        loss = loss_cls + hp["beta"] * loss_aux
    elif mode == "gpw":
        # GSP (criterion.py:57-92): the sampled rows of all ranks are gathered and the all-pairs loss is evaluated in full on
        # every rank (same NumPy draw everywhere; S <= 4096 in the configuration of record)
        from . import ops_pairwise
        loss_cls = ce() if n_tr else attached_zero
        if _static_sample_ready(prob):
            ss = prob.static_sample
            f, t = heads(ss.idx_dev[:ss.m])
            fs, ts = _static_sampled_pair(prob, f, t, normalize=False, reduce_grad=False)
        else:
            idx, counts = _sampled_rows(prob, hp["max_samples"], dev)
            f, t = heads(idx)
            fs = _GatherSampledRows.apply(f, counts, prob.rank, group)
            ts = _GatherSampledRows.apply(t, counts, prob.rank, group)
        loss_aux = ops_pairwise.gsp_loss(fs, ts, None, hp["kernel"])      # the GLOBAL value on every rank
        loss = loss_cls + hp["beta"] * loss_aux
    elif mode == "lpw":
        # LSP (criterion.py:95-126): every rank owns the softmax groups of its train nodes; remote train neighbours' rows of the
        # student's hidden state arrive by a halo exchange (autograd: reverse exchange), the teacher's once
        from . import ops_edge
        loss_cls = ce() if n_tr else attached_zero
        sub = _train_subgraph(prob)
        f_ext = sub.extend(model.out_feat)
        if sub.teacher_ext is None:
            with torch.no_grad():
                sub.teacher_ext = sub.extend(prob.teacher_out_feat.contiguous())
        if sub.e_local > 0:
            loss_aux = ops_edge.lsp_loss(f_ext, sub.teacher_ext, sub.edge_index, hp["kernel"]) * (sub.e_local / max(sub.e_global, 1))
        else:
            # no edge here: an exact zero that keeps the exchange's backward in the graph -- from ONE row: a reduction over all of
            # [n_ext, C] is the operator class _audit.CaptureAudit refuses inside a captured step (on this rank only: the ranks diverge)
            loss_aux = f_ext[:1].sum() * 0.0
        loss = loss_cls + hp["beta"] * loss_aux
    else:
        raise NotImplementedError(f"sharded training mode '{mode}'")
    fg = _flat_grads_of(optimizer)
    fg.zero()
    loss.backward()
    fg.all_reduce(group)
    optimizer.step()
    aux_is_global = mode in ("nce", "gpw")
    rep = torch.stack([loss_cls.detach(), loss_aux.detach() if not aux_is_global else zero, loss_aux.detach()])
    dist.all_reduce(rep[:2], group=group)                     # loss_aux of nce / gpw is already global: not reduced
    return rep


@torch.no_grad()
def sharded_evaluate_tensors(model, prob: ShardedProblem):
    """``test()`` on shards without the host read: (local logits, device tensor of the three GLOBAL hit counts)."""
    model.eval()
    out = model(prob.x, prob.adj)
    # argmax + the three hit counts in one pass of this package's kernel (no long torch reduction inside the captured epoch, _audit.py)
    correct = ops.split_accuracy(out, prob.y, prob.split_local, counts=True)[:3].to(torch.float32)
    dist.all_reduce(correct, group=prob.group)
    return out, correct


def sharded_evaluate(model, prob: ShardedProblem):
    out, correct = sharded_evaluate_tensors(model, prob)
    hits = correct.tolist()                                   # one device->host read for the three counts
    accs = tuple(hits[i] / max(1, prob.split_sizes[k]) for i, k in enumerate(("train", "valid", "test")))
    return out, accs


class ShardedGraphedEpoch:
    """One epoch of the sharded run (train step + eval on this rank's shard, collectives included) captured ONCE as a hipGraph
    and replayed -- the sharded counterpart of ``models.GraphedEpoch``: the ~250 launches and ~25 collectives of an epoch are
    enqueued by one call.  RCCL collectives are stream operations and are captured like kernels (all ranks capture and replay
    the same program, see tests/test_dist_gloo.py::test_every_rank_issues_the_same_collective_sequence).
    Static shapes are required.  ``kd`` / ``supervised`` / ``lpw`` have them at any world size; the sampled criteria (G-CRD,
    GSP) get them from ``StaticSample``: fixed-capacity row blocks per rank + a per-step permutation, so the BASELINE metric's
    own step (GCN + G-CRD) is capturable on every rank count.  A draw that does not fit the capacity (~1e-9 per step) runs
    that one step with eager launches and dynamic shapes -- every rank sees the same draw and takes the same branch.
    Host randomness as in the eager step: one np.random.choice per step, a fresh dropout seed per replay; one device->host
    read per epoch."""

    @staticmethod
    def capturable(world: int, mode: str) -> bool:
        return mode in ("kd", "supervised", "lpw", "nce", "gpw")

    def __init__(self, model, prob: ShardedProblem, optimizer, mode: str, hp: dict, student_proj=None, teacher_proj=None, warmup: int = 3,
                 static_sample: bool | None = None):
        if not prob.x.is_cuda:
            raise ValueError("ShardedGraphedEpoch needs GPU tensors")
        if not self.capturable(prob.world, mode):
            raise ValueError(f"the sharded '{mode}' step is not capturable")
        self.mode, self.hp, self.prob = mode, hp, prob
        self._args = (model, prob, optimizer, mode, hp, student_proj, teacher_proj)
        dev = prob.x.device
        S = hp.get("max_samples", 0) if mode in ("nce", "gpw") else 0
        self.n_pick = min(S, prob.n_train_global) if S else 0
        # one rank owns the whole sample: a static [S] id buffer does (the single-GPU GraphedEpoch's way); several ranks: StaticSample
        self.static = None
        if self.n_pick and (prob.world > 1 if static_sample is None else static_sample):
            self.static = StaticSample(prob, self.n_pick)
            self.static.external = True
        self._overflow = None            # the draw of the NEXT step when it does not fit the static capacity
        self._pick_dev = torch.zeros(max(self.n_pick, 1), dtype=torch.int64, device=dev)
        self._pick_host = torch.zeros(max(self.n_pick, 1), dtype=torch.int64).pin_memory()
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._seed_host = torch.zeros(1, dtype=torch.int64).pin_memory()

        def body():
            rep = sharded_train_step_tensors(model, prob, optimizer, mode, hp, student_proj, teacher_proj)
            _, correct = sharded_evaluate_tensors(model, prob)
            return rep, correct
        self._body = body
        from . import _cache
        # cached structures the captured launches read through raw pointers stay alive with this object (_cache.pinning)
        with self._installed(), _cache.pinning() as self._pinned:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for i in range(warmup):
                    self._refresh(require_fit=True)
                    if i == warmup - 1:   # no long torch reduction may be captured (their memset node: _audit.py); same check on every rank
                        from ._audit import CaptureAudit
                        with CaptureAudit() as audit:
                            body()
                        audit.check("ShardedGraphedEpoch")
                    else:
                        body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            self._refresh(require_fit=True)
            torch.cuda.synchronize(dev)
            # thread-local capture mode: the process group's watchdog thread polls the events of the warm-up collectives; in the
            # default (global) mode a call from ANY thread invalidates the capture (seen: abort in capture_end)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.rep, self.correct = body()
            # structural guard on every capture (_audit.check_captured_graph): no memset / host node next to the kernels and RCCL's nodes
            from ._audit import LongReductionInCapture, check_captured_graph, graph_node_kinds
            try:
                self.node_kinds = check_captured_graph(self.graph, "ShardedGraphedEpoch", kernels_only=False)
            except LongReductionInCapture as e:
                # One rank over RCCL has been observed (kernel + memcpy nodes only) and is held to that.  What RCCL itself puts into a
                # captured collective with SEVERAL ranks has never been observed (no multi-GPU box in any round): there a violation is
                # reported, not fatal -- the operator-name audit above still refuses long torch reductions on every rank count.
                if prob.world == 1:
                    raise
                import warnings
                warnings.warn(f"{e}  (world_size {prob.world}: reported only; RCCL's own graph nodes cannot be told apart here)")
                self.node_kinds = graph_node_kinds(self.graph)
            self.graph.instantiate()
            torch.cuda.synchronize(dev)
        self._refresh()

    class _Install:
        """The capture hooks (static sample buffers, device-side dropout seed) for the duration of a block."""

        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            o, prob = self.o, self.o.prob
            self.prev = (prob.sample_hook, prob.static_sample, ops._DROPOUT_SEED_DEV)
            if o.static is not None:
                prob.sample_hook, prob.static_sample = None, o.static
            else:
                prob.sample_hook = lambda S_: (o._pick_dev[:o.n_pick], [o.n_pick])     # one rank owns the whole sample
            ops._DROPOUT_SEED_DEV = o._seed_dev

        def __exit__(self, *exc):
            self.o.prob.sample_hook, self.o.prob.static_sample, ops._DROPOUT_SEED_DEV = self.prev

    def _installed(self):
        return ShardedGraphedEpoch._Install(self)

    def _draw(self, require_fit: bool = False):
        if getattr(self, "_uploaded", None) is not None:
            self._uploaded.synchronize()     # the previous upload has left the pinned buffers (microseconds: it sits in front of the replay)
        self._overflow = None
        if self.static is not None:
            pick = self.static.draw()
            while not self.static.fill(pick):
                if not require_fit:          # a replay cannot take this draw: step() runs it with eager launches
                    self._overflow = pick
                    break
                pick = self.static.draw()    # warm-up / capture only need SOME sample that fits (their results are discarded)
        elif self.n_pick:
            ntr = self.prob.n_train_global
            pick = np.random.choice(ntr, self.n_pick, replace=False) if self.n_pick < ntr else np.arange(ntr)
            self._pick_host.copy_(self.prob.train_localpos[torch.from_numpy(pick)])       # one rank: every picked row is local
        self._seed_host.random_()
        self._seed_host.bitwise_and_(0x3FFFFFFFFFFFFFFF)

    def _upload(self):
        if self.static is not None:
            if self._overflow is None:
                self.static.upload()
        elif self.n_pick:
            self._pick_dev.copy_(self._pick_host, non_blocking=True)
        self._seed_dev.copy_(self._seed_host, non_blocking=True)
        # the copies above are only stream-ordered: the host must not rewrite the pinned staging buffers before the DMA has read them
        self._uploaded = torch.cuda.Event()
        self._uploaded.record()

    def _refresh(self, require_fit: bool = False):
        self._draw(require_fit)
        self._upload()

    def redraw(self):
        """Discard the randomness prepared for the next step and draw it again (after re-seeding NumPy / torch)."""
        torch.cuda.current_stream().synchronize()
        self._refresh()

    def _eager_step(self, pick):
        """The step whose draw does not fit the static capacity: same program with dynamic shapes and eager launches."""
        prev = (self.prob.sample_hook, self.prob.static_sample, ops._DROPOUT_SEED_DEV)
        self.prob.sample_hook, self.prob.static_sample, ops._DROPOUT_SEED_DEV = None, None, self._seed_dev
        self.prob._forced_pick = pick
        try:
            rep, correct = self._body()
        finally:
            self.prob.sample_hook, self.prob.static_sample, ops._DROPOUT_SEED_DEV = prev
        return rep, correct

    def _launch(self):
        """Enqueue one epoch (the replay, or -- a draw that does not fit the static capacity -- the same program with eager launches);
        returns the device tensors holding its (losses, hit counts)."""
        if self._overflow is not None:
            return self._eager_step(self._overflow)
        self.graph.replay()
        return self.rep, self.correct

    def _decode(self, vals):
        accs = tuple(vals[3 + i] / max(1, self.prob.split_sizes[k]) for i, k in enumerate(("train", "valid", "test")))
        return finish_losses(vals[:3], self.mode, self.hp), accs

    def step(self):
        """Replay one epoch; returns ((loss, loss_cls, loss_aux), (train, valid, test accuracies))."""
        if getattr(self, "_pending", None) is not None:
            raise RuntimeError("ShardedGraphedEpoch.step() after step_async(): call drain() first (an epoch's values are still in flight)")
        rep, correct = self._launch()
        self._draw()                                              # the next step's host draw overlaps the replay
        vals = torch.cat([rep, correct]).tolist()                 # one device->host read per epoch
        self._upload()
        return self._decode(vals)

    # the loop without an idle GPU between epochs (models.GraphedEpoch.step_async on shards): epoch k is launched -- collectives included --
    # before the host reads the values of epoch k - 1; same replays, same draws in the same order on every rank
    def step_async(self):
        """Launch one epoch and return the values of the PREVIOUS ``step_async`` epoch (None on the first call); ``drain()`` hands out
        the last one."""
        if getattr(self, "_res_host", None) is None:
            self._res_host = [torch.zeros(6, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._res_done, self._pending, self._k = [None, None], None, 0
        slot = self._k & 1
        self._k += 1
        rep, correct = self._launch()
        self._res_host[slot].copy_(torch.cat([rep, correct]).to(torch.float32), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._res_done[slot] = done
        self._draw()
        self._upload()
        prev, self._pending = self._pending, slot
        return None if prev is None else self._values(prev)

    def _values(self, slot):
        self._res_done[slot].synchronize()
        return self._decode(self._res_host[slot].tolist())

    def drain(self):
        prev, self._pending = getattr(self, "_pending", None), None
        return None if prev is None else self._values(prev)


# ------------------------------------------------------------------------------------------------
# bench entry (called by bench.py when WORLD_SIZE > 1)
# ------------------------------------------------------------------------------------------------
def mag_problem(scale: float, seed: int):
    """BASELINE.json configs[4] shape: the ogbn-mag graph made homogeneous (N = 1 939 743, 42.2 M directed entries after the
    reverse edges, x [N,128], 349 classes; /root/reference/mag_pyg/gnn.py:322-346), a GraphSage(mean) student distilled
    from (synthetic) R-GCN teacher logits -- the operator of mag_pyg/gnn.py:151,162 on node-range shards."""
    from . import data as D
    d = D.mag_like(scale, seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    n = d.num_nodes
    d.y = torch.randint(0, d.num_classes, (n, 1), generator=g)
    perm = torch.randperm(n, generator=g)
    n_tr, n_va = int(0.325 * n), int(0.033 * n)        # 629 571 / 64 879 / 41 939 labelled papers of 1.94 M nodes
    d.split_idx = {"train": perm[:n_tr].clone(), "valid": perm[n_tr:n_tr + n_va].clone(), "test": perm[n_tr + n_va:n_tr + 2 * n_va].clone()}
    d.teacher_logits = torch.randn(n, d.num_classes, generator=g) * 3.0
    d.teacher_out_feat = None
    return d


def bench_main(args, hp, model_cfg, rank, world, device, backend: str = "nccl", emit=print):
    from . import data as D
    from . import models as PM
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **({"device_id": device} if on_gpu and backend == "nccl" else {}))
    import random
    for s in (random.seed, np.random.seed, torch.manual_seed):
        s(args.seed)
    if on_gpu:
        torch.cuda.manual_seed_all(args.seed)

    def sync():
        dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
    workload = getattr(args, "workload", "arxiv")
    if workload == "mag":    # config 5: SAGE-mean student + logit KD on the MAG-shaped graph
        data = mag_problem(args.scale, args.seed)
        args.gnn, args.training = "sage", "kd"
    else:
        data = D.arxiv_like(args.scale, seed=args.seed, graph=getattr(args, "graph_kind", "chunglu"))   # same seeded graph on every rank (host-side, one-off)
    # locality-aware node order before the ranges are cut (SURVEY 8(e) "optional: locality reordering"): kept when it lowers the halo
    partition = dict(order="node ids as given")
    if world > 1 and getattr(args, "partition", "auto") != "range":
        perm, halo_before, halo_after = locality_order(data, world, device if on_gpu else None)
        partition = dict(order="community order (sparse.community_order)" if perm is not None else "node ids as given (the community order did not lower the halo)",
                         halo_rows_per_rank_as_given=halo_before, halo_rows_per_rank_community_order=halo_after,
                         halo_rows_change=round(sum(halo_after) / max(sum(halo_before), 1) - 1.0, 4))
        if perm is not None:
            data = reorder_nodes(data, perm)
    prob = ShardedProblem(data, world, rank, device, None, need_gcn=(args.gnn == "gcn"))
    Net = PM.GCN if args.gnn == "gcn" else PM.SAGE
    model = Net(data.num_features, model_cfg["hidden"], data.num_classes, model_cfg["layers"], model_cfg["dropout"]).to(device)
    swap_batchnorm(model)
    sp = tp = None
    groups = [{"params": model.parameters(), "lr": model_cfg["lr"]}]
    if args.training in ("nce", "gpw"):
        sp = swap_batchnorm(PM.make_projection(model_cfg["hidden"], hp["proj_dim"]).to(device))
        tp = swap_batchnorm(PM.make_projection(data.teacher_out_feat.shape[1], hp["proj_dim"]).to(device))
        groups += [{"params": sp.parameters(), "lr": model_cfg["lr"]}, {"params": tp.parameters(), "lr": model_cfg["lr"]}]
    if on_gpu:   # same hyper-parameters in every group (gnn.py:308-312): one launch
        groups = [{"params": [p for g in groups for p in g["params"]], "lr": model_cfg["lr"]}]
    opt = torch.optim.Adam(groups, fused=on_gpu, capturable=on_gpu)
    torch.manual_seed(args.seed + 1000 + rank)               # dropout masks differ per shard

    def eager_epoch():
        l = sharded_train_step(model, prob, opt, args.training, hp, sp, tp)
        _, a = sharded_evaluate(model, prob)
        return l, a

    # hipGraph replay of the epoch where the step has static shapes (ShardedGraphedEpoch.capturable): --graph on / auto
    graphed, graph_note = None, "eager launches"
    want_graph = getattr(args, "graph", "off") in ("on", "auto") and on_gpu
    if want_graph and ShardedGraphedEpoch.capturable(world, args.training):
        err = None
        try:
            graphed = ShardedGraphedEpoch(model, prob, opt, args.training, hp, sp, tp, warmup=3)
            graph_note = "hipGraph replay of the sharded train step + eval, collectives captured (dist.ShardedGraphedEpoch)"
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {str(e)[:200]}"
        # every rank must time the SAME collective program: agree on the outcome (a capture that failed on one rank only would
        # leave it issuing eager warm-up collectives against its peers' replays -- a hang instead of a report)
        failed = torch.tensor([0.0 if err is None else 1.0], device=device)
        dist.all_reduce(failed)
        if float(failed.item()) > 0:
            graphed = None
            why = err or "capture failed on another rank"
            if getattr(args, "graph", "off") == "on":    # the headline number is the replayed epoch: never silently time something else
                raise SystemExit(f"bench.py: hipGraph capture of the sharded epoch failed on {int(failed.item())} of {world} ranks ({why}); "
                                 f"use --graph auto or --graph off to time eager launches")
            graph_note = f"capture of the sharded epoch failed, eager launches on every rank: {why}"
    elif want_graph:
        graph_note = f"eager launches: the sharded '{args.training}' step is not capturable"
    epoch = graphed.step if graphed is not None else eager_epoch

    def run(n):
        """n epochs of the timed program; replays launch ahead (step_async: epoch k before the values of epoch k - 1 are read)."""
        if graphed is None:
            for _ in range(n):
                vals = eager_epoch()
            return vals
        for _ in range(n):
            graphed.step_async()
        return graphed.drain()
    # untimed settle phase (as in bench.py's single-GPU path; declared in the line): the devices leave set-up in a low clock state
    settle_s = float(getattr(args, "settle_seconds", 0.0)) if graphed is not None else 0.0
    if settle_s > 0:
        ts = time.perf_counter()
        go = torch.ones(1, device=device)
        while float(go.item()) > 0:          # every rank runs the same number of blocks: rank 0's clock decides
            run(max(args.steps, 10))
            go.fill_(1.0 if (time.perf_counter() - ts) < settle_s else 0.0)
            dist.broadcast(go, src=0)
    if args.warmup > 0:               # the W untimed warm-up steps: of the program that is timed (replays, or eager epochs)
        run(args.warmup)
    # The interpreter holds ~170k long-lived objects after the torch / RCCL imports; a full (generation-2) collection
    # walks all of them (~40 ms) and the per-step autograd / collective bookkeeping triggers one every few steps.
    # Freezing the survivors of set-up keeps later collections proportional to the per-step garbage.
    import gc
    gc.collect()
    gc.freeze()
    # rank 0 brackets its local aggregation launches (HIP events on the launch stream) for the roofline object
    probe_records = []
    orig_raw = ops.spmm_raw
    if on_gpu and rank == 0:
        def probed(adj, x, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_raw(adj, x, *a, **kw)
            e1.record()
            probe_records.append((x.shape[1], adj.sparse_sizes(), adj.nnz(), adj.spmm_algorithmic_bytes(x.shape[1]), e0, e1))
            return out
        ops.spmm_raw = probed
    sync()
    t0 = time.perf_counter()
    losses, accs = run(args.steps)
    sync()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    halo = torch.tensor([float(prob.adj.plan.n_halo)], device=device)
    dist.all_reduce(halo)
    # one more epoch, outside the timed region, with the collectives recorded: per-rank communication volume, the compute
    # window every halo exchange is hidden under and what is left exposed (HIP events on the compute stream)
    with CommTrace(probe_overlap=on_gpu) as trace:
        eager_epoch()
        if on_gpu:
            torch.cuda.synchronize()
    ops.spmm_raw = orig_raw          # (the aggregation brackets cover the timed region when it ran eagerly, and this epoch)
    comm = trace.summary()
    per_rank_comm = [None] * world
    dist.all_gather_object(per_rank_comm, comm)
    if rank == 0:
        el = float(elapsed)
        roofline = None
        if probe_records:   # the widest aggregation of the step, per launch on rank 0's shard (algorithmic bytes of the shard's pieces)
            kmax = max(r[0] for r in probe_records)
            sel = [r for r in probe_records if r[0] == kmax]
            secs = sum(a.elapsed_time(b) for *_, a, b in sel) * 1e-3
            nbytes = sum(r[3] for r in sel)
            gbs = nbytes / secs / 1e9
            roofline = dict(bound="hbm", kernel=f"rank 0's local aggregation launches (egnn_spmm_csr_blk_f32 on the shard's own-column and "
                                                f"halo-column pieces), K={kmax}", achieved=round(gbs, 1), peak=8000.0, unit="GB/s",
                            frac=round(gbs / 8000.0, 4), launches_timed=len(sel), algorithmic_bytes_timed=int(nbytes), traffic=None)
        if workload == "mag":
            metric = "training epochs/sec, ogbn-mag-shaped graph, GraphSage(mean) student + logit KD, node-range shards"
            wl = (f"ogbn-mag-shaped synthetic homogeneous graph (N={data.num_nodes}, nnz={data.adj_t.nnz()}), {model_cfg['layers']}-layer "
                  f"SAGE-{model_cfg['hidden']} (mean) student + logit KD from synthetic teacher logits, full-graph train step + eval per epoch")
        else:
            metric = "training epochs/sec, ogbn-arxiv 3-layer GCN student + G-CRD, 1/2/4/8 MI355X"
            wl = (f"ogbn-arxiv-shaped synthetic graph (N={data.num_nodes}, nnz_sym={data.adj_t.nnz()}), 3-layer {args.gnn.upper()}-{model_cfg['hidden']} student + "
                  f"{args.training} loss (max_samples={hp['max_samples']}, proj_dim={hp['proj_dim']}), full-graph train step + eval per epoch")
        out = dict(
            metric=metric, value=round(args.steps / el, 3), unit="epochs/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=round(1e3 * el / args.steps, 3), higher_is_better=True, scaling="strong", vs_baseline=None,
            dtype="f32", data="synthetic",
            config=dict(workload=wl,
                        partitioning=f"node-range shards x{world}: halo all_to_all {'overlapped with the own-column aggregation' if _OVERLAP else '(blocking)'}"
                                     f" + SyncBN all-reduce + flat grad all-reduce over "
                                     + ("RCCL" if backend == "nccl" else f"{backend} (host-staged: all ranks share one GPU, hostcomm.py -- a functional run, "
                                                                          f"not a scaling measurement)" if on_gpu else backend),
                        mean_halo_rows_per_rank=int(float(halo) / world), node_order=partition,
                        aggregation_exchange=dict(mode=_AGG_MODE, sliced_for_width=[K for K in (32, 64, 128, 256, 512) if prob.adj.sliced_pays(K)],
                                                  rule="sliced (feature columns re-sharded around the aggregation, 2 N K 4 (G-1)/G^2 bytes per rank) where "
                                                       "the all-rank mean halo x G^2 > 2 N (G-1) and K % (4 G) == 0; halo rows otherwise")),
            launch=graph_note + ("; step_async loop (epoch k launched before the values of epoch k-1 are read), "
                                 f"{settle_s:g} s of untimed settle replays in front of the warm-up" if graphed is not None else ""),
            comm_per_epoch=dict(what="one epoch (train step + eval) traced after the timed region: payload bytes per rank and kind; "
                                     "overlap_window_us = own-column aggregation time the halo exchanges run under, exposed_comm_us = what the "
                                     "compute stream still waits for them afterwards (forward exchanges; HIP events)",
                                per_rank=per_rank_comm,
                                halo_rows=partition),     # halo rows per rank in the given order / in the community order, and which one was cut
            roofline=roofline, cpu_baseline=None,
            last_losses=[round(float(v), 5) for v in losses], last_accs=[round(float(a), 4) for a in accs])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush that first so that the JSON line is the LAST line of stdout
        import ctypes
        import sys
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        emit(json.dumps(out))
        sys.stdout.flush()
