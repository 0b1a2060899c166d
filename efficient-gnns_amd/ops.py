"""autograd.Function wrappers around the C ABI (include/egnn_hip.h).  GPU only; no CPU fallback."""
from __future__ import annotations

import os

import torch
from torch import Tensor

from . import _cache, _lib

_REDUCE = {"sum": 0, "add": 0, "mean": 1, "max": 2}
_OVERLAP_HEAVY_ROWS = True   # (no environment switch any more; tools/spmm_pmc.py clears the attribute to trace one stream)
_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]



def _rowmajor(x: Tensor) -> Tensor:
    """fp32, unit inner stride (the kernels take an explicit leading dimension)."""
    if x.dtype != torch.float32:
        raise TypeError(f"expected float32, got {x.dtype}")
    if x.dim() != 2:
        raise ValueError("expected a 2-D tensor")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    return x


# ------------------------------------------------------------------------------------------------
# SpMM
# ------------------------------------------------------------------------------------------------
# 'blocks' (default): egnn_spmm_csr_blk_f32; 'segments': egnn_spmm_csr_seg_f32; 'classes': short / mid / long row classes
_SPMM_SCHEDULE = "blocks"   # "blocks" (row blocks, K >= 64) -> "segments" (K < 64) -> "classes" (max / unaligned); tests pin one by setting the attribute


class HipStatsUnavailable(RuntimeError):
    """spmm_raw(want_stats=True) on a shape / schedule whose kernel has no statistics epilogue (callers fall back to the
    separate egnn_bn_stats_f32 pass)."""


def spmm_raw(adj, x: Tensor, reduce: str = "sum", src_scale: Tensor | None = None, use_plan: bool = True,
             bias: Tensor | None = None, out: Tensor | None = None, stat_shift: Tensor | None = None, want_stats: bool = False,
             addend: Tensor | None = None, relu: bool = False):
    """``relu``: Y = max(REDUCE(adj, X) + bias, 0), fused into the block kernel's store (eval-mode BatchNorm folded into the
    weights + ReLU, ``bn_fold``); on the other schedules a clamp follows."""
    fused = []   # filled by _spmm_raw when the kernel applied the ReLU in its store (a value, not a tag on the -- possibly caller-owned -- output buffer)
    res = _spmm_raw(adj, x, reduce, src_scale, use_plan, bias, out, stat_shift, want_stats, addend, relu, fused)
    if relu and not fused:
        res[0].clamp_(min=0)
    return res


def _spmm_raw(adj, x: Tensor, reduce: str = "sum", src_scale: Tensor | None = None, use_plan: bool = True,
              bias: Tensor | None = None, out: Tensor | None = None, stat_shift: Tensor | None = None, want_stats: bool = False,
              addend: Tensor | None = None, relu: bool = False, relu_fused: list | None = None):
    """Y = REDUCE(adj, X) on the GPU.  Returns (Y, argmax | None), or (Y, None, (mean, biased var)) with ``want_stats``.

    Schedules: 'blocks' (default; egnn_spmm_csr_blk_f32: one launch over hub segments + row blocks, int32 indices, then the
    fixed-order combine of the hub rows), 'segments' (round-1 egnn_spmm_csr_seg_f32), 'classes' (egnn_spmm_csr_f32: also the
    path of ``max`` and of shapes the float4 kernels do not take).
    ``out``: optional [n_rows, K] destination with unit column stride (e.g. a column block of a wider matrix).
    ``addend``: optional [n_rows, K] matrix added to the result in the kernel's store (Y = A X + addend; sum / mean).
    ``want_stats`` (sum / mean on the block schedule only): per-column mean and biased variance of Y over all rows, formed in
    the aggregation's epilogue (BatchNorm statistics, gnn.py:47-48); ``stat_shift`` [K]: shift of the shifted sums."""
    _lib.require_gpu(x, adj._col)
    x = _rowmajor(x)
    n_rows, n_src = adj.sparse_sizes()
    if x.shape[0] != n_src:
        raise ValueError(f"matmul: adjacency has {n_src} columns, x has {x.shape[0]} rows")
    K = x.shape[1]
    red = _REDUCE[reduce]
    if out is not None:
        if tuple(out.shape) != (n_rows, K) or out.stride(1) != 1 or out.dtype != torch.float32 or out.device != x.device:
            raise ValueError("spmm_raw: `out` must be a float32 [n_rows, K] view with unit column stride on x's device")
        y = out
    else:
        y = torch.empty(n_rows, K, dtype=torch.float32, device=x.device)
    arg = torch.empty(n_rows, K, dtype=torch.int64, device=x.device) if red == 2 else None
    if addend is not None:
        if red == 2 or tuple(addend.shape) != (n_rows, K) or addend.dtype != torch.float32:
            raise ValueError("spmm_raw: `addend` is a float32 [n_rows, K] matrix (sum / mean only)")
        addend = _rowmajor(addend)
    if adj.nnz() == 0:   # no stored entry at all: every row is empty (sum / mean / max = 0, argmax = -1), plus the bias
        y.zero_()
        if bias is not None:
            y.add_(bias)
        if addend is not None:
            y.add_(addend)
        if arg is not None:
            arg.fill_(-1)
        if want_stats:
            raise HipStatsUnavailable()
        return y, arg
    rowptr, col, bits = adj._index_arrays()
    lib = _lib.load()
    float4_ok = (use_plan and red != 2 and K % 4 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
                 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 16 == 0))
    add_fused = addend is not None and addend.stride(0) % 4 == 0 and addend.data_ptr() % 16 == 0
    # the block schedule wins from two 128-byte column slices up (K = 256: 343 vs 385 us, K = 128: 167 vs 186 us on the
    # arxiv-shaped graph); below that (the 40-class output layer) the 16-lane form of the segment kernel is faster (92 vs 113 us)
    if (float4_ok and _SPMM_SCHEDULE == "blocks" and bits == 32 and K >= 64 and n_src * x.stride(0) * 4 < 2 ** 31 and n_rows > 0
            and (addend is None or add_fused)):
        from .sparse import BLK_ROWS, SEG_MAX
        hseg, crow, cptr, slots = adj._blk_plan()
        partial = adj._scratch("partial", (max(slots, 1), K))
        loc = adj._struct.get("locality")          # opt-in: (rows_per_blk, win) of SparseTensor.stage_diagonal_blocks()
        use_lds = loc is not None and n_rows == n_src and K % 32 == 0
        rows_blk = loc[0] if use_lds else BLK_ROWS
        n_blk = (n_rows + rows_blk - 1) // rows_blk
        # Y rows are stored write-through (dropped from L2: they are not re-read here and would only evict gathered X lines)
        flags = 4 | (8 if (relu and not want_stats) else 0)
        n_stat = lib.egnn_spmm_blk_stat_rows(n_rows, rows_blk, int(use_lds)) if want_stats else 0
        n_hub = crow.numel()
        stat_part = adj._scratch("stat", (n_stat + n_hub, 2, K)) if want_stats else None   # one partial row per wave + per hub row
        if stat_shift is not None:
            stat_shift = stat_shift.detach().contiguous()
        rc = lib.egnn_spmm_csr_blk_f32(n_rows, n_src, K, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(adj._value), _lib.ptr(src_scale),
                                       _lib.ptr(bias), _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), red, SEG_MAX, rows_blk, None, 0,
                                       _lib.ptr(loc[1]) if use_lds else None, _lib.ptr(hseg), hseg.shape[0], _lib.ptr(partial),
                                       _lib.ptr(addend), 0 if addend is None else addend.stride(0),
                                       _lib.ptr(stat_part), _lib.ptr(stat_shift) if want_stats else None, flags, _lib.stream())
        if rc == 0:
            if n_hub > 0:   # the hub rows: fixed-order sum of their partial slots (+ mean / bias, + their statistics rows)
                _lib.check(lib.egnn_spmm_combine_f32(n_rows, K, _lib.ptr(rowptr), bits, _lib.ptr(bias), _lib.ptr(y), y.stride(0), red,
                                                     _lib.ptr(crow), _lib.ptr(cptr), n_hub, _lib.ptr(partial), _lib.ptr(addend),
                                                     0 if addend is None else addend.stride(0), _lib.ptr(stat_part), n_stat,
                                                     _lib.ptr(stat_shift) if want_stats else None, flags, _lib.stream()), "egnn_spmm_combine_f32")
            if not want_stats:
                if flags & 8 and relu_fused is not None:
                    relu_fused.append(True)
                return y, None
            mean = torch.empty(K, dtype=torch.float32, device=x.device)
            var = torch.empty(K, dtype=torch.float32, device=x.device)
            nws = lib.egnn_bn_stats_merge_ws_floats(K)
            ws = adj._scratch("statfold", (nws,))
            _lib.check(lib.egnn_bn_stats_merge_f32(_lib.ptr(stat_part), n_stat + n_hub, K, None, 0, None, 0,
                                                   _lib.ptr(stat_shift), n_rows, _lib.ptr(mean), _lib.ptr(var), _lib.ptr(ws), nws,
                                                   _lib.stream()), "egnn_bn_stats_merge_f32")
            return y, None, (mean, var)
        if rc != -4:   # EGNN_EALIGN: shape outside the block kernel's forms -> the schedules below
            _lib.check(rc, "egnn_spmm_csr_blk_f32")
    if want_stats:
        raise HipStatsUnavailable()
    if float4_ok and _SPMM_SCHEDULE in ("blocks", "segments"):
        # every row as ranges of <= 64 entries through the sub-group-per-row kernel (hub rows get the bulk's parallelism)
        seg, crow, cptr, slots = adj._seg_plan()
        partial = adj._scratch("partial", (max(slots, 1), K))
        rc = lib.egnn_spmm_csr_seg_f32(n_rows, n_src, K, _lib.ptr(rowptr), _lib.ptr(col), bits, _lib.ptr(adj._value), _lib.ptr(src_scale),
                                       _lib.ptr(bias), _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), red, _lib.ptr(seg),
                                       seg.shape[0], _lib.ptr(crow), _lib.ptr(cptr), crow.numel(), _lib.ptr(partial), slots, _lib.stream())
        if rc == 0:
            if addend is not None:
                y.add_(addend)
            return y, None
        if rc != -4:   # EGNN_EALIGN: shape outside the segment kernel's float4 forms -> classic schedule below
            _lib.check(rc, "egnn_spmm_csr_seg_f32")
    short, mid, long_ = adj._row_plan() if use_plan else (None, None, None)

    def lst(t):
        return (None, 0) if t is None or t.numel() == 0 else (_lib.ptr(t), t.numel())

    def launch(s, m, l, stream):
        (ps, ns), (pm, nm), (pl, nl) = lst(s), lst(m), lst(l)
        return lib.egnn_spmm_csr_f32(
            n_rows, n_src, K, _lib.ptr(rowptr), _lib.ptr(col), bits, _lib.ptr(adj._value), _lib.ptr(src_scale), _lib.ptr(bias),
            _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), red, _lib.ptr(arg), ps, ns, pm, nm, pl, nl, stream)

    heavy = use_plan and short is not None and short.numel() > 0 and (mid.numel() + long_.numel()) > 0
    if heavy and _OVERLAP_HEAVY_ROWS:
        # the few mid / long rows run on a side stream underneath the short-row bulk (fork / join with events)
        main = torch.cuda.current_stream()
        side = _side_stream(x.device)
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        rc = launch(None, mid, long_, side.cuda_stream)
        _lib.check(rc, "egnn_spmm_csr_f32")
        rc = launch(short, None, None, main.cuda_stream)
        join = torch.cuda.Event()
        join.record(side)
        main.wait_event(join)
        for t in (x, y, arg, adj._value, src_scale, bias):
            if t is not None:
                t.record_stream(side)
    else:
        rc = launch(short, mid, long_, _lib.stream())
    _lib.check(rc, "egnn_spmm_csr_f32")
    if addend is not None:
        y.add_(addend)
    return y, arg


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, reduce, bias=None, stat_shift=None, want_stats=False):
        b = None if bias is None else bias.detach().contiguous()
        stats = None
        if want_stats:
            try:
                y, arg, stats = spmm_raw(adj, x, reduce, bias=b, stat_shift=stat_shift, want_stats=True)
            except HipStatsUnavailable:
                y, arg = spmm_raw(adj, x, reduce, bias=b)
        else:
            y, arg = spmm_raw(adj, x, reduce, bias=b)
        ctx.adj, ctx.reduce, ctx.has_bias = adj, reduce, bias is not None
        ctx.tap_box = getattr(x, "_egnn_tap", None)     # x is a tapped tensor (grad_tap): its gradient may be completed in place, see _fresh
        if arg is not None:
            ctx.save_for_backward(arg)
        if stats is None:
            return y, None, None
        ctx.mark_non_differentiable(*stats)
        return y, stats[0], stats[1]

    @staticmethod
    def backward(ctx, gy, _gm=None, _gv=None):
        adj, reduce = ctx.adj, ctx.reduce
        gy = _rowmajor(gy)
        if reduce in ("sum", "add"):
            gx, _ = spmm_raw(adj.t(), gy, "sum")
        elif reduce == "mean":
            # dX = A^T (dY / cnt): per-source-row scale of the transposed aggregation
            gx, _ = spmm_raw(adj.t(), gy, "sum", src_scale=adj._inv_rowcount())
        else:
            (arg,) = ctx.saved_tensors
            n_rows, n_src = adj.sparse_sizes()
            K = gy.shape[1]
            gx = torch.zeros(n_src, K, dtype=torch.float32, device=gy.device)
            if adj.nnz() == 0:   # no stored entry: nothing receives gradient
                return gx, None, None, (colsum(gy) if ctx.has_bias and ctx.needs_input_grad[3] else None), None, None
            _, col, bits = adj._index_arrays()
            rc = _lib.load().egnn_spmm_csr_max_bwd_f32(n_rows, K, _lib.ptr(col), bits, _lib.ptr(adj._value), _lib.ptr(arg),
                                                       _lib.ptr(gy), gy.stride(0), _lib.ptr(gx), gx.stride(0), _lib.stream())
            _lib.check(rc, "egnn_spmm_csr_max_bwd_f32")
        gb = colsum(gy) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return _fresh(gx, ctx.tap_box), None, None, gb, None, None


def _fresh(g, tap_box):
    """Marks a gradient this package has just allocated in a backward FOR the tapped tensor whose box is ``tap_box``: nobody else
    holds it yet, so that tensor's ``_GradTap`` may add its row-compact pieces into it in place.  The mark names the box: a gradient
    that reaches a tap through a pass-through node (an add, a view: the same tensor object can then also be a sibling input's
    gradient) was produced for some OTHER tensor, carries no mark for this box, and is copied first."""
    if g is not None and tap_box is not None:
        g._egnn_fresh_for = tap_box
    return g


def colsum(g: Tensor) -> Tensor:
    """Column sums of a [n, C] gradient (bias gradients).  A gradient that comes straight out of the fused BatchNorm
    backward carries them already (formed while dx was written: egnn_bn_act_bwd_colsum_f32), tagged with the tensor
    version they belong to -- any in-place change of the gradient since then (autograd accumulation) voids the tag."""
    tag = getattr(g, "_egnn_colsum", None)
    if tag is not None and tag[1] == g._version and tag[0].shape[0] == g.shape[1]:
        return tag[0]
    _lib.require_gpu(g)
    if g.dim() != 2 or g.dtype != torch.float32 or g.shape[0] == 0:
        return g.sum(0)
    # egnn_colsum_f32, not ``g.sum(0)``: torch's multi-block reduction zeroes its semaphores with a memset node, and inside replayed
    # hipGraphs on this stack such reductions were seen to leave their output unwritten (ops_edge._LspLoss)
    g = _rowmajor(g)
    n, C = g.shape
    lib = _lib.load()
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    ws = torch.empty(lib.egnn_colsum_ws_floats(C), dtype=torch.float32, device=g.device)
    _lib.check(lib.egnn_colsum_f32(_lib.ptr(g), g.stride(0), n, C, _lib.ptr(out), _lib.ptr(ws), _lib.stream()), "egnn_colsum_f32")
    return out


class _AddBias(torch.autograd.Function):
    """x + bias (broadcast over rows); the bias gradient is ``colsum`` -- not autograd's ``sum_to_size``, a long torch reduction over
    the node dimension (see _audit.py)."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias

    @staticmethod
    def backward(ctx, g):
        return g, (colsum(g) if ctx.needs_input_grad[1] else None)


def add_bias(x: Tensor, bias: Tensor | None) -> Tensor:
    if bias is None:
        return x
    _lib.require_gpu(x)
    return _AddBias.apply(x, bias)


def spmm(adj, x: Tensor, reduce: str = "sum", bias: Tensor | None = None, bn_stats_shift: Tensor | None = None,
         want_bn_stats: bool = False, addend: Tensor | None = None) -> Tensor:
    """adj @ x with the given reduction; ``bias`` ([K]) is added in the kernel's store (sum / mean only).

    ``want_bn_stats``: the caller will BatchNorm the result next (gnn.py:47-48): the column mean / biased variance are
    formed in the aggregation's epilogue and attached to the returned tensor (``_egnn_bn_stats``) for ``ops.bn_act``;
    ``bn_stats_shift``: shift of the shifted sums (the BatchNorm's running_mean).  Where the kernel in use has no such
    epilogue nothing is attached and ``bn_act`` runs its own statistics pass."""
    if reduce not in _REDUCE:
        raise ValueError(f"unknown reduce '{reduce}'")
    if addend is not None:   # accumulating form (adj @ x + addend), used without autograd by the sharded run's own Functions
        if torch.is_grad_enabled() and (x.requires_grad or addend.requires_grad):
            raise NotImplementedError("spmm(addend=...) is a forward-only form")
        return spmm_raw(adj, x, reduce, bias=bias, addend=addend)[0]
    if bias is not None and (reduce == "max" or (x.shape[1] % 4 == 0 and bias.data_ptr() % 16 != 0)):
        return _SpMM.apply(x, adj, reduce, None)[0] + bias
    y, mean, var = _SpMM.apply(x, adj, reduce, bias, bn_stats_shift, bool(want_bn_stats) and reduce != "max")
    if mean is not None:
        y._egnn_bn_stats = (mean, var, y._version)   # version-checked by bn_act: an in-place edit of y voids the statistics
    return y


class _TakeRows(torch.autograd.Function):
    """x[idx] for UNIQUE row ids: the backward is a plain scatter (ATen's index backward sorts the ids every step)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gx = torch.zeros(ctx.n, *g.shape[1:], dtype=g.dtype, device=g.device)
        gx.index_copy_(0, idx, g)
        return gx, None


def take_rows(x: Tensor, idx: Tensor) -> Tensor:
    return _TakeRows.apply(x, idx)


# ------------------------------------------------------------------------------------------------
# dense GEMM (fp32 MFMA)
# ------------------------------------------------------------------------------------------------
def gemm_backend() -> str:
    """Always 'hip': every GEMM of the package runs on the hand-written kernels (egnn_gemm_f32 and relatives).  The
    rocBLAS comparison lives in tools/kernel_bench.py, outside the product."""
    return "hip"


def _pitch_ok(t: Tensor) -> bool:
    """2-D, unit column stride, rows 16-byte aligned: the kernels' float4 path takes it as it is."""
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def _gemm_operand(t: Tensor) -> Tensor:
    """A row-major view the kernels can address: kept as it is when only its row pitch is padded (see pad_pitch)."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return _rowmajor(t)


def pad_pitch(t: Tensor) -> Tensor:
    """The same [n, C] values behind a row pitch that is a multiple of 4 floats (a view of an [n, C+pad] buffer), so
    that rows start 16-byte aligned and the GEMM / gather kernels can use float4 loads (C = 750 teacher features)."""
    n, C = t.shape
    if C % 4 == 0 and _pitch_ok(t):
        return t
    buf = torch.zeros(n, (C + 3) // 4 * 4, dtype=t.dtype, device=t.device)
    buf[:, :C] = t
    return buf[:, :C]


def gemm_raw(a: Tensor, b: Tensor, trans_a: bool = False, trans_b: bool = False, bias: Tensor | None = None,
             alpha: float = 1.0, split_k: int | None = None, a_rows: Tensor | None = None, b_rows: Tensor | None = None,
             relu: bool = False, addend: Tensor | None = None) -> Tensor:
    """C = alpha * op(a) @ op(b) (+ bias) (+ addend [M,N], added in the kernel's store) via egnn_gemm_f32 / egnn_gemm_rows_f32.

    a_rows [M] (trans_a False): op(a) = a[a_rows];  b_rows [K] (trans_b False): b = b[b_rows] -- the gather is fused
    into the operand load."""
    _lib.require_gpu(a, b)
    a, b = _gemm_operand(a), _gemm_operand(b)
    if a_rows is not None:
        M, K = a_rows.numel(), a.shape[1]
    else:
        M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    if b_rows is not None:
        Kb, N = b_rows.numel(), b.shape[1]
    else:
        Kb, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if K != Kb:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if K == 0 or M == 0 or N == 0:
        # an empty reduction (dW = dY^T X[rows] on a shard that owns no train row) or an empty output: exact zeros (+ bias / addend),
        # no launch -- the entry points take non-null operands only
        c = torch.zeros(M, N, dtype=torch.float32, device=a.device)
        if bias is not None and M > 0 and N > 0:
            c += bias
        if addend is not None:
            c += addend
        return c.clamp_(min=0) if relu else c
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if split_k is None:
        # reductions over many rows into a small output (dW = X^T dY): spread K over the chip
        # (two workgroups per CU = 512 slots: 2 or 4 output tiles take 128 ranges -- 163 vs 178 us on 256 x 256 x 169 343, 106 vs 140 us on
        # 128 x 256 x 169 343; from 6 tiles on 64 ranges stay ahead: profiles/r05_splitk_sweep.txt)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        split_k = 1 if tiles >= 128 or K < 4096 else max(1, min(128 if tiles <= 4 else 64, 512 // max(tiles, 1), K // 1024))
    lib = _lib.load()
    ws = None
    # split-K partials, the skinny kernels' row-chunk partials, the bf16 planes of a small B operand (gemm_split.h)
    nws = lib.egnn_gemm_ws_floats(int(trans_a), int(trans_b), M, N, K, split_k)
    if a_rows is not None or b_rows is not None:
        nws = max(nws, split_k * M * N if split_k > 1 else 0)
    if nws > 0:
        ws = torch.empty(nws, dtype=torch.float32, device=a.device)
    if addend is not None:
        if a_rows is not None or b_rows is not None or relu or tuple(addend.shape) != (M, N):
            raise ValueError("gemm_raw: `addend` is an [M, N] matrix (no fused gather, no ReLU)")
        addend = _rowmajor(addend)
        rc = lib.egnn_gemm_add_f32(int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0),
                                   _lib.ptr(bias), _lib.ptr(addend), addend.stride(0), _lib.ptr(c), c.stride(0), split_k, _lib.ptr(ws),
                                   0 if ws is None else ws.numel() * 4, _lib.stream())
        _lib.check(rc, "egnn_gemm_add_f32")
    elif a_rows is None and b_rows is None:
        rc = lib.egnn_gemm_ex_f32(int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.ptr(a), a.stride(0), _lib.ptr(b),
                                  b.stride(0), _lib.ptr(bias), _lib.ptr(c), c.stride(0), split_k, _lib.ptr(ws),
                                  0 if ws is None else ws.numel() * 4, 1 if relu else 0, _lib.stream())
        _lib.check(rc, "egnn_gemm_ex_f32")
    else:
        _lib.require_gpu(*(t for t in (a_rows, b_rows) if t is not None))
        rc = lib.egnn_gemm_rows_f32(int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.ptr(a), a.stride(0), _lib.ptr(a_rows),
                                    _lib.ptr(b), b.stride(0), _lib.ptr(b_rows), _lib.ptr(bias), _lib.ptr(c), c.stride(0), split_k,
                                    _lib.ptr(ws), 0 if ws is None else ws.numel() * 4, _lib.stream())
        _lib.check(rc, "egnn_gemm_rows_f32")
        if relu:
            c.clamp_(min=0)
    return c


def bn_fold(weight: Tensor, bias: Tensor | None, bn: "torch.nn.BatchNorm1d"):
    """(W', b') with  relu(bn_eval(x W + b)) == relu(x W' + b')  (egnn_bn_fold_f32): an eval-mode BatchNorm1d folded into the
    [in, out] weight and the bias of the layer in front of it -- also across a (linear) aggregation between the two."""
    _lib.require_gpu(weight, bn.running_mean)
    w = _rowmajor(weight.detach())
    rows, C = w.shape
    wo = torch.empty(rows, C, dtype=torch.float32, device=w.device)
    bo = torch.empty(C, dtype=torch.float32, device=w.device)
    rc = _lib.load().egnn_bn_fold_f32(_lib.ptr(w), w.stride(0), rows, C, _lib.ptr(None if bias is None else bias.detach()),
                                      _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.ptr(None if bn.weight is None else bn.weight.detach()),
                                      _lib.ptr(None if bn.bias is None else bn.bias.detach()), float(bn.eps), _lib.ptr(wo), wo.stride(0), _lib.ptr(bo),
                                      _lib.stream())
    _lib.check(rc, "egnn_bn_fold_f32")
    return wo, bo


class _MatMul(torch.autograd.Function):
    """y = x @ w (+ bias); w stored [K,N] (``transposed=False``, GCNConv) or [N,K] (nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, w, bias, transposed):
        ctx.save_for_backward(x, w)
        ctx.transposed, ctx.has_bias = transposed, bias is not None
        ctx.tap_box = getattr(x, "_egnn_tap", None)
        return gemm_raw(x, w, False, transposed, bias)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _rowmajor(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm_raw(gy, w, False, not ctx.transposed)           # dX = dY W^T  (or dY W)
        if ctx.needs_input_grad[1]:
            gw = gemm_raw(gy, x, True, False) if ctx.transposed else gemm_raw(x, gy, True, False)  # dW = dY^T X | X^T dY
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(gy)
        return _fresh(gx, ctx.tap_box), gw, gb, None


def matmul(x: Tensor, w: Tensor, bias: Tensor | None = None) -> Tensor:
    """x [M,K] @ w [K,N] (+ bias [N], added in the GEMM's store)."""
    return _MatMul.apply(x, w, bias, False)


class _SageLayer(torch.autograd.Function):
    """SAGEConv (PyG <= 1.7: ``lin_l(aggr_j x_j) + lin_r(x_i)``; gnn.py:79-84) as ONE node of the autograd graph, so that
    neither ``lin_l(..) + lin_r(..)`` nor the two-path sum of the input gradient is an element-wise pass over [N, C]: the second
    product of each pair is added in the store of the kernel that forms it (GEMM ``addend`` / SpMM ``addend``).
    ``narrow``: aggregate ``x W_l^T`` instead of x (mean / sum are linear; taken when out < in: the gather moves `out` floats)."""

    @staticmethod
    def forward(ctx, x, adj, wl, bl, wr, reduce, narrow):
        ctx.tap_box = getattr(x, "_egnn_tap", None)
        x = _rowmajor(x)
        r = gemm_raw(x, wr, False, True)                                   # lin_r(x)
        if narrow:
            t = gemm_raw(x, wl, False, True)                               # x W_l^T, then aggregated with bias + lin_r(x) in the store
            out = spmm_raw(adj, t, reduce, bias=bl, addend=r)[0]
            agg = None
        else:
            agg = spmm_raw(adj, x, reduce)[0]
            out = gemm_raw(agg, wl, False, True, bias=bl, addend=r)        # lin_l(agg) + lin_r(x) in one store
        ctx.save_for_backward(x, wl, wr, *([] if agg is None else [agg]))
        ctx.adj, ctx.reduce, ctx.narrow, ctx.has_bias = adj, reduce, narrow, bl is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, wl, wr = ctx.saved_tensors[:3]
        agg = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        adj, reduce = ctx.adj, ctx.reduce
        g = _rowmajor(g)
        need_x = ctx.needs_input_grad[0]
        scale = adj._inv_rowcount() if reduce == "mean" else None          # dX = A^T (dY / cnt) for the mean
        gx = gwl = gbl = gwr = None
        if ctx.needs_input_grad[4]:
            gwr = gemm_raw(g, x, True, False)                              # dW_r = g^T x
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gbl = colsum(g)
        if ctx.narrow:
            need_t = need_x or ctx.needs_input_grad[2]
            dt = spmm_raw(adj.t(), g, "sum", src_scale=scale)[0] if need_t else None
            if ctx.needs_input_grad[2]:
                gwl = gemm_raw(dt, x, True, False)                         # dW_l = dt^T x
            if need_x:
                gx = gemm_raw(dt, wl, False, False, addend=gemm_raw(g, wr, False, False))     # dt W_l + g W_r
        else:
            if ctx.needs_input_grad[2]:
                gwl = gemm_raw(g, agg, True, False)                        # dW_l = g^T agg
            if need_x:
                d_agg = gemm_raw(g, wl, False, False)
                gx = spmm_raw(adj.t(), d_agg, "sum", src_scale=scale, addend=gemm_raw(g, wr, False, False))[0]   # A^T d_agg + g W_r
        return _fresh(gx, ctx.tap_box), None, gwl, gbl, gwr, None, None


class _LinearAdd(torch.autograd.Function):
    """x @ w^T + bias + addend with the sum formed in the GEMM's store (egnn_gemm_add_f32): SAGEConv's ``lin_l(agg) + lin_r(x)`` on
    node-range shards, where the aggregation is a collective operator of its own (dist._OverlapAggregate)."""

    @staticmethod
    def forward(ctx, x, w, bias, addend):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.tap_box = getattr(x, "_egnn_tap", None)
        return gemm_raw(x, w, False, True, bias, addend=addend)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _rowmajor(gy)
        gx = gemm_raw(gy, w, False, False) if ctx.needs_input_grad[0] else None
        gw = gemm_raw(gy, x, True, False) if ctx.needs_input_grad[1] else None
        gb = colsum(gy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return _fresh(gx, ctx.tap_box), gw, gb, (gy if ctx.needs_input_grad[3] else None)


def linear_add(x: Tensor, weight: Tensor, bias: Tensor | None, addend: Tensor) -> Tensor:
    """F.linear(x, weight, bias) + addend, one store."""
    _lib.require_gpu(x)
    if x.shape[0] == 0:
        return torch.nn.functional.linear(x, weight, bias) + addend
    return _LinearAdd.apply(x, weight, bias, addend)


def sage_layer(x: Tensor, adj, lin_l, lin_r, reduce: str, narrow: bool) -> Tensor:
    return _SageLayer.apply(x, adj, lin_l.weight, lin_l.bias, lin_r.weight, reduce, narrow)


class _TapBox:
    """Row-compact gradient pieces of one tapped tensor, waiting for its ``_GradTap.backward``."""
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = []


class _GradTap(torch.autograd.Function):
    """Identity with a side door for row-compact gradients.  The student's last hidden state h feeds the next conv AND
    ``student_proj(h[train_idx])`` (gnn.py:150-156); autograd would add the two gradients as dense [N,C] tensors (zero
    fill + scatter of the projection's rows + a full-size add: ~160 us at N = 169 343).  Consumers that only touch some
    rows (``linear_rows``) leave (ids, rows) in the box and return NO gradient for the tapped tensor -- the graph edge
    still orders them before this node -- and the rows are added into the dense gradient of the other consumer here."""

    @staticmethod
    def forward(ctx, x, box):
        ctx.box, ctx.meta = box, (x.shape, x.dtype, x.device)
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        pend, ctx.box.pending = ctx.box.pending, []
        if not pend:
            return g, None
        if g is None:
            shape, dtype, dev = ctx.meta
            g = torch.zeros(shape, dtype=dtype, device=dev)
        elif g.is_sparse or not g.is_contiguous():
            g = g.to_dense().contiguous() if g.is_sparse else g.contiguous()
        elif getattr(g, "_egnn_fresh_for", None) is not ctx.box:
            # autograd forbids changing a grad_output in place: a consumer whose backward hands its own grad_output through
            # (add, a view, identity) or a caller-supplied gradient would be corrupted.  Only a tensor one of this package's
            # backward functions has just allocated FOR THIS tapped tensor (``_fresh`` names the box) is added into directly.
            g = g.clone()
        for idx, rows in pend:   # unique ids: a plain read-modify-write of those rows, deterministic
            rows = _rowmajor(rows)
            _lib.check(_lib.load().egnn_rows_add_f32(_lib.ptr(g), g.stride(0), _lib.ptr(idx), _lib.ptr(rows), rows.stride(0), rows.shape[0],
                                                     rows.shape[1], _lib.stream()), "egnn_rows_add_f32")
        return g, None


_SPLIT_IDS = _cache.TensorKeyedCache(capacity=16)


def split_ids(split_idx: dict, n: int, device) -> Tensor:
    """int8 [n]: 0 / 1 / 2 for the train / valid / test nodes of ``split_idx``, -1 elsewhere (a node listed twice keeps the
    later split); built once per split dict (keyed on the identity and version of its index tensors, which the entry keeps alive)."""
    names = ("train", "valid", "test")

    def build():
        sid = torch.full((n,), -1, dtype=torch.int8, device=device)
        for i, k in enumerate(names):
            sid[split_idx[k].to(device)] = i
        return sid
    return _SPLIT_IDS.get(tuple(split_idx[k] for k in names), (n, str(device)), build)


def split_accuracy(logits: Tensor, y: Tensor, split_idx: dict, counts: bool = False, out: Tensor | None = None) -> Tensor:
    """float64 [3]: the Evaluator accuracies of test() (gnn.py:198-218) for train / valid / test in one pass over the
    logits (egnn_split_accuracy_f32: first-max argmax, integer hit counts over split sizes).
    ``counts``: float64 [6] = (hits of the three splits, sizes of the three splits) instead of the ratios (a shard's contribution
    to the all-rank accuracies)."""
    _lib.require_gpu(logits, y)
    logits = _rowmajor(logits)
    n, C = logits.shape
    yv = y.reshape(-1)
    if yv.dtype != torch.int64 or yv.numel() != n:
        raise TypeError("split_accuracy: labels must be int64 [n] or [n,1]")
    if n == 0:
        return torch.zeros(6, dtype=torch.float64, device=logits.device) if counts else torch.full((3,), float("nan"), dtype=torch.float64, device=logits.device)
    yv = yv.contiguous()
    sid = split_ids(split_idx, n, logits.device)
    lib = _lib.load()
    nws = lib.egnn_split_accuracy_ws_ints()
    ws = torch.empty(nws, dtype=torch.int32, device=logits.device)
    acc = out if out is not None else torch.empty(6 if counts else 3, dtype=torch.float64, device=logits.device)
    if acc.dtype != torch.float64 or acc.numel() != (6 if counts else 3) or not acc.is_contiguous() or acc.device != logits.device:
        raise TypeError("split_accuracy: `out` must be a contiguous float64 tensor of 3 (6 with counts) elements on the logits' device")
    fn, name = (lib.egnn_split_counts_f32, "egnn_split_counts_f32") if counts else (lib.egnn_split_accuracy_f32, "egnn_split_accuracy_f32")
    _lib.check(fn(_lib.ptr(logits), logits.stride(0), n, C, _lib.ptr(yv), _lib.ptr(sid), _lib.ptr(acc), _lib.ptr(ws), nws, _lib.stream()), name)
    return acc


def grad_tap(x: Tensor) -> Tensor:
    """``x`` again, marked so that ``linear_rows(x, idx, ...)`` hands its input gradient over in row-compact form."""
    if not (torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dim() == 2):
        return x
    box = _TapBox()
    y = _GradTap.apply(x, box)
    y._egnn_tap = box
    return y


_CONST_PLANES = _cache.TensorKeyedCache(capacity=4, max_bytes=2 << 30)    # plane images of constant operands: at most 2 GiB of the 288


def _dw_const_rows(gy: Tensor, x: Tensor, idx: Tensor):
    """dW = gy^T x[idx] for a CONSTANT x (the teacher's features under the teacher projection head, gnn.py:155,303-306): x[idx] is cut
    once per (x, idx) identity + version into bf16 planes with the row position as the reduction index (egnn_gemm_tn_planes_pack_f32,
    6 bytes per element: 419 MB for [90 941, 750], kept as long as the cache entry lives); the per-step product then takes the
    transposed-operand x planes form of the DMA pipeline (egnn_gemm_tn_planes_f32).  None = shape not taken (the caller's generic
    gather-fused GEMM)."""
    K, M = gy.shape
    N = x.shape[1]
    if not (M % 256 == 0 and K >= 16384 and gy.stride(0) % 4 == 0 and gy.data_ptr() % 16 == 0 and x.stride(1) == 1 and idx.numel() == K):
        return None
    lib = _lib.load()

    def build():
        nbytes = lib.egnn_gemm_tn_planes_bytes(N, K)
        planes = _aligned_bytes(nbytes, x.device)
        _lib.check(lib.egnn_gemm_tn_planes_pack_f32(_lib.ptr(x), x.stride(0), _lib.ptr(idx), N, K, _lib.ptr(planes), nbytes, _lib.stream()),
                   "egnn_gemm_tn_planes_pack_f32")
        return planes
    planes = _CONST_PLANES.get((x, idx), (), build)
    gw = torch.empty(M, N, dtype=torch.float32, device=gy.device)
    nws = lib.egnn_gemm_tn_planes_ws_floats(M, N, K)
    ws = torch.empty(nws, dtype=torch.float32, device=gy.device)
    rc = lib.egnn_gemm_tn_planes_f32(M, N, K, 1.0, _lib.ptr(gy), gy.stride(0), _lib.ptr(planes), _lib.ptr(gw), gw.stride(0), _lib.ptr(ws), nws,
                                     _lib.stream())
    if rc == _lib.EGNN_EALIGN:
        return None
    _lib.check(rc, "egnn_gemm_tn_planes_f32")
    return gw


_CONST_ROW_PLANES = _cache.TensorKeyedCache(capacity=4, max_bytes=2 << 30)


def _aligned_bytes(nbytes: int, device) -> Tensor:
    """uint8 [nbytes] starting on a 1 KB boundary (the DMA pipeline copies plane tiles in 1 KB pieces)."""
    buf = torch.empty(nbytes + 1024, dtype=torch.uint8, device=device)
    off = (-buf.data_ptr()) % 1024
    return buf[off:off + nbytes]


def _fwd_const_rows(x: Tensor, idx: Tensor, w: Tensor, bias: Tensor | None):
    """y = x[idx] @ w^T + bias for a CONSTANT x (see ``_dw_const_rows``): the gathered rows live as tile-packed bf16 planes, cut once
    per (x, idx); w is cut per call (egnn_gemm_rows_planes_f32: planes x planes on the DMA pipeline).  None = shape not taken."""
    M, K = idx.numel(), x.shape[1]
    N = w.shape[0]
    if not (N % 128 == 0 and M >= 16384 and x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K):
        return None
    lib = _lib.load()

    def build():
        nbytes = lib.egnn_gemm_rows_planes_bytes(M, K)
        planes = _aligned_bytes(nbytes, x.device)
        _lib.check(lib.egnn_gemm_rows_planes_pack_f32(_lib.ptr(x), x.stride(0), _lib.ptr(idx), M, K, _lib.ptr(planes), nbytes, _lib.stream()),
                   "egnn_gemm_rows_planes_pack_f32")
        return planes
    planes = _CONST_ROW_PLANES.get((x, idx), (), build)
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    nws = lib.egnn_gemm_rows_planes_ws_bytes(N, K)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    rc = lib.egnn_gemm_rows_planes_f32(M, N, K, 1.0, _lib.ptr(planes), _lib.ptr(w), w.stride(0), _lib.ptr(bias), _lib.ptr(y), y.stride(0),
                                       _lib.ptr(ws), nws, _lib.stream())
    if rc == _lib.EGNN_EALIGN:
        return None
    _lib.check(rc, "egnn_gemm_rows_planes_f32")
    return y


class _LinearRows(torch.autograd.Function):
    """y = x[idx] @ weight^T + bias for UNIQUE row ids, without materialising x[idx] (egnn_gemm_rows_f32): the forward
    gathers in the A-operand load, dW = dY^T x[idx] in the B-operand load, dx scatters the rows of dY W (or leaves
    them with the tap of a ``grad_tap`` tensor)."""

    @staticmethod
    def forward(ctx, x, idx, weight, bias, box, const_input=False):
        w = pad_pitch(weight) if not _pitch_ok(weight) else weight   # 0.8 MB at 256 x 750: rows 16-byte aligned
        ctx.save_for_backward(x, idx, w)
        ctx.has_bias, ctx.box = bias is not None, box
        # a CONSTANT input (the caller says so: the teacher's features, gnn.py:155): its gathered rows as planes, cut once per (x, idx)
        ctx.const_input = bool(const_input) and not x.requires_grad
        y = _fwd_const_rows(x, idx, w, bias) if ctx.const_input else None
        return y if y is not None else gemm_raw(x, w, False, True, bias, a_rows=idx)

    @staticmethod
    def backward(ctx, gy):
        x, idx, w = ctx.saved_tensors
        gy = _rowmajor(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            rows = gemm_raw(gy, w, False, False)
            if ctx.box is not None:
                ctx.box.pending.append((idx, rows))
            else:
                gx = torch.zeros(x.shape, dtype=gy.dtype, device=gy.device)
                gx.index_copy_(0, idx, rows)
        if ctx.needs_input_grad[2]:
            gw = _dw_const_rows(gy, x, idx) if ctx.const_input else None
            if gw is None:
                gw = gemm_raw(gy, x, True, False, b_rows=idx)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gb = colsum(gy)
        return gx, None, gw, gb, None, None


def linear_rows(x: Tensor, idx: Tensor, weight: Tensor, bias: Tensor | None = None, const_input: bool = False) -> Tensor:
    """``F.linear(x[idx], weight, bias)`` with the row gather fused into the GEMM (unique ``idx``).
    ``const_input=True`` (opt-in; the teacher projection head): ``x`` does not change between calls -- its gathered rows are re-laid
    ONCE per (x, idx) identity + version as tile-packed bf16 planes (6 bytes per element, held by a byte-bounded cache) and the forward /
    weight-gradient products run on the planes x planes forms.  Without the flag no plane image is ever built: activations under
    ``no_grad`` (requires_grad False as well) or RGCN's per-node-type calls take the gather-fused GEMM."""
    _lib.require_gpu(x)
    return _LinearRows.apply(x, idx, weight, bias, getattr(x, "_egnn_tap", None) if torch.is_grad_enabled() else None, const_input)


def linear(x: Tensor, weight: Tensor, bias: Tensor | None = None) -> Tensor:
    """x @ weight^T + bias with ``weight`` in nn.Linear layout [out, in]."""
    return _MatMul.apply(x, weight, bias, True)


# ------------------------------------------------------------------------------------------------
# cross entropy (+ logit KD)
# ------------------------------------------------------------------------------------------------
class _CeKd(torch.autograd.Function):
    """(mean CE, mean KD) of the rows ``rows`` (None: all) of logits / teacher with labels[rows] (fused row gathers)."""

    @staticmethod
    def forward(ctx, logits, labels, teacher, T, rows=None):
        _lib.require_gpu(logits, labels, teacher, rows)
        logits = _rowmajor(logits)
        teacher = None if teacher is None else _rowmajor(teacher)
        # the kernels index with the labels: PyTorch's own checks (class-index targets are int64 [n]) are kept on the host;
        # out-of-range values (incl. ignore_index = -100) are ignored inside the kernel, never used as an address
        if labels.dtype != torch.int64:
            raise TypeError(f"cross_entropy: expected int64 class-index labels, got {labels.dtype} "
                            "(multi-label float targets are the PPI path: criterion.ppi_kd_criterion)")
        if labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
            raise ValueError(f"cross_entropy: labels must have shape ({logits.shape[0]},), got {tuple(labels.shape)}")
        if teacher is not None and teacher.shape != logits.shape:
            raise ValueError(f"kd: teacher logits {tuple(teacher.shape)} do not match the student's {tuple(logits.shape)}")
        labels = labels.contiguous()
        if rows is not None:
            if rows.dtype != torch.int64 or rows.dim() != 1:
                raise TypeError("cross_entropy: `rows` must be a 1-D int64 index tensor")
            rows = rows.contiguous()
        n = logits.shape[0] if rows is None else rows.numel()
        C = logits.shape[1]
        lib = _lib.load()
        out = torch.empty(3, dtype=torch.float32, device=logits.device)
        ws = torch.empty(lib.egnn_ce_kd_ws_floats(n), dtype=torch.float32, device=logits.device)
        rc = lib.egnn_ce_kd_fwd_f32(_lib.ptr(logits), logits.stride(0), _lib.ptr(teacher), 0 if teacher is None else teacher.stride(0),
                                    _lib.ptr(labels), _lib.ptr(rows), n, C, float(T), _lib.ptr(out), _lib.ptr(ws), _lib.stream())
        _lib.check(rc, "egnn_ce_kd_fwd_f32")
        ctx.save_for_backward(logits, labels, out, *([] if teacher is None else [teacher]), *([] if rows is None else [rows]))
        ctx.T, ctx.has_teacher, ctx.has_rows, ctx.n = float(T), teacher is not None, rows is not None, n
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_kd):
        saved = list(ctx.saved_tensors)
        logits, labels, out = saved[:3]
        rest = saved[3:]
        teacher = rest.pop(0) if ctx.has_teacher else None
        rows = rest.pop(0) if ctx.has_rows else None
        n_total, C = logits.shape
        dl = torch.empty_like(logits)
        g_cls = None if g_cls is None else g_cls.contiguous().to(torch.float32)
        g_kd = None if g_kd is None else g_kd.contiguous().to(torch.float32)
        rc = _lib.load().egnn_ce_kd_bwd_f32(_lib.ptr(logits), logits.stride(0), _lib.ptr(teacher),
                                            0 if teacher is None else teacher.stride(0), _lib.ptr(labels), _lib.ptr(rows), n_total, ctx.n,
                                            C, ctx.T, _lib.ptr(out), _lib.ptr(g_cls), _lib.ptr(g_kd), _lib.ptr(dl), dl.stride(0), _lib.stream())
        _lib.check(rc, "egnn_ce_kd_bwd_f32")
        return dl, None, None, None, None


def cross_entropy(logits: Tensor, labels: Tensor, rows: Tensor | None = None) -> Tensor:
    """mean CE (F.cross_entropy) on the fused kernel; ``rows``: evaluate logits[rows] against labels[rows] without the copies."""
    return _CeKd.apply(logits, labels, None, 1.0, rows)[0]


def ce_and_kd(logits: Tensor, labels: Tensor, teacher_logits: Tensor, T: float, rows: Tensor | None = None):
    """(mean CE, F.kl_div(log_softmax(logits/T), softmax(teacher/T)) with reduction='mean'); ``rows`` as in cross_entropy."""
    return _CeKd.apply(logits, labels, teacher_logits, T, rows)


# ------------------------------------------------------------------------------------------------
# gather + L2 normalise
# ------------------------------------------------------------------------------------------------
class _GatherNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, eps):
        _lib.require_gpu(x, idx)
        x = _rowmajor(x)
        n = x.shape[0] if idx is None else idx.numel()
        D = x.shape[1]
        out = torch.empty(n, D, dtype=torch.float32, device=x.device)
        inv = torch.empty(n, dtype=torch.float32, device=x.device)
        rc = _lib.load().egnn_gather_normalize_rows_f32(_lib.ptr(x), x.stride(0), _lib.ptr(idx), n, D, float(eps), _lib.ptr(out),
                                                        out.stride(0), _lib.ptr(inv), _lib.stream())
        _lib.check(rc, "egnn_gather_normalize_rows_f32")
        ctx.save_for_backward(out, inv, *([] if idx is None else [idx]))
        ctx.eps, ctx.n_in = float(eps), x.shape[0]
        return out

    @staticmethod
    def backward(ctx, gout):
        saved = ctx.saved_tensors
        out, inv = saved[0], saved[1]
        idx = saved[2] if len(saved) > 2 else None
        gout = _rowmajor(gout)
        n, D = out.shape
        dx = torch.zeros(ctx.n_in, D, dtype=torch.float32, device=out.device) if idx is not None else torch.empty_like(out)
        rc = _lib.load().egnn_normalize_rows_bwd_f32(_lib.ptr(out), out.stride(0), _lib.ptr(gout), gout.stride(0), _lib.ptr(inv),
                                                     _lib.ptr(idx), n, D, ctx.eps, _lib.ptr(dx), dx.stride(0), 0, _lib.stream())
        _lib.check(rc, "egnn_normalize_rows_bwd_f32")
        return dx, None, None


def gather_normalize(x: Tensor, idx: Tensor | None = None, eps: float = 1e-12) -> Tensor:
    """F.normalize(x[idx], p=2, dim=-1) fused (rows of ``idx`` must be unique, as np.random.choice(replace=False) gives)."""
    return _GatherNormalize.apply(x, idx, eps)


# ------------------------------------------------------------------------------------------------
# FitNet / attention-transfer losses (criterion.py:24-54) as row kernels
# ------------------------------------------------------------------------------------------------
class _FeatureLoss(torch.autograd.Function):
    """kind 'fitnet': mse(F.normalize(f), F.normalize(t));  kind 'at': mse of the per-node energies normalised ACROSS nodes."""

    @staticmethod
    def forward(ctx, f, t, kind, eps):
        _lib.require_gpu(f, t)
        f, t = _rowmajor(f), _rowmajor(t)
        if f.dim() != 2 or t.dim() != 2 or f.shape[0] != t.shape[0] or (kind == "fitnet" and f.shape[1] != t.shape[1]):
            raise ValueError(f"{kind}: student features {tuple(f.shape)} and teacher features {tuple(t.shape)} do not match")
        n = f.shape[0]
        lib, dev = _lib.load(), f.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        nws = lib.egnn_feature_loss_ws_floats(n)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        if kind == "fitnet":
            rc = lib.egnn_fitnet_fwd_f32(_lib.ptr(f), f.stride(0), _lib.ptr(t), t.stride(0), n, f.shape[1], float(eps), _lib.ptr(loss),
                                         _lib.ptr(ws), nws, _lib.stream())
        else:
            rc = lib.egnn_at_fwd_f32(_lib.ptr(f), f.stride(0), f.shape[1], _lib.ptr(t), t.stride(0), t.shape[1], n, float(eps), _lib.ptr(loss),
                                     _lib.ptr(ws), nws, _lib.stream())
        _lib.check(rc, f"egnn_{kind}_fwd_f32")
        ctx.save_for_backward(f, t, ws)
        ctx.kind, ctx.eps = kind, float(eps)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        f, t, ws = ctx.saved_tensors
        n = f.shape[0]
        g = g.contiguous().to(torch.float32).reshape(1)
        df = torch.empty_like(f) if ctx.needs_input_grad[0] else None
        dt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        lib = _lib.load()
        if ctx.kind == "fitnet":
            rc = lib.egnn_fitnet_bwd_f32(_lib.ptr(f), f.stride(0), _lib.ptr(t), t.stride(0), n, f.shape[1], ctx.eps, _lib.ptr(g),
                                         _lib.ptr(df), 0 if df is None else df.stride(0), _lib.ptr(dt), 0 if dt is None else dt.stride(0),
                                         _lib.stream())
        else:
            rc = lib.egnn_at_bwd_f32(_lib.ptr(f), f.stride(0), f.shape[1], _lib.ptr(t), t.stride(0), t.shape[1], n, ctx.eps, _lib.ptr(ws),
                                     _lib.ptr(g), _lib.ptr(df), 0 if df is None else df.stride(0), _lib.ptr(dt),
                                     0 if dt is None else dt.stride(0), _lib.stream())
        _lib.check(rc, f"egnn_{ctx.kind}_bwd_f32")
        return df, dt, None, None


def fitnet_loss(feat: Tensor, teacher_feat: Tensor, eps: float = 1e-12) -> Tensor:
    """F.mse_loss(F.normalize(feat), F.normalize(teacher_feat)) in one pass per direction (egnn_fitnet_*_f32)."""
    return _FeatureLoss.apply(feat, teacher_feat, "fitnet", eps)


def at_loss(feat: Tensor, teacher_feat: Tensor, eps: float = 1e-12) -> Tensor:
    """criterion.py:44-50: per-node energies sum_d x^2, each length-n vector L2-normalised across the nodes, then mse."""
    return _FeatureLoss.apply(feat, teacher_feat, "at", eps)


# ------------------------------------------------------------------------------------------------
# G-CRD / InfoNCE on unit rows
# ------------------------------------------------------------------------------------------------
class _NCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fhat, that, tau):
        _lib.require_gpu(fhat, that)
        fhat, that = fhat.contiguous(), that.contiguous()
        S, P = fhat.shape
        if that.shape != fhat.shape:
            raise ValueError("nce: student and teacher features must have the same shape")
        lib = _lib.load()
        dev = fhat.device
        Z = torch.empty(S, S, dtype=torch.float32, device=dev)
        lse = torch.empty(S, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        nws = lib.egnn_nce_ws_floats(S)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        rc = lib.egnn_nce_fwd_f32(_lib.ptr(fhat), _lib.ptr(that), S, P, fhat.stride(0), float(tau), 1, _lib.ptr(Z), _lib.ptr(lse),
                                  _lib.ptr(loss), _lib.ptr(ws), nws, _lib.stream())
        _lib.check(rc, "egnn_nce_fwd_f32")
        ctx.save_for_backward(fhat, that, Z, lse)
        ctx.tau = float(tau)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        fhat, that, Z, lse = ctx.saved_tensors
        S, P = fhat.shape
        g = g.contiguous().to(torch.float32)
        df = torch.empty_like(fhat) if ctx.needs_input_grad[0] else None
        dt = torch.empty_like(that) if ctx.needs_input_grad[1] else None
        lib = _lib.load()
        nws = lib.egnn_nce_bwd_ws_floats(S, S, P)
        ws = torch.empty(nws, dtype=torch.float32, device=fhat.device)
        rc = lib.egnn_nce_bwd_f32(_lib.ptr(fhat), _lib.ptr(that), S, P, fhat.stride(0), ctx.tau, 1, _lib.ptr(Z), _lib.ptr(lse),
                                  _lib.ptr(g), _lib.ptr(df), _lib.ptr(dt), _lib.ptr(ws), nws, _lib.stream())
        _lib.check(rc, "egnn_nce_bwd_f32")
        return df, dt, None


def nce_unit(fhat: Tensor, that: Tensor, tau: float) -> Tensor:
    """mean_i(logsumexp_j(fhat_i . that_j / tau) - fhat_i . that_i / tau) for unit-norm rows."""
    return _NCE.apply(fhat, that, tau)


def nce_block_fwd(fhat: Tensor, t_all: Tensor, diag_off: int, tau: float, inv_count: float, unit_rows: bool = True):
    """Row block of the G-CRD loss (egnn_nce_block_fwd_f32): returns (Z [Sr,Sc], lse [Sr], loss_sum*inv_count [1]).
    Z is the saved score matrix for nce_block_bwd: logits, or exp(logit - 1.0001/tau) in the unit-rows form
    (egnn_nce_saves_exp); pass the same tau / unit_rows to the backward."""
    _lib.require_gpu(fhat, t_all)
    Sr, P = fhat.shape
    Sc = t_all.shape[0]
    lib, dev = _lib.load(), fhat.device
    Z = torch.empty(Sr, Sc, dtype=torch.float32, device=dev)
    lse = torch.empty(Sr, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nws = lib.egnn_nce_ws_floats(Sr)
    ws = torch.empty(nws, dtype=torch.float32, device=dev)
    rc = lib.egnn_nce_block_fwd_f32(_lib.ptr(fhat), fhat.stride(0), _lib.ptr(t_all), t_all.stride(0), Sr, Sc, diag_off, P, float(tau),
                                    float(inv_count), int(unit_rows), _lib.ptr(Z), _lib.ptr(lse), _lib.ptr(loss), _lib.ptr(ws), nws, _lib.stream())
    _lib.check(rc, "egnn_nce_block_fwd_f32")
    return Z, lse, loss


def nce_block_bwd(fhat: Tensor, t_all: Tensor, diag_off: int, scale: float, Z: Tensor, lse: Tensor, g: Tensor, tau: float,
                  unit_rows: bool = True):
    """(dfhat [Sr,P], this rank's contribution to dthat_all [Sc,P]) via egnn_nce_block_bwd_f32."""
    Sr, P = fhat.shape
    Sc = t_all.shape[0]
    df, dt = torch.empty_like(fhat), torch.empty_like(t_all)
    nws = _lib.load().egnn_nce_bwd_ws_floats(Sr, Sc, P)
    ws = torch.empty(nws, dtype=torch.float32, device=fhat.device)
    rc = _lib.load().egnn_nce_block_bwd_f32(_lib.ptr(fhat), fhat.stride(0), _lib.ptr(t_all), t_all.stride(0), Sr, Sc, diag_off, P,
                                            float(tau), float(scale), int(unit_rows), _lib.ptr(Z), _lib.ptr(lse), _lib.ptr(g), _lib.ptr(df), df.stride(0),
                                            _lib.ptr(dt), dt.stride(0), _lib.ptr(ws), nws, _lib.stream())
    _lib.check(rc, "egnn_nce_block_bwd_f32")
    return df, dt


# ------------------------------------------------------------------------------------------------
# fused BatchNorm1d (+ ReLU + dropout)
# ------------------------------------------------------------------------------------------------
# Optional per-step dropout seed kept ON THE DEVICE (int64 [1]): the kernels add it to the per-call host seed.  A captured
# hipGraph of the train step (models.GraphedEpoch) refreshes it before every replay, so that replays draw fresh masks.
_DROPOUT_SEED_DEV: Tensor | None = None


def _draw_dropout_seed() -> int:
    """The per-call 63-bit seed of a fused dropout mask, from torch's HOST generator (``torch.manual_seed`` reproducible).  The mask of
    element (r, c) is a counter hash of (seed [+ the device-side per-step seed], r * C + c): csrc/bn_common.h.  One function so that
    the training-parity harness (oracle/training_parity.py) can record the seeds and rebuild the masks for the oracle."""
    return int(torch.empty((), dtype=torch.int64).random_()) & 0x7FFFFFFFFFFFFFFF


def _bn_shape_ok(x: Tensor) -> bool:
    C = x.shape[1]
    return x.is_cuda and C % 4 == 0 and C <= 1024 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0


class _BnAct(torch.autograd.Function):
    """``pick`` (int64 [S], unique row ids) = form only the output rows ``pick``: y[i] = act(bn(x[pick[i]])) -- the statistics still span
    all rows of x, and so does dx (egnn_bn_act_rows_*_f32)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, relu, p, seed, batch_stats, pick=None):
        x = _rowmajor(x)
        n, C = x.shape
        seed_dev = _DROPOUT_SEED_DEV if p > 0 else None
        if pick is None:
            y = torch.empty(n, C, dtype=torch.float32, device=x.device)
            rc = _lib.load().egnn_bn_act_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), float(eps), _lib.ptr(gamma),
                                                 _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev), _lib.ptr(y), y.stride(0),
                                                 _lib.stream())
            _lib.check(rc, "egnn_bn_act_fwd_f32")
        else:
            if pick.dtype != torch.int64 or pick.dim() != 1 or not pick.is_contiguous() or pick.numel() > n or pick.numel() == 0:
                raise ValueError("bn_act: `pick` must be a non-empty contiguous 1-D int64 tensor of unique row ids")
            y = torch.empty(pick.numel(), C, dtype=torch.float32, device=x.device)
            rc = _lib.load().egnn_bn_act_rows_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(pick), pick.numel(), _lib.ptr(mean), _lib.ptr(var),
                                                      float(eps), _lib.ptr(gamma), _lib.ptr(beta), int(relu), float(p), int(seed),
                                                      _lib.ptr(seed_dev), _lib.ptr(y), y.stride(0), _lib.stream())
            _lib.check(rc, "egnn_bn_act_rows_fwd_f32")
        ctx.save_for_backward(x, gamma, beta, mean, var, *([] if pick is None else [pick]))
        ctx.cfg = (float(eps), int(relu), float(p), int(seed), int(batch_stats))
        ctx.seed_dev = seed_dev
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, var = ctx.saved_tensors[:5]
        pick = ctx.saved_tensors[5] if len(ctx.saved_tensors) > 5 else None
        eps, relu, p, seed, batch_stats = ctx.cfg
        gy = _rowmajor(gy)
        n, C = x.shape
        lib, dev = _lib.load(), x.device
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        nws = lib.egnn_bn_ws_floats(C)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        # the producer of x usually added a bias (GCNConv / nn.Linear in front of the BatchNorm): its gradient is the column
        # sum of dx, formed in the same pass (ops.colsum picks the tag up)
        cs = torch.empty(C, dtype=torch.float32, device=dev) if batch_stats else None
        if pick is None:
            rc = lib.egnn_bn_act_bwd_colsum_f32(_lib.ptr(x), x.stride(0), _lib.ptr(gy), gy.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), eps,
                                                _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev), batch_stats,
                                                _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dx), dx.stride(0), _lib.ptr(cs), _lib.ptr(ws), nws,
                                                _lib.stream())
            _lib.check(rc, "egnn_bn_act_bwd_colsum_f32")
        else:
            rc = lib.egnn_bn_act_rows_bwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(pick), pick.numel(), _lib.ptr(gy), gy.stride(0),
                                              _lib.ptr(mean), _lib.ptr(var), eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed,
                                              _lib.ptr(ctx.seed_dev), batch_stats, _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dx), dx.stride(0),
                                              _lib.ptr(cs), _lib.ptr(ws), nws, _lib.stream())
            _lib.check(rc, "egnn_bn_act_rows_bwd_f32")
        if cs is not None:
            dx._egnn_colsum = (cs, dx._version)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def _bn_prepare(x: Tensor, bn, p: float, training: bool):
    """Statistics + module state update of a fused BatchNorm call; None when the kernels do not take the shape (torch fallback).
    Returns (x row-major, mean, var, use_batch, drop, seed)."""
    x_in = x
    if not _bn_shape_ok(_rowmajor(x)) or not bn.track_running_stats or bn.weight is None:
        return None
    x = _rowmajor(x)
    n, C = x.shape
    use_batch = training  # nn.BatchNorm1d: batch statistics in training mode, running statistics in eval mode
    if use_batch:
        lib, dev = _lib.load(), x.device
        pre = getattr(x_in, "_egnn_bn_stats", None)   # formed in the producing aggregation's epilogue (ops.spmm)
        if pre is not None and pre[0].shape[0] == C and pre[2] == x_in._version:
            mean, var = pre[0], pre[1]
        else:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            var = torch.empty(C, dtype=torch.float32, device=dev)
            nws = lib.egnn_bn_ws_floats(C)
            ws = torch.empty(nws, dtype=torch.float32, device=dev)
            _lib.check(lib.egnn_bn_stats_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), _lib.ptr(ws), nws, _lib.stream()),
                       "egnn_bn_stats_f32")
        with torch.no_grad():   # nn.BatchNorm1d's state update, one launch (momentum=None: the cumulative average, formed on the device)
            rm, rv, nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
            if rm.dtype == torch.float32 and rv.dtype == torch.float32 and nbt.dtype == torch.int64 and rm.is_contiguous() and rv.is_contiguous():
                _lib.check(lib.egnn_bn_running_update_f32(_lib.ptr(mean), _lib.ptr(var), C, n, -1.0 if bn.momentum is None else float(bn.momentum),
                                                          _lib.ptr(rm), _lib.ptr(rv), _lib.ptr(nbt), _lib.stream()), "egnn_bn_running_update_f32")
            else:
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var, alpha=m * n / max(n - 1, 1))
    else:
        mean, var = bn.running_mean, bn.running_var
    drop = p if (training and p > 0) else 0.0
    seed = _draw_dropout_seed() if drop > 0 else 0
    return x, mean, var, use_batch, drop, seed


def bn_act(x: Tensor, bn: "torch.nn.BatchNorm1d", relu: bool = True, p: float = 0.0, training: bool | None = None,
           pick: Tensor | None = None) -> Tensor:
    """dropout(relu(bn(x)), p) in two kernels (statistics + apply); same module state updates as nn.BatchNorm1d.
    ``pick``: return only the rows ``pick`` of that result (unique int64 ids; the statistics still span all rows of x).

    Falls back to the torch operators (still on the GPU) for shapes the kernel does not take (C % 4 != 0, C > 1024)."""
    training = bn.training if training is None else training
    prep = _bn_prepare(x, bn, p, training)
    if prep is None:
        y = bn(x)
        y = torch.relu(y) if relu else y
        y = torch.nn.functional.dropout(y, p, training) if p > 0 else y
        return y if pick is None else y[pick]
    x, mean, var, use_batch, drop, seed = prep
    return _BnAct.apply(x, bn.weight, bn.bias, mean, var, bn.eps, relu, drop, seed, use_batch, pick)


_INV_ROWS = _cache.TensorKeyedCache(capacity=16)


def _inverse_rows(idx: Tensor, n: int) -> Tensor:
    """int32 [n]: position of row r in the unique id list ``idx``, -1 where r is not listed.  Built once per index tensor (identity +
    version + length: the train split does not change between steps); the entry keeps ``idx`` alive (_cache.py)."""
    def build():
        inv = torch.full((n,), -1, dtype=torch.int32, device=idx.device)
        inv[idx] = torch.arange(idx.numel(), dtype=torch.int32, device=idx.device)
        return inv
    return _INV_ROWS.get((idx,), (n,), build)


class _BnActLinear(torch.autograd.Function):
    """(h, h @ w) with h = drop(relu(bn(x))) and a NARROW w [C, Ks] (the class count): the student's last hidden layer feeding its
    output conv (gnn.py:47-52).  h also carries the gradient tap of ``grad_tap`` (the projection head's row-compact input gradient).
    Backward: dh = G w^T + tap rows (+ any dense gradient of h) is formed in MFMA tiles and goes through the BatchNorm backward
    without being stored (egnn_skinny_dx_bn_bwd_f32): one pass over [n, C] instead of four."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, relu, p, seed, batch_stats, w, box):
        x = _rowmajor(x)
        n, C = x.shape
        h = torch.empty(n, C, dtype=torch.float32, device=x.device)
        seed_dev = _DROPOUT_SEED_DEV if p > 0 else None
        w = _rowmajor(w)
        lib = _lib.load()
        rc = _lib.EGNN_EALIGN
        if w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0:
            # one pass over x: h is stored and multiplied by w while its pieces are in registers
            xw = torch.empty(n, w.shape[1], dtype=torch.float32, device=x.device)
            rc = lib.egnn_bn_act_linear_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), float(eps), _lib.ptr(gamma),
                                                _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev), _lib.ptr(w), w.stride(0), 0,
                                                w.shape[1], _lib.ptr(h), h.stride(0), _lib.ptr(xw), xw.stride(0), _lib.stream())
            if rc != _lib.EGNN_EALIGN:
                _lib.check(rc, "egnn_bn_act_linear_fwd_f32")
        if rc == _lib.EGNN_EALIGN:
            rc = lib.egnn_bn_act_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), float(eps), _lib.ptr(gamma),
                                         _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev), _lib.ptr(h), h.stride(0),
                                         _lib.stream())
            _lib.check(rc, "egnn_bn_act_fwd_f32")
            xw = gemm_raw(h, w, False, False)
        ctx.save_for_backward(x, gamma, beta, mean, var, h, w)
        ctx.cfg = (float(eps), int(relu), float(p), int(seed), int(batch_stats))
        ctx.seed_dev, ctx.box, ctx.w_index = seed_dev, box, 10
        ctx.set_materialize_grads(False)
        return h, xw

    @staticmethod
    def backward(ctx, g_h, g_xw):
        x, gamma, beta, mean, var, h, w = ctx.saved_tensors
        eps, relu, p, seed, batch_stats = ctx.cfg
        dx, dgamma, dbeta, gw = _tail_backward(ctx, g_h, g_xw, x, gamma, beta, mean, var, h, w, eps, relu, p, seed, batch_stats, None)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, gw, None


def _tail_backward(ctx, g_h, g_xw, x, gamma, beta, mean, var, h, w, eps, relu, p, seed, batch_stats, sync):
    """Backward of (h, h @ w) = tail(x): dW = h^T G; dh = G w^T (+ tap rows, + dense g_h) through the BatchNorm backward.
    ``sync`` = None: statistics of this tensor alone (1 / n).  ``sync`` = (total rows [1] on the device, group): the column sums
    (sum d, sum d xhat) are all-reduced between the reduce half and the apply half (SyncBN on node-range shards); the returned
    dgamma / dbeta are then this shard's LOCAL sums (the flat gradient all-reduce adds the shards)."""
    pend = []
    if ctx.box is not None:
        pend, ctx.box.pending = ctx.box.pending, []
        pend = [(i, r) for i, r in pend if r.shape[0] > 0]      # (a shard without train rows taps zero rows: nothing to add)
    n, C = x.shape
    Ks = w.shape[1]
    lib, dev = _lib.load(), x.device
    gw = None
    if g_xw is not None:
        g_xw = _rowmajor(g_xw)
        if ctx.needs_input_grad[ctx.w_index]:
            gw = gemm_raw(h, g_xw, True, False) if n > 0 else torch.zeros_like(w)     # dW = h^T G
    if not ctx.needs_input_grad[0]:
        return None, None, None, gw
    dx = torch.empty_like(x)
    sums = torch.empty(2 * C, dtype=torch.float32, device=dev)       # [dbeta | dgamma]; written by the reduce half (zeroed on an empty shard)
    if n == 0:
        sums.zero_()
    dbeta, dgamma = sums[:C], sums[C:]
    cs = torch.empty(C, dtype=torch.float32, device=dev) if (batch_stats and n > 0) else None
    if g_h is not None:
        g_h = _rowmajor(g_h.to_dense() if g_h.is_sparse else g_h)
    fused = (n > 0 and g_xw is not None and len(pend) <= 1 and C % 64 == 0 and Ks <= 64 and x.stride(0) % 4 == 0
             and (g_h is None or (g_h.stride(0) % 4 == 0 and g_h.data_ptr() % 16 == 0)))
    rows = inv = None
    if fused and pend:
        idx, rows = pend[0]
        rows = _rowmajor(rows)
        fused = rows.stride(0) % 4 == 0 and rows.data_ptr() % 16 == 0 and rows.shape[1] == C
        if fused:
            inv = _inverse_rows(idx, n)

    def apply_sums():
        """(sum_dbeta, sum_dgamma, inv_count) the apply half uses; the all-rank sum on shards."""
        if sync is None:
            return dbeta, dgamma, (1.0 / n if batch_stats else 0.0)
        import torch.distributed as dist
        total, group = sync
        red = sums
        if dist.get_world_size(group) > 1:
            red = sums.clone()
            dist.all_reduce(red, group=group)
        red = red / total                                          # scaled on the device (no host read of the row count)
        return red[:C], red[C:], 1.0

    done = False
    if fused:
        nws = lib.egnn_skinny_dx_bn_ws_floats(n, C)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        rc = lib.egnn_skinny_dx_bn_bwd_reduce_f32(_lib.ptr(g_xw), g_xw.stride(0), _lib.ptr(w), w.stride(0), 0, n, C, Ks, 1.0,
                                                  _lib.ptr(g_h), 0 if g_h is None else g_h.stride(0), _lib.ptr(rows),
                                                  0 if rows is None else rows.stride(0), _lib.ptr(inv), _lib.ptr(x), x.stride(0), _lib.ptr(mean),
                                                  _lib.ptr(var), eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev),
                                                  _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dx), dx.stride(0), _lib.ptr(ws), nws, _lib.stream())
        if rc != _lib.EGNN_EALIGN:     # EGNN_EALIGN: beyond the kernel's 32-bit element offsets (n * ld >= 2^31) -> the separate passes below
            _lib.check(rc, "egnn_skinny_dx_bn_bwd_reduce_f32")
            sb, sg, inv_count = apply_sums()
            _lib.check(lib.egnn_bn_bwd_apply_stored_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), eps, _lib.ptr(gamma),
                                                        _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev), _lib.ptr(sb), _lib.ptr(sg),
                                                        inv_count, _lib.ptr(dx), dx.stride(0), _lib.ptr(cs), _lib.ptr(ws), nws, _lib.stream()),
                       "egnn_bn_bwd_apply_stored_f32")
            done = True
    if not done:
        # the separate passes: dense dh, the tap rows added into it, the BatchNorm backward (reduce, [all-reduce,] apply)
        dh = None
        if n > 0:
            dh = gemm_raw(g_xw, w, False, True) if g_xw is not None else None
            if dh is None:
                dh = g_h.clone() if g_h is not None else torch.zeros_like(x)
            elif g_h is not None:
                dh = dh + g_h
            for idx, rows in pend:
                rows = _rowmajor(rows)
                _lib.check(lib.egnn_rows_add_f32(_lib.ptr(dh), dh.stride(0), _lib.ptr(idx), _lib.ptr(rows), rows.stride(0), rows.shape[0],
                                                 rows.shape[1], _lib.stream()), "egnn_rows_add_f32")
            nws = lib.egnn_bn_ws_floats(C)
            ws = torch.empty(nws, dtype=torch.float32, device=dev)
            _lib.check(lib.egnn_bn_act_bwd_reduce_f32(_lib.ptr(x), x.stride(0), _lib.ptr(dh), dh.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var),
                                                      eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev), _lib.ptr(dgamma),
                                                      _lib.ptr(dbeta), _lib.ptr(ws), nws, _lib.stream()), "egnn_bn_act_bwd_reduce_f32")
        sb, sg, inv_count = apply_sums()
        if n > 0 and cs is not None:
            # apply half + the column sums of dx (the bias gradient of the conv in front) in the same pass, as on the fused route
            _lib.check(lib.egnn_bn_act_bwd_apply_colsum_f32(_lib.ptr(x), x.stride(0), _lib.ptr(dh), dh.stride(0), n, C, _lib.ptr(mean),
                                                            _lib.ptr(var), eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed,
                                                            _lib.ptr(ctx.seed_dev), _lib.ptr(sb), _lib.ptr(sg), inv_count, _lib.ptr(dx),
                                                            dx.stride(0), _lib.ptr(cs), _lib.ptr(ws), nws, _lib.stream()),
                       "egnn_bn_act_bwd_apply_colsum_f32")
        elif n > 0:
            _lib.check(lib.egnn_bn_act_bwd_apply_f32(_lib.ptr(x), x.stride(0), _lib.ptr(dh), dh.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var),
                                                     eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev), _lib.ptr(sb),
                                                     _lib.ptr(sg), inv_count, _lib.ptr(dx), dx.stride(0), _lib.stream()), "egnn_bn_act_bwd_apply_f32")
    if cs is not None:
        dx._egnn_colsum = (cs, dx._version)
    return dx, dgamma, dbeta, gw


def bn_act_linear(x: Tensor, bn: "torch.nn.BatchNorm1d", w: Tensor, relu: bool = True, p: float = 0.0, training: bool | None = None):
    """(h, h @ w) for h = dropout(relu(bn(x)), p) and a narrow ``w`` [C, Ks <= 64] -- ``bn_act`` + ``grad_tap`` + ``matmul`` with ONE
    backward pass over the [n, C] tensors (see _BnActLinear).  h carries the tap for ``linear_rows``.  None when the fused kernels do
    not take the shape (the caller then composes the three calls)."""
    training = bn.training if training is None else training
    if not (torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and w.dim() == 2 and w.shape[0] == x.shape[1] and w.shape[1] <= 64
            and x.shape[1] % 64 == 0):
        return None
    prep = _bn_prepare(x, bn, p, training)
    if prep is None:
        return None
    x, mean, var, use_batch, drop, seed = prep
    box = _TapBox()
    h, xw = _BnActLinear.apply(x, bn.weight, bn.bias, mean, var, bn.eps, relu, drop, seed, use_batch, w, box)
    h._egnn_tap = box
    return h, xw


# ------------------------------------------------------------------------------------------------
# the same fused BatchNorm + ReLU + dropout with batch statistics that span all ranks (node-range shards)
# ------------------------------------------------------------------------------------------------
_ROW_COUNTS: dict = {}


def _row_count(n: int, dev) -> Tensor:
    """float32 [1] device constant holding ``n`` (made once per value and device, never written)."""
    key = (int(n), str(dev))
    t = _ROW_COUNTS.get(key)
    if t is None:
        # never evicted (4 bytes each): captured graphs read these constants through raw pointers for as long as they live
        t = _ROW_COUNTS[key] = torch.full((1,), float(n), dtype=torch.float32, device=dev)
    return t


def _sync_stats(x: Tensor, group):
    """(mean, biased var, total rows [1]) of the rows of ALL ranks: the per-shard (mean, var, n) triples are all-gathered and merged in
    rank order (Chan's parallel-variance formula, identical on every rank).  An empty shard contributes n = 0."""
    import torch.distributed as dist
    n, C = x.shape
    lib, dev = _lib.load(), x.device
    world = dist.get_world_size(group)
    stats = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)  # [mean | var | n]
    if n == 0:
        stats.zero_()
    else:
        nws = lib.egnn_bn_ws_floats(C)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        _lib.check(lib.egnn_bn_stats_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(stats), _lib.ptr(stats[C:]), _lib.ptr(ws), nws,
                                         _lib.stream()), "egnn_bn_stats_f32")
    if world == 1:
        return stats[:C], stats[C:2 * C], _row_count(n, dev)          # (a cached constant: no fill launch per call)
    if n > 0:
        stats[2 * C:].fill_(float(n))     # a fill kernel, not a host->device copy
    allst = torch.empty(world, 2 * C + 1, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(allst, stats, group=group)
    merged = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
    mean, var, total = merged[:C], merged[C:2 * C], merged[2 * C:]
    _lib.check(lib.egnn_bn_merge_shards_f32(_lib.ptr(allst), world, C, _lib.ptr(mean), _lib.ptr(var), _lib.ptr(total), _lib.stream()),
               "egnn_bn_merge_shards_f32")
    return mean, var, total


class _SyncBnAct(torch.autograd.Function):
    """Two small collectives per direction: the per-shard (n, mean, var) triples are all-gathered and merged (``_sync_stats``); the
    backward all-reduces [sum d, sum d*xhat].  ``pick`` (unique int64 row ids, may be empty): only those output rows are formed
    (egnn_bn_act_rows_fwd_f32) -- the statistics, and dx, still span every row of every rank (the projection heads under a sampled
    criterion, gnn.py:296-306 -> criterion.py:62-65,134-137)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, p, seed, group, pick=None):
        x = _rowmajor(x)
        n, C = x.shape
        lib, dev = _lib.load(), x.device
        mean, var, total = _sync_stats(x, group)
        seed_dev = _DROPOUT_SEED_DEV if p > 0 else None     # the per-step seed of a replayed graph (fresh masks in every replay)
        if pick is None:
            y = torch.empty(n, C, dtype=torch.float32, device=dev)
            if n > 0:
                rc = lib.egnn_bn_act_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), float(eps), _lib.ptr(gamma),
                                             _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev), _lib.ptr(y), y.stride(0), _lib.stream())
                _lib.check(rc, "egnn_bn_act_fwd_f32")
        else:
            if pick.dtype != torch.int64 or pick.dim() != 1 or not pick.is_contiguous() or pick.numel() > n:
                raise ValueError("sync_bn_act: `pick` must be a contiguous 1-D int64 tensor of unique row ids")
            y = torch.empty(pick.numel(), C, dtype=torch.float32, device=dev)
            if pick.numel() > 0:
                rc = lib.egnn_bn_act_rows_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(pick), pick.numel(), _lib.ptr(mean), _lib.ptr(var),
                                                  float(eps), _lib.ptr(gamma), _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev),
                                                  _lib.ptr(y), y.stride(0), _lib.stream())
                _lib.check(rc, "egnn_bn_act_rows_fwd_f32")
        ctx.save_for_backward(x, gamma, beta, mean, var, total, *([] if pick is None else [pick]))
        ctx.cfg = (float(eps), int(relu), float(p), int(seed), group)
        ctx.seed_dev = seed_dev
        ctx.mark_non_differentiable(mean, var, total)
        return y, mean, var, total

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gt):
        import torch.distributed as dist
        x, gamma, beta, mean, var, total = ctx.saved_tensors[:6]
        pick = ctx.saved_tensors[6] if len(ctx.saved_tensors) > 6 else None
        eps, relu, p, seed, group = ctx.cfg
        gy = _rowmajor(gy)
        n, C = x.shape
        lib, dev = _lib.load(), x.device
        n_red = n if pick is None else pick.numel()                  # rows that carry a gradient
        sums = torch.empty(2 * C, dtype=torch.float32, device=dev)   # [dbeta | dgamma] of this shard
        nws = lib.egnn_bn_ws_floats(C)
        ws = torch.empty(nws, dtype=torch.float32, device=dev) if n > 0 else None
        if n_red > 0 and pick is None:
            rc = lib.egnn_bn_act_bwd_reduce_f32(_lib.ptr(x), x.stride(0), _lib.ptr(gy), gy.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var),
                                                eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev), _lib.ptr(sums[C:]),
                                                _lib.ptr(sums), _lib.ptr(ws), nws, _lib.stream())
            _lib.check(rc, "egnn_bn_act_bwd_reduce_f32")
        elif n_red > 0:
            rc = lib.egnn_bn_act_rows_bwd_reduce_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(pick), n_red, _lib.ptr(gy), gy.stride(0),
                                                     _lib.ptr(mean), _lib.ptr(var), eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed,
                                                     _lib.ptr(ctx.seed_dev), _lib.ptr(sums[C:]), _lib.ptr(sums), _lib.ptr(ws), nws, _lib.stream())
            _lib.check(rc, "egnn_bn_act_rows_bwd_reduce_f32")
        else:
            sums.zero_()
        local = sums                                                 # parameter grads stay local (the flat all-reduce sums them)
        if dist.get_world_size(group) > 1:
            sums = sums.clone()
            dist.all_reduce(sums, group=group)
        scaled = sums / total                                        # scaled on the device (no host read of the row count)
        dx = torch.empty_like(x)
        if n > 0:
            # the column sums of dx (bias gradient of the conv / Linear in front) come out of the same pass (ops.colsum picks the tag up)
            cs = torch.empty(C, dtype=torch.float32, device=dev)
            if pick is None:
                rc = lib.egnn_bn_act_bwd_apply_colsum_f32(_lib.ptr(x), x.stride(0), _lib.ptr(gy), gy.stride(0), n, C, _lib.ptr(mean),
                                                          _lib.ptr(var), eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed, _lib.ptr(ctx.seed_dev),
                                                          _lib.ptr(scaled), _lib.ptr(scaled[C:]), 1.0, _lib.ptr(dx), dx.stride(0), _lib.ptr(cs),
                                                          _lib.ptr(ws), nws, _lib.stream())
                _lib.check(rc, "egnn_bn_act_bwd_apply_colsum_f32")
            else:
                rc = lib.egnn_bn_act_rows_bwd_apply_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(pick), n_red, _lib.ptr(gy), gy.stride(0),
                                                        _lib.ptr(mean), _lib.ptr(var), eps, _lib.ptr(gamma), _lib.ptr(beta), relu, p, seed,
                                                        _lib.ptr(ctx.seed_dev), _lib.ptr(scaled), _lib.ptr(scaled[C:]), 1.0, _lib.ptr(local),
                                                        _lib.ptr(dx), dx.stride(0), _lib.ptr(cs), _lib.ptr(ws), nws, _lib.stream())
                _lib.check(rc, "egnn_bn_act_rows_bwd_apply_f32")
            dx._egnn_colsum = (cs, dx._version)
        return dx, local[C:], local[:C], None, None, None, None, None, None


def sync_bn_act(x: Tensor, bn, relu: bool, p: float, training: bool, group=None, pick: Tensor | None = None):
    """Training-mode dropout(relu(bn(x))) with all-rank statistics; returns (y, mean, biased var, total rows) so that the
    module can update its running statistics.  ``bn`` needs weight / bias / eps (dist.SyncBatchNorm1d).  ``pick``: only those rows of
    the result (see _SyncBnAct)."""
    drop = p if (training and p > 0) else 0.0
    seed = _draw_dropout_seed() if drop > 0 else 0
    return _SyncBnAct.apply(x, bn.weight, bn.bias, bn.eps, relu, drop, seed, group, pick)


class _SyncBnActLinear(torch.autograd.Function):
    """``_BnActLinear`` with all-rank batch statistics: (h, h @ w, mean, var, total) for h = drop(relu(bn(x))) on a node-range shard.
    Forward: statistics all-gather, then the one-pass tail kernel; backward: the tail's reduce half, ONE all-reduce of
    [sum d | sum d xhat], the apply half (egnn_skinny_dx_bn_bwd_reduce_f32 / egnn_bn_bwd_apply_stored_f32)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, p, seed, w, box, group):
        x = _rowmajor(x)
        n, C = x.shape
        lib, dev = _lib.load(), x.device
        mean, var, total = _sync_stats(x, group)
        seed_dev = _DROPOUT_SEED_DEV if p > 0 else None
        w = _rowmajor(w)
        h = torch.empty(n, C, dtype=torch.float32, device=dev)
        xw = torch.empty(n, w.shape[1], dtype=torch.float32, device=dev)
        if n > 0:
            rc = _lib.EGNN_EALIGN
            if w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0:
                rc = lib.egnn_bn_act_linear_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), float(eps), _lib.ptr(gamma),
                                                    _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev), _lib.ptr(w), w.stride(0), 0,
                                                    w.shape[1], _lib.ptr(h), h.stride(0), _lib.ptr(xw), xw.stride(0), _lib.stream())
                if rc != _lib.EGNN_EALIGN:
                    _lib.check(rc, "egnn_bn_act_linear_fwd_f32")
            if rc == _lib.EGNN_EALIGN:
                _lib.check(lib.egnn_bn_act_fwd_f32(_lib.ptr(x), x.stride(0), n, C, _lib.ptr(mean), _lib.ptr(var), float(eps), _lib.ptr(gamma),
                                                   _lib.ptr(beta), int(relu), float(p), int(seed), _lib.ptr(seed_dev), _lib.ptr(h), h.stride(0),
                                                   _lib.stream()), "egnn_bn_act_fwd_f32")
                xw = gemm_raw(h, w, False, False)
        ctx.save_for_backward(x, gamma, beta, mean, var, h, w, total)
        ctx.cfg = (float(eps), int(relu), float(p), int(seed), group)
        ctx.seed_dev, ctx.box, ctx.w_index = seed_dev, box, 7
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(mean, var, total)
        return h, xw, mean, var, total

    @staticmethod
    def backward(ctx, g_h, g_xw, _gm, _gv, _gt):
        x, gamma, beta, mean, var, h, w, total = ctx.saved_tensors
        eps, relu, p, seed, group = ctx.cfg
        dx, dgamma, dbeta, gw = _tail_backward(ctx, g_h, g_xw, x, gamma, beta, mean, var, h, w, eps, relu, p, seed, 1, (total, group))
        return dx, dgamma, dbeta, None, None, None, None, gw, None, None


def sync_bn_act_linear(x: Tensor, bn, w: Tensor, relu: bool, p: float, training: bool, group=None):
    """(h, h @ w, mean, var, total) -- ``bn_act_linear`` with all-rank statistics (dist.SyncBatchNorm1d.fused_act_linear); None when
    the fused kernels do not take the shape."""
    if not (training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and w.dim() == 2 and w.shape[0] == x.shape[1]
            and w.shape[1] <= 64 and x.shape[1] % 64 == 0 and _bn_shape_ok(_rowmajor(x))):
        return None
    drop = p if p > 0 else 0.0
    seed = _draw_dropout_seed() if drop > 0 else 0
    box = _TapBox()
    h, xw, mean, var, total = _SyncBnActLinear.apply(x, bn.weight, bn.bias, bn.eps, relu, drop, seed, w, box, group)
    h._egnn_tap = box
    return h, xw, mean, var, total


def bn_shape_ok(x: Tensor) -> bool:
    return _bn_shape_ok(_rowmajor(x))
