"""LSP: per-edge similarity + segment softmax loss (kernels in csrc/edge_softmax.hip, csrc/spmm.hip)."""
from __future__ import annotations

import os

import torch
from torch import Tensor

from . import _cache, _lib, ops
from .sparse import SparseTensor, _ind2ptr

_KERNELS = {"cosine": 0, "poly": 1, "l2": 2, "rbf": 3}


# ------------------------------------------------------------------------------------------------
# segment softmax (= torch_geometric.utils.softmax on sorted segments)
# ------------------------------------------------------------------------------------------------
class _SegSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr):
        _lib.require_gpu(x, seg_ptr)
        x = x.contiguous()
        p = torch.zeros_like(x)
        rc = _lib.load().egnn_segment_softmax_fwd_f32(_lib.ptr(seg_ptr), _lib.ptr(x), seg_ptr.numel() - 1, _lib.ptr(p), _lib.stream())
        _lib.check(rc, "egnn_segment_softmax_fwd_f32")
        ctx.save_for_backward(p, seg_ptr)
        return p

    @staticmethod
    def backward(ctx, gp):
        p, seg_ptr = ctx.saved_tensors
        gp = gp.contiguous()
        gx = torch.zeros_like(p)
        rc = _lib.load().egnn_segment_softmax_bwd_f32(_lib.ptr(seg_ptr), _lib.ptr(p), _lib.ptr(gp), seg_ptr.numel() - 1,
                                                      _lib.ptr(gx), _lib.stream())
        _lib.check(rc, "egnn_segment_softmax_bwd_f32")
        return gx, None


def segment_softmax(src: Tensor, index: Tensor, num_nodes: int | None = None) -> Tensor:
    """``utils.softmax(src, index)`` for an arbitrary (unsorted) index: sort once, softmax per segment, unsort."""
    _lib.require_gpu(src, index)
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    perm = torch.argsort(index, stable=True)
    seg_ptr = _ind2ptr(index[perm].contiguous(), n)
    p_sorted = _SegSoftmax.apply(src[perm], seg_ptr)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device)
    return p_sorted[inv]


# ------------------------------------------------------------------------------------------------
# edge plan: both CSR groupings of an edge list, cached per edge_index tensor
# ------------------------------------------------------------------------------------------------
class EdgePlan:
    """edge_index = (a, b).  Keeps the edges grouped by ``b`` (softmax segments, criterion.py:101 ``dst``) and by
    ``a``, as value-less SparseTensors whose row plans are reused by the backward SpMMs."""

    def __init__(self, edge_index: Tensor, n: int):
        a, b = edge_index[0].contiguous(), edge_index[1].contiguous()
        self.n, self.E = n, a.numel()
        self.perm_b = torch.argsort(b, stable=True)             # segment (dst-major) order
        self.a_in_b = a[self.perm_b].contiguous()
        self.b_in_b = b[self.perm_b].contiguous()
        self.ptr_b = _ind2ptr(self.b_in_b, n)
        self.by_b = SparseTensor(rowptr=self.ptr_b, col=self.a_in_b, sparse_sizes=(n, n))       # row b gathers a
        perm_a = torch.argsort(self.a_in_b, stable=True)        # position in b-order -> a-major order
        self.perm_a = perm_a
        self.ptr_a = _ind2ptr(self.a_in_b[perm_a].contiguous(), n)
        self.by_a = SparseTensor(rowptr=self.ptr_a, col=self.b_in_b[perm_a].contiguous(), sparse_sizes=(n, n))  # row a gathers b
        # "row b gathers a" PLUS one diagonal entry per node: the backward's  sum_e alpha_e F[a_e] + d_i F[i]  is then ONE aggregation
        # (values = the step's alpha followed by the step's diagonal, brought into CSR order by ``perm_bd``) instead of an
        # aggregation + a scale pass + an add pass over [n, D]
        ids = torch.arange(n, dtype=torch.int64, device=a.device)
        rows_bd = torch.cat([self.b_in_b, ids])
        self.perm_bd = torch.argsort(rows_bd, stable=True)
        self.by_bd = SparseTensor(rowptr=_ind2ptr(rows_bd[self.perm_bd].contiguous(), n), col=torch.cat([self.a_in_b, ids])[self.perm_bd].contiguous(),
                                  sparse_sizes=(n, n))


_PLANS = _cache.TensorKeyedCache(capacity=16)


def edge_plan(edge_index: Tensor, n: int) -> EdgePlan:
    """The plan of an edge list, built once per edge_index tensor (identity + version); the entry keeps the tensor alive (_cache.py)."""
    return _PLANS.get((edge_index,), (n,), lambda: EdgePlan(edge_index, n))


class _EdgeSim(torch.autograd.Function):
    """sim[e] = k(F[a_e], F[b_e]) for the edges of ``plan`` in dst-major order."""

    @staticmethod
    def forward(ctx, F, plan: EdgePlan, kernel: str):
        _lib.require_gpu(F)
        ctx.tap_box = getattr(F, "_egnn_tap", None)
        F = ops._rowmajor(F)
        E = plan.E
        sim = torch.empty(E, dtype=torch.float32, device=F.device)
        aux = torch.empty(E, 3, dtype=torch.float32, device=F.device)
        rc = _lib.load().egnn_edge_sim_f32(_lib.ptr(F), F.stride(0), F.shape[1], _lib.ptr(plan.a_in_b), _lib.ptr(plan.b_in_b), E,
                                           _KERNELS[kernel], _lib.ptr(sim), _lib.ptr(aux), _lib.stream())
        _lib.check(rc, "egnn_edge_sim_f32")
        ctx.save_for_backward(F, sim, aux)
        ctx.plan, ctx.kernel = plan, kernel
        return sim

    @staticmethod
    def backward(ctx, g):
        F, sim, aux = ctx.saved_tensors
        plan, E = ctx.plan, ctx.plan.E
        lib, st, dev = _lib.load(), _lib.stream(), F.device
        g = g.contiguous()
        alpha, beta_a, beta_b = (torch.empty(E, dtype=torch.float32, device=dev) for _ in range(3))
        _lib.check(lib.egnn_edge_sim_coef_f32(_lib.ptr(g), _lib.ptr(sim), _lib.ptr(aux), E, _KERNELS[ctx.kernel], _lib.ptr(alpha),
                                              _lib.ptr(beta_a), _lib.ptr(beta_b), st), "egnn_edge_sim_coef_f32")
        # dF[b] += sum_e alpha_e F[a_e],  dF[a] += sum_e alpha_e F[b_e],  dF[i] += (sum_{e: b_e = i} beta_b + sum_{e: a_e = i} beta_a) F[i]:
        # two aggregations with edge values -- the second carries the diagonal as extra entries and adds the first in its store
        alpha_a = alpha[plan.perm_a].contiguous()
        gF2, _ = ops.spmm_raw(plan.by_a.set_value(alpha_a), F, "sum")
        sb = torch.empty(plan.n, dtype=torch.float32, device=dev)
        sa = torch.empty(plan.n, dtype=torch.float32, device=dev)
        _lib.check(lib.egnn_segment_sum_f32(_lib.ptr(plan.ptr_b), _lib.ptr(beta_b), plan.n, _lib.ptr(sb), st), "egnn_segment_sum_f32")
        beta_a_a = beta_a[plan.perm_a].contiguous()
        _lib.check(lib.egnn_segment_sum_f32(_lib.ptr(plan.ptr_a), _lib.ptr(beta_a_a), plan.n, _lib.ptr(sa), st), "egnn_segment_sum_f32")
        vals = torch.cat([alpha, sa + sb])[plan.perm_bd]
        gF, _ = ops.spmm_raw(plan.by_bd.set_value(vals), F, "sum", addend=gF2)
        return ops._fresh(gF, ctx.tap_box), None, None


class _LspLoss(torch.autograd.Function):
    """mean_e criterion(softmax_seg(sim_s), softmax_seg(sim_t)) -- both segment softmaxes, the KL / MSE term and the mean over the
    edges in one kernel per direction (egnn_lsp_loss_{fwd,bwd}_f32).  No torch reduction is involved: the multi-block form of
    ``tensor.mean()`` zeroes its semaphores with a memset node, and inside replayed hipGraphs on this stack such a reduction was seen
    to leave its output unwritten (profiles/r04_lsp_trace.txt: loss_aux kept the stale bytes of an earlier workspace)."""

    @staticmethod
    def forward(ctx, sim_s, sim_t, seg_ptr, criterion):
        _lib.require_gpu(sim_s, sim_t, seg_ptr)
        sim_s, sim_t = sim_s.contiguous(), sim_t.contiguous()
        E, n_seg = sim_s.numel(), seg_ptr.numel() - 1
        dev = sim_s.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        if E == 0:     # F.kl_div / F.mse_loss of empty tensors with reduction='mean': nan
            ctx.empty = True
            ctx.shapes = (sim_s.shape, sim_t.shape)
            return loss.fill_(float("nan"))
        ctx.empty = False
        lib = _lib.load()
        p_s, p_t = torch.empty_like(sim_s), torch.empty_like(sim_t)
        ws = torch.empty(lib.egnn_lsp_loss_ws_floats(), dtype=torch.float32, device=dev)
        _lib.check(lib.egnn_lsp_loss_fwd_f32(_lib.ptr(seg_ptr), _lib.ptr(sim_s), _lib.ptr(sim_t), n_seg, E, criterion, _lib.ptr(p_s),
                                             _lib.ptr(p_t), _lib.ptr(loss), _lib.ptr(ws), _lib.stream()), "egnn_lsp_loss_fwd_f32")
        ctx.save_for_backward(p_s, p_t, seg_ptr)
        ctx.criterion = criterion
        if _DEBUG_CHECKS and not torch.cuda.is_current_stream_capturing():
            _check_lsp_invariants(p_s, p_t, seg_ptr, loss, sim_s, sim_t, criterion)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.empty:
            return torch.zeros(ctx.shapes[0], device=g.device), None, None, None
        p_s, p_t, seg_ptr = ctx.saved_tensors
        E, n_seg = p_s.numel(), seg_ptr.numel() - 1
        g = g.contiguous().to(torch.float32)
        gs = torch.empty_like(p_s)
        gt = torch.empty_like(p_t) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().egnn_lsp_loss_bwd_f32(_lib.ptr(seg_ptr), _lib.ptr(p_s), _lib.ptr(p_t), n_seg, E, ctx.criterion, _lib.ptr(g),
                                                     _lib.ptr(gs), _lib.ptr(gt), _lib.stream()), "egnn_lsp_loss_bwd_f32")
        return gs, gt, None, None


_DEBUG_CHECKS = os.environ.get("EGNN_DEBUG_CHECKS", "0") == "1"


def _check_lsp_invariants(p_s, p_t, seg_ptr, loss, sim_s, sim_t, criterion):
    """Debug (EGNN_DEBUG_CHECKS=1, eager launches only; host reads): every non-empty segment of both softmaxes sums to 1, and the
    KL mean respects the bound that holds for ANY finite features when the similarities lie in [lo, hi]:
    -log p_s <= (hi - lo) + ln(deg), sum_seg p_t = 1  =>  0 <= loss <= (n_seg / E) * ((hi - lo) + ln(max_deg))."""
    import math
    n_seg = seg_ptr.numel() - 1
    deg = seg_ptr[1:] - seg_ptr[:-1]
    has = deg > 0
    seg = torch.repeat_interleave(torch.arange(n_seg, device=p_s.device), deg)
    for name, p in (("p_s", p_s), ("p_t", p_t)):
        tot = torch.zeros(n_seg, dtype=torch.float64, device=p.device).index_add_(0, seg, p.double())
        worst = float((tot[has] - 1).abs().max())
        if not worst <= 1e-4:
            raise AssertionError(f"lsp_loss: a segment of {name} sums to 1 +- {worst:.3e}")
    if criterion == 0:
        span = float(torch.maximum(sim_s.max() - sim_s.min(), sim_t.max() - sim_t.min()))
        bound = float(has.sum()) / p_s.numel() * (span + math.log(max(int(deg.max()), 1))) + 1e-6
        val = float(loss)
        if not (-1e-6 <= val <= bound):
            raise AssertionError(f"lsp_loss: KL mean {val:.6e} outside [0, {bound:.4f}]")


def lsp_loss(feat: Tensor, teacher_feat: Tensor, edge_index: Tensor, kernel: str, criterion: str = "kld") -> Tensor:
    """criterion.py:100-122: softmax over the edges sharing ``dst`` of the per-edge similarity, KL or MSE vs the teacher."""
    n = feat.shape[0]
    plan = edge_plan(edge_index, n)
    sim_s = _EdgeSim.apply(feat, plan, kernel)
    sim_t = _EdgeSim.apply(teacher_feat, plan, kernel)
    return _LspLoss.apply(sim_s, sim_t, plan.ptr_b, 1 if criterion == "mse" else 0)
