"""LSP: per-edge similarity + segment softmax loss (kernels in csrc/edge_softmax.hip, csrc/spmm.hip)."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib, ops
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


_PLANS: dict = {}


def edge_plan(edge_index: Tensor, n: int) -> EdgePlan:
    key = (edge_index.data_ptr(), edge_index.shape[1], n, edge_index._version)
    plan = _PLANS.get(key)
    if plan is None:
        if len(_PLANS) > 8:
            _PLANS.clear()
        plan = _PLANS[key] = EdgePlan(edge_index, n)
    return plan


class _EdgeSim(torch.autograd.Function):
    """sim[e] = k(F[a_e], F[b_e]) for the edges of ``plan`` in dst-major order."""

    @staticmethod
    def forward(ctx, F, plan: EdgePlan, kernel: str):
        _lib.require_gpu(F)
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
        # dF[b] += sum_e alpha_e F[a_e]   and   dF[a] += sum_e alpha_e F[b_e]  : two aggregations with edge values
        gF, _ = ops.spmm_raw(plan.by_b.set_value(alpha), F, "sum")
        alpha_a = alpha[plan.perm_a].contiguous()
        gF2, _ = ops.spmm_raw(plan.by_a.set_value(alpha_a), F, "sum")
        # diagonal part: (sum_{e: b_e = i} beta_b + sum_{e: a_e = i} beta_a) * F[i]
        sb = torch.empty(plan.n, dtype=torch.float32, device=dev)
        sa = torch.empty(plan.n, dtype=torch.float32, device=dev)
        _lib.check(lib.egnn_segment_sum_f32(_lib.ptr(plan.ptr_b), _lib.ptr(beta_b), plan.n, _lib.ptr(sb), st), "egnn_segment_sum_f32")
        beta_a_a = beta_a[plan.perm_a].contiguous()
        _lib.check(lib.egnn_segment_sum_f32(_lib.ptr(plan.ptr_a), _lib.ptr(beta_a_a), plan.n, _lib.ptr(sa), st), "egnn_segment_sum_f32")
        gF = gF + gF2 + (sa + sb).unsqueeze(1) * F
        return gF, None, None


def lsp_loss(feat: Tensor, teacher_feat: Tensor, edge_index: Tensor, kernel: str, criterion: str = "kld") -> Tensor:
    """criterion.py:100-122: softmax over the edges sharing ``dst`` of the per-edge similarity, KL or MSE vs the teacher."""
    n = feat.shape[0]
    plan = edge_plan(edge_index, n)
    p_s = _SegSoftmax.apply(_EdgeSim.apply(feat, plan, kernel), plan.ptr_b)
    p_t = _SegSoftmax.apply(_EdgeSim.apply(teacher_feat, plan, kernel), plan.ptr_b)
    if criterion == "mse":
        d = p_s - p_t
        return (d * d).mean()
    # F.kl_div(log p_s, p_t, reduction='mean'): mean over edges of p_t (log p_t - log p_s), 0 where p_t == 0
    return torch.nn.functional.kl_div(torch.log(p_s), p_t, log_target=False, reduction="mean")
