"""GSP all-pairs similarity loss and the PPI BCE pair (kernels in csrc/pairwise.hip)."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib, ops

_KERNELS = {"cosine": 0, "poly": 1, "l2": 2, "rbf": 3}


class _GSP(torch.autograd.Function):
    """loss = mean_ij (k(xs_i,xs_j) - k(xt_i,xt_j))^2 on S rows (unit rows for cosine/poly, raw rows for l2/rbf)."""

    @staticmethod
    def forward(ctx, xs, xt, kernel):
        _lib.require_gpu(xs, xt)
        xs, xt = ops._rowmajor(xs), ops._rowmajor(xt)
        S = xs.shape[0]
        if xt.shape[0] != S:
            raise ValueError("gsp: student and teacher need the same number of rows")
        lib, dev = _lib.load(), xs.device
        Ws = torch.empty(S, S, dtype=torch.float32, device=dev)
        Wt = torch.empty(S, S, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        nws = lib.egnn_gsp_ws_floats(S)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        rc = lib.egnn_gsp_fwd_f32(_lib.ptr(xs), xs.stride(0), xs.shape[1], _lib.ptr(xt), xt.stride(0), xt.shape[1], S,
                                  _KERNELS[kernel], _lib.ptr(Ws), _lib.ptr(Wt), _lib.ptr(loss), _lib.ptr(ws), nws, _lib.stream())
        _lib.check(rc, "egnn_gsp_fwd_f32")
        ctx.save_for_backward(xs, xt, Ws, Wt)
        ctx.dist = kernel in ("l2", "rbf")
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        xs, xt, Ws, Wt = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        lib, st = _lib.load(), _lib.stream()
        outs = []
        for need, x, W in ((ctx.needs_input_grad[0], xs, Ws), (ctx.needs_input_grad[1], xt, Wt)):
            if not need:
                outs.append(None)
                continue
            S, P = x.shape
            wx = ops.gemm_raw(W, x)  # [S,S] @ [S,P] on the fp32 MFMA
            r = None
            if ctx.dist:
                r = torch.empty(S, dtype=torch.float32, device=x.device)
                _lib.check(lib.egnn_rowsum_f32(_lib.ptr(W), W.stride(0), S, S, _lib.ptr(r), st), "egnn_rowsum_f32")
            dx = torch.empty_like(x)
            _lib.check(lib.egnn_scale_rowcorr_f32(_lib.ptr(wx), wx.stride(0), _lib.ptr(x), x.stride(0), _lib.ptr(r), _lib.ptr(g),
                                                  S, P, _lib.ptr(dx), dx.stride(0), st), "egnn_scale_rowcorr_f32")
            outs.append(dx)
        return outs[0], outs[1], None


def gsp_loss(feat: Tensor, teacher_feat: Tensor, idx: Tensor | None, kernel: str) -> Tensor:
    if kernel in ("cosine", "poly"):
        xs, xt = ops.gather_normalize(feat, idx), ops.gather_normalize(teacher_feat, idx)
    else:
        xs, xt = (feat, teacher_feat) if idx is None else (feat[idx], teacher_feat[idx])
    return _GSP.apply(xs, xt, kernel)


class _BcePair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, teacher):
        _lib.require_gpu(logits, labels, teacher)
        logits, labels, teacher = logits.contiguous(), labels.contiguous().to(torch.float32), teacher.contiguous()
        lib, dev = _lib.load(), logits.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.egnn_bce_pair_ws_floats(), dtype=torch.float32, device=dev)
        rc = lib.egnn_bce_pair_fwd_f32(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(teacher), logits.numel(), _lib.ptr(out),
                                       _lib.ptr(ws), _lib.stream())
        _lib.check(rc, "egnn_bce_pair_fwd_f32")
        ctx.save_for_backward(logits, labels, teacher)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_kd):
        logits, labels, teacher = ctx.saved_tensors
        dl = torch.empty_like(logits)
        g_cls = None if g_cls is None else g_cls.contiguous().to(torch.float32)
        g_kd = None if g_kd is None else g_kd.contiguous().to(torch.float32)
        rc = _lib.load().egnn_bce_pair_bwd_f32(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(teacher), logits.numel(),
                                               _lib.ptr(g_cls), _lib.ptr(g_kd), _lib.ptr(dl), _lib.stream())
        _lib.check(rc, "egnn_bce_pair_bwd_f32")
        return dl, None, None


def bce_with_logits_pair(logits: Tensor, labels: Tensor, teacher_logits: Tensor):
    """(mean BCEWithLogits(logits, labels), mean BCEWithLogits(logits, sigmoid(teacher_logits)))."""
    return _BcePair.apply(logits, labels, teacher_logits)
