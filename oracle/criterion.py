"""ORACLE (test infrastructure): the distillation losses of the reference, restated.

Each function follows /root/reference/arxiv_pyg/criterion.py (line ranges in the docstrings) and
returns the reference's 3-tuple ``(loss, loss_cls, loss_aux)``.  Paper names: LSP = ``lpw``,
GSP = ``gpw``, G-CRD = ``nce``.  ``ppi_*_criterion`` follow /root/reference/ppi_pyg/criterion.py:8-146 (the same
feature losses with the multi-label BCE-with-logits classification term).

Pinned against the reference's own file by ``tests/golden/make_golden.py`` (see oracle/__init__.py).
Host-RNG coupling is preserved: exactly one ``np.random.choice(n, S, replace=False)`` per ``gpw`` /
``nce`` call when ``S < n`` (criterion.py:63,135).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .utils import softmax as _segment_softmax


def _ce(logits, labels):
    return F.cross_entropy(logits, labels)


def _unit(x):
    return F.normalize(x, p=2, dim=-1)


def _subsample(feat, teacher_feat, max_samples):
    """criterion.py:62-65 / 134-137 -- host NumPy global RNG, one draw."""
    n = feat.shape[0]
    if max_samples < n:
        pick = np.random.choice(n, max_samples, replace=False)
        feat, teacher_feat = feat[pick], teacher_feat[pick]
    return feat, teacher_feat


def kd_criterion(logits, labels, teacher_logits, alpha=0.9, T=4):
    """criterion.py:8-21.  KL uses F.kl_div's default reduction='mean' (divides by n*C)."""
    loss_cls = _ce(logits, labels)
    log_q = F.log_softmax(logits / T, dim=1)
    p = F.softmax(teacher_logits / T, dim=1)
    loss_kd = F.kl_div(log_q, p, log_target=False)
    return loss_kd * (alpha * T * T) + loss_cls * (1 - alpha), loss_cls, loss_kd


def loss_kd_only(logits, labels, teacher_logits, alpha=0.9, T=4):
    """north_star alias: third return of ``kd_criterion``."""
    return kd_criterion(logits, labels, teacher_logits, alpha, T)[2]


def fitnet_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """criterion.py:24-36."""
    loss_cls = _ce(logits, labels)
    loss_aux = F.mse_loss(_unit(feat), _unit(teacher_feat))
    return loss_cls + beta * loss_aux, loss_cls, loss_aux


def at_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """criterion.py:39-54.  Per-node energy vectors are L2-normalised ACROSS nodes."""
    loss_cls = _ce(logits, labels)
    e_s = (feat * feat).sum(-1)
    e_t = (teacher_feat * teacher_feat).sum(-1)
    loss_aux = F.mse_loss(_unit(e_s), _unit(e_t))
    return loss_cls + beta * loss_aux, loss_cls, loss_aux


def _pairwise(x, kernel):
    """All-pairs similarity, flattened [S*S] (criterion.py:69-86)."""
    if kernel in ("cosine", "poly"):
        u = _unit(x)
        g = (u @ u.t()).flatten()
        return g if kernel == "cosine" else g ** 2
    diff = x.unsqueeze(0) - x.unsqueeze(1)  # [S,S,D] materialised, as the reference does
    if kernel == "l2":
        return diff.norm(p=2, dim=-1).flatten()
    if kernel == "rbf":
        return torch.exp(-0.5 * (diff ** 2).sum(dim=-1).flatten())
    raise NotImplementedError


def gpw_criterion(logits, labels, feat, teacher_feat, kernel="cosine", beta=1, max_samples=8192):
    """GSP, criterion.py:57-92."""
    loss_cls = _ce(logits, labels)
    feat, teacher_feat = _subsample(feat, teacher_feat, max_samples)
    loss_aux = F.mse_loss(_pairwise(feat, kernel), _pairwise(teacher_feat, kernel))
    return loss_cls + beta * loss_aux, loss_cls, loss_aux


def _edge_sim(x, src, dst, kernel):
    """Per-edge similarity (criterion.py:102-115)."""
    a, b = x[src], x[dst]
    if kernel == "cosine":
        return F.cosine_similarity(a, b)
    if kernel == "poly":
        return F.cosine_similarity(a, b) ** 2
    if kernel == "l2":
        return (a - b).norm(p=2, dim=-1)
    if kernel == "rbf":
        return torch.exp(-0.5 * ((a - b) ** 2).sum(dim=-1))
    raise NotImplementedError


def lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel="cosine", beta=100, criterion="kld"):
    """LSP, criterion.py:95-126.  Softmax is over edges grouped by ``dst``."""
    loss_cls = _ce(logits, labels)
    src, dst = edge_index
    p_s = _segment_softmax(_edge_sim(feat, src, dst, kernel), dst)
    p_t = _segment_softmax(_edge_sim(teacher_feat, src, dst, kernel), dst)
    if criterion == "mse":
        loss_aux = F.mse_loss(p_s, p_t)
    elif criterion == "kld":
        loss_aux = F.kl_div(torch.log(p_s), p_t, log_target=False)
    else:
        raise NotImplementedError
    return loss_cls + beta * loss_aux, loss_cls, loss_aux


def nce_criterion(logits, labels, feat, teacher_feat, beta=0.5, nce_T=0.075, max_samples=8192):
    """G-CRD, criterion.py:129-149: InfoNCE with the same-node teacher row as the positive."""
    loss_cls = _ce(logits, labels)
    feat, teacher_feat = _subsample(feat, teacher_feat, max_samples)
    z = _unit(feat) @ _unit(teacher_feat).t()
    target = torch.arange(z.shape[0])
    loss_aux = F.cross_entropy(z / nce_T, target)
    return loss_cls + beta * loss_aux, loss_cls, loss_aux


def ppi_kd_criterion(logits, labels, teacher_logits, alpha=0.5, T=1):
    """Multi-label logit KD, /root/reference/ppi_pyg/criterion.py:8-18 (T only scales the mix)."""
    loss_cls = F.binary_cross_entropy_with_logits(logits, labels)
    loss_kd = F.binary_cross_entropy_with_logits(logits, torch.sigmoid(teacher_logits))
    return loss_kd * (alpha * T * T) + loss_cls * (1 - alpha), loss_cls, loss_kd


def _with_bce(fn):
    """The PPI flavour of an arxiv criterion: /root/reference/ppi_pyg/criterion.py differs from arxiv_pyg/criterion.py in the
    classification term only (F.binary_cross_entropy_with_logits, lines 24,42,57,98,132)."""
    def wrapped(*a, **kw):
        global _ce
        prev = _ce
        _ce = lambda logits, labels: F.binary_cross_entropy_with_logits(logits, labels)   # noqa: E731
        try:
            return fn(*a, **kw)
        finally:
            _ce = prev
    return wrapped


def ppi_fitnet_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """ppi_pyg/criterion.py:21-36."""
    return _with_bce(fitnet_criterion)(logits, labels, feat, teacher_feat, beta)


def ppi_at_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """ppi_pyg/criterion.py:39-54."""
    return _with_bce(at_criterion)(logits, labels, feat, teacher_feat, beta)


def ppi_gpw_criterion(logits, labels, feat, teacher_feat, kernel="cosine", beta=1, max_samples=8192):
    """ppi_pyg/criterion.py:57-92."""
    return _with_bce(gpw_criterion)(logits, labels, feat, teacher_feat, kernel, beta, max_samples)


def ppi_lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel="cosine", beta=100, criterion="kld"):
    """ppi_pyg/criterion.py:95-126."""
    return _with_bce(lpw_criterion)(logits, labels, feat, teacher_feat, edge_index, kernel, beta, criterion)


def ppi_nce_criterion(logits, labels, feat, teacher_feat, beta=0.5, nce_T=0.075, max_samples=8192):
    """ppi_pyg/criterion.py:129-149."""
    return _with_bce(nce_criterion)(logits, labels, feat, teacher_feat, beta, nce_T, max_samples)
