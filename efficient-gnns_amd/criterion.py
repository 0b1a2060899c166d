"""The reference's distillation criteria on the gfx950 kernels -- same names, signatures, defaults and
3-tuple returns ``(loss, loss_cls, loss_aux)`` as /root/reference/arxiv_pyg/criterion.py:8,24,39,57,95,129,
so ``from criterion import *`` in the reference's train loops resolves unchanged.

Paper names: LSP = ``lpw_criterion``, GSP = ``gpw_criterion``, G-CRD = ``nce_criterion``.
``loss_aux`` carries its own autograd graph (gnn_kd_and_aux.py:121-127 recombines it).
Host-RNG coupling is preserved: one ``np.random.choice(n, S, replace=False)`` per gpw/nce call when S < n.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

__all__ = ["kd_criterion", "fitnet_criterion", "at_criterion", "gpw_criterion", "lpw_criterion", "nce_criterion",
           "loss_kd_only", "ppi_kd_criterion", "ppi_fitnet_criterion", "ppi_at_criterion", "ppi_gpw_criterion", "ppi_lpw_criterion",
           "ppi_nce_criterion"]


# Optional override of the sampled-row source: ``callable(n, max_samples, device) -> int64 device tensor | None``.
# models.GraphedEpoch installs one that returns a STATIC device buffer (refilled from the same NumPy draw before every
# replay), because a captured hipGraph cannot contain the host-side draw + upload.
_ROW_SAMPLER = None


def _sample_rows(n: int, max_samples: int, device):
    """criterion.py:62-65,134-137: host NumPy global RNG, exactly one draw; None = keep every row.
    (Making the draw earlier in the step -- before the forward, or at the end of the previous step -- was measured:
    no gain, the step is GPU-bound around it.)"""
    if _ROW_SAMPLER is not None:
        return _ROW_SAMPLER(n, max_samples, device)
    if max_samples < n:
        pick = np.random.choice(n, max_samples, replace=False)
        return torch.from_numpy(pick).to(device=device, dtype=torch.int64, non_blocking=True)
    return None


def _rows_of(x, idx):
    """(tensor, index) to hand to a kernel that gathers ``x[idx]`` itself.  A deferred activation (lazy.py: the reference's own model
    code under dropin/accel.py) forms only the rows ``idx`` -- what ``models.distill_loss`` does through ``forward_rows(pick=)``."""
    m = getattr(x, "_egnn_materialise", None)
    return (x, idx) if m is None else (m(pick=idx), None)


def _plain(x):
    m = getattr(x, "_egnn_materialise", None)
    return x if m is None else m()


def _plain_args(fn):
    """The classification / KD arguments (logits, labels, teacher_logits) of a criterion as real tensors (a deferred constant gather,
    dropin/accel.py, may reach them); the feature arguments are consumed lazily where that pays (``_rows_of``)."""
    import functools

    @functools.wraps(fn)
    def wrapped(logits, labels, *args, **kw):
        return fn(_plain(logits), _plain(labels), *args, **kw)
    return wrapped


def _ce_term(logits, labels, rows=None):
    return ops.cross_entropy(logits, labels, rows)


def _bce_term(logits, labels, rows=None):
    """Multi-label classification term of the PPI scripts (ppi_pyg/criterion.py:11,24,42,57,98,132): mean BCE-with-logits."""
    from .ops_pairwise import bce_with_logits_pair
    if rows is not None:
        logits, labels = logits[rows], labels[rows]
    return bce_with_logits_pair(logits, labels, logits.detach())[0]


# The public criteria keep the reference's signatures exactly.  Each has a ``rows_*`` twin (an extension; the reference has no
# such thing) that takes the FULL [N, .] ``logits`` / ``labels`` (/ ``teacher_logits``) plus the row ids (and, keyword-only, the
# classification term ``_cls``: class-index cross entropy for the arxiv / MAG scripts, multi-label BCE for the PPI ones): the classification / KD
# terms are evaluated on those rows inside the kernels -- the ``[train_idx]`` gathers of gnn.py:109-110,121 and the zero-fill +
# scatter of their backward never exist.  ``models.train_step`` uses the twins.


def kd_criterion(logits, labels, teacher_logits, alpha=0.9, T=4):
    """Logit KD (criterion.py:8-21): CE + KL(softmax(teacher/T) || softmax(logits/T)) with reduction='mean'."""
    return rows_kd_criterion(logits, labels, teacher_logits, alpha, T, rows=None)


def rows_kd_criterion(logits, labels, teacher_logits, alpha=0.9, T=4, rows=None):
    """``kd_criterion`` on the rows ``rows`` of full-size logits / labels (None: all rows, the reference's call)."""
    loss_cls, loss_kd = ops.ce_and_kd(_plain(logits), _plain(labels), _plain(teacher_logits), T, rows)
    loss = loss_kd * (alpha * T * T) + loss_cls * (1 - alpha)
    return loss, loss_cls, loss_kd


def loss_kd_only(logits, labels, teacher_logits, alpha=0.9, T=4):
    """north_star alias: the KD term alone (third return of ``kd_criterion``)."""
    return kd_criterion(logits, labels, teacher_logits, alpha, T)[2]


def fitnet_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """FitNet (criterion.py:24-36): MSE between L2-normalised rows."""
    return rows_fitnet_criterion(logits, labels, feat, teacher_feat, beta, rows=None)


@_plain_args
def rows_fitnet_criterion(logits, labels, feat, teacher_feat, beta=1000, rows=None, *, _cls=None):
    """``fitnet_criterion`` on the rows ``rows`` of full-size logits / labels (None: all rows, the reference's call)."""
    loss_cls = (_cls or _ce_term)(logits, labels, rows)
    loss_aux = ops.fitnet_loss(_plain(feat), _plain(teacher_feat))
    return torch.add(loss_cls, loss_aux, alpha=beta), loss_cls, loss_aux


def at_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """Attention transfer (criterion.py:39-54): per-node energies, L2-normalised ACROSS nodes."""
    return rows_at_criterion(logits, labels, feat, teacher_feat, beta, rows=None)


@_plain_args
def rows_at_criterion(logits, labels, feat, teacher_feat, beta=1000, rows=None, *, _cls=None):
    """``at_criterion`` on the rows ``rows`` of full-size logits / labels (None: all rows, the reference's call)."""
    loss_cls = (_cls or _ce_term)(logits, labels, rows)
    loss_aux = ops.at_loss(_plain(feat), _plain(teacher_feat))
    return torch.add(loss_cls, loss_aux, alpha=beta), loss_cls, loss_aux


def gpw_criterion(logits, labels, feat, teacher_feat, kernel="cosine", beta=1, max_samples=8192):
    """GSP (criterion.py:57-92): MSE between all-pairs similarity matrices of student and teacher rows."""
    return rows_gpw_criterion(logits, labels, feat, teacher_feat, kernel, beta, max_samples, rows=None)


@_plain_args
def rows_gpw_criterion(logits, labels, feat, teacher_feat, kernel="cosine", beta=1, max_samples=8192, rows=None, presampled=False, *, _cls=None):
    """``gpw_criterion`` on the rows ``rows`` of full-size logits / labels (None: all rows, the reference's call).
    ``presampled``: ``feat`` / ``teacher_feat`` already are the sampled rows (the caller made the draw with ``_sample_rows``)."""
    from .ops_pairwise import gsp_loss
    if kernel not in ("cosine", "poly", "l2", "rbf"):
        raise NotImplementedError
    loss_cls = (_cls or _ce_term)(logits, labels, rows)
    idx = None if presampled else _sample_rows(feat.shape[0], max_samples, feat.device)
    if hasattr(feat, "_egnn_materialise") or hasattr(teacher_feat, "_egnn_materialise"):   # deferred activations: only the rows idx
        from .lazy import materialise
        feat, teacher_feat, idx = materialise(feat, idx), materialise(teacher_feat, idx), None
    loss_aux = gsp_loss(feat, teacher_feat, idx, kernel)
    return torch.add(loss_cls, loss_aux, alpha=beta), loss_cls, loss_aux


def lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel="cosine", beta=100, criterion="kld"):
    """LSP (criterion.py:95-126): per-edge similarity, softmax over the edges sharing ``dst``, KL or MSE."""
    return rows_lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel, beta, criterion, rows=None)


@_plain_args
def rows_lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel="cosine", beta=100, criterion="kld", rows=None, *, _cls=None):
    """``lpw_criterion`` on the rows ``rows`` of full-size logits / labels (None: all rows, the reference's call)."""
    from .ops_edge import lsp_loss
    if kernel not in ("cosine", "poly", "l2", "rbf") or criterion not in ("kld", "mse"):
        raise NotImplementedError
    loss_cls = (_cls or _ce_term)(logits, labels, rows)
    loss_aux = lsp_loss(_plain(feat), _plain(teacher_feat), edge_index, kernel, criterion)
    return torch.add(loss_cls, loss_aux, alpha=beta), loss_cls, loss_aux


def nce_criterion(logits, labels, feat, teacher_feat, beta=0.5, nce_T=0.075, max_samples=8192):
    """G-CRD (criterion.py:129-149): InfoNCE between unit student rows and unit teacher rows."""
    return rows_nce_criterion(logits, labels, feat, teacher_feat, beta, nce_T, max_samples, rows=None)


@_plain_args
def rows_nce_criterion(logits, labels, feat, teacher_feat, beta=0.5, nce_T=0.075, max_samples=8192, rows=None, presampled=False, *, _cls=None):
    """``nce_criterion`` on the rows ``rows`` of full-size logits / labels (None: all rows, the reference's call).
    ``presampled``: as in ``rows_gpw_criterion``."""
    loss_cls = (_cls or _ce_term)(logits, labels, rows)
    idx = None if presampled else _sample_rows(feat.shape[0], max_samples, feat.device)
    fhat = ops.gather_normalize(*_rows_of(feat, idx))
    that = ops.gather_normalize(*_rows_of(teacher_feat, idx))
    loss_aux = ops.nce_unit(fhat, that, nce_T)
    return torch.add(loss_cls, loss_aux, alpha=beta), loss_cls, loss_aux


# The auxiliary criteria of /root/reference/ppi_pyg/criterion.py:21-146: the same feature losses as the arxiv scripts with the
# multi-label BCE-with-logits classification term (the file differs from arxiv_pyg/criterion.py in that line only).
def ppi_fitnet_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """ppi_pyg/criterion.py:21-36."""
    return rows_fitnet_criterion(logits, labels, feat, teacher_feat, beta, _cls=_bce_term)


def ppi_at_criterion(logits, labels, feat, teacher_feat, beta=1000):
    """ppi_pyg/criterion.py:39-54."""
    return rows_at_criterion(logits, labels, feat, teacher_feat, beta, _cls=_bce_term)


def ppi_gpw_criterion(logits, labels, feat, teacher_feat, kernel="cosine", beta=1, max_samples=8192):
    """ppi_pyg/criterion.py:57-92."""
    return rows_gpw_criterion(logits, labels, feat, teacher_feat, kernel, beta, max_samples, _cls=_bce_term)


def ppi_lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel="cosine", beta=100, criterion="kld"):
    """ppi_pyg/criterion.py:95-126."""
    return rows_lpw_criterion(logits, labels, feat, teacher_feat, edge_index, kernel, beta, criterion, _cls=_bce_term)


def ppi_nce_criterion(logits, labels, feat, teacher_feat, beta=0.5, nce_T=0.075, max_samples=8192):
    """ppi_pyg/criterion.py:129-149."""
    return rows_nce_criterion(logits, labels, feat, teacher_feat, beta, nce_T, max_samples, _cls=_bce_term)


def ppi_kd_criterion(logits, labels, teacher_logits, alpha=0.5, T=1):
    """Multi-label KD of /root/reference/ppi_pyg/criterion.py:8-18 (BCE-with-logits twice)."""
    from .ops_pairwise import bce_with_logits_pair
    loss_cls, loss_kd = bce_with_logits_pair(_plain(logits), _plain(labels), _plain(teacher_logits))
    return loss_kd * (alpha * T * T) + loss_cls * (1 - alpha), loss_cls, loss_kd
