"""GSP all-pairs similarity loss and the PPI BCE pair (kernels in csrc/pairwise.hip)."""
from __future__ import annotations


def gsp_loss(feat, teacher_feat, idx, kernel):
    raise NotImplementedError("GSP kernel not built yet")


def bce_with_logits_pair(logits, labels, teacher_logits):
    raise NotImplementedError("BCE kernel not built yet")
