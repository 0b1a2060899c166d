"""LSP edge-wise similarity + segment softmax loss (kernels in csrc/edge_softmax.hip)."""
from __future__ import annotations


def segment_softmax(src, index, num_nodes=None):
    raise NotImplementedError("segment softmax kernel not built yet")


def lsp_loss(feat, teacher_feat, edge_index, kernel, criterion):
    raise NotImplementedError("LSP kernel not built yet")
