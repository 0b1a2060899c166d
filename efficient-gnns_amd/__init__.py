"""efficient-gnns_amd: MI355X-native (gfx950 / CDNA4) GNN-distillation hot path of chaitjo/efficient-gnns.

Host-side mirror of the reference's operator interface over the C ABI in
``include/egnn_hip.h`` (``lib/libegnn_hip.so``).  Import name: ``efficient_gnns_amd``.
"""
from efficient_gnns_amd import _lib  # noqa: F401
from efficient_gnns_amd.sparse import SparseTensor, gcn_norm  # noqa: F401
from efficient_gnns_amd.nn import GATConv, GCNConv, RGCNConv, SAGEConv  # noqa: F401
from efficient_gnns_amd.transforms import ToSparseTensor, to_sparse_tensor  # noqa: F401
from efficient_gnns_amd.utils import softmax, subgraph  # noqa: F401
from efficient_gnns_amd.criterion import (  # noqa: F401
    kd_criterion, fitnet_criterion, at_criterion, gpw_criterion, lpw_criterion, nce_criterion, loss_kd_only,
    ppi_kd_criterion)

__version__ = "0.1.0"
