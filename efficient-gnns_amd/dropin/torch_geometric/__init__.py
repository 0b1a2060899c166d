"""Drop-in shim: resolves the ``torch_geometric`` names the reference's hot path imports
(/root/reference/arxiv_pyg/gnn.py:11-14, criterion.py:5) to the MI355X-native package.
Put ``efficient-gnns_amd/dropin`` (and the repo root) on PYTHONPATH -- see INTEGRATION.md."""
from . import nn, transforms, utils  # noqa: F401

__version__ = "1.7.0+egnn_amd"
