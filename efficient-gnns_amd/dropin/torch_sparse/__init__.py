"""Drop-in shim for ``from torch_sparse import SparseTensor`` (/root/reference/mag_pyg/gnn.py:13)."""
from efficient_gnns_amd.sparse import SparseTensor  # noqa: F401
