"""Drop-in for the reference's ``criterion.py`` (``from criterion import *``, /root/reference/arxiv_pyg/gnn.py:20):
the six distillation losses on the gfx950 kernels, same names / signatures / 3-tuple returns."""
from efficient_gnns_amd.criterion import *  # noqa: F401,F403
from efficient_gnns_amd.criterion import __all__  # noqa: F401
