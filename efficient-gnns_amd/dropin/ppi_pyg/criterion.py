"""Drop-in for /root/reference/ppi_pyg/criterion.py (``from criterion import *``, ppi_pyg/gnn.py): the PPI scripts are
MULTI-LABEL -- every criterion's classification term is BCE-with-logits (criterion.py:11,24,42,57,98,132) and ``kd_criterion`` is
BCE based as well (criterion.py:8-18, defaults alpha=0.5, T=1), not the class-index CE / KL of arxiv_pyg/criterion.py.
``dropin/launch.py`` puts this directory (and not ``dropin/criterion.py``) in front of ppi_pyg scripts.  Same names, signatures
and defaults as the reference file; the feature losses run on the same kernels as the arxiv criteria."""
from efficient_gnns_amd.criterion import ppi_at_criterion as at_criterion  # noqa: F401
from efficient_gnns_amd.criterion import ppi_fitnet_criterion as fitnet_criterion  # noqa: F401
from efficient_gnns_amd.criterion import ppi_gpw_criterion as gpw_criterion  # noqa: F401
from efficient_gnns_amd.criterion import ppi_kd_criterion as kd_criterion  # noqa: F401
from efficient_gnns_amd.criterion import ppi_lpw_criterion as lpw_criterion  # noqa: F401
from efficient_gnns_amd.criterion import ppi_nce_criterion as nce_criterion  # noqa: F401

__all__ = ["kd_criterion", "fitnet_criterion", "at_criterion", "gpw_criterion", "lpw_criterion", "nce_criterion"]
