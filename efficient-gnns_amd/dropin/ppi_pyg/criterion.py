"""Drop-in for /root/reference/ppi_pyg/criterion.py (``from criterion import *``, ppi_pyg/gnn.py): the PPI scripts
are MULTI-LABEL, their ``kd_criterion`` is BCE-with-logits based (criterion.py:8-18, defaults alpha=0.5, T=1) -- not the
class-index CE/KL of arxiv_pyg/criterion.py.  ``dropin/launch.py`` puts this directory (and not ``dropin/criterion.py``)
in front of ppi_pyg scripts.  Only the loss BASELINE.json's PPI configuration uses is native; the auxiliary PPI losses
are outside the hot path and raise instead of silently computing arxiv semantics."""
from efficient_gnns_amd.criterion import ppi_kd_criterion as kd_criterion  # noqa: F401

__all__ = ["kd_criterion"]


def __getattr__(name):
    if name in ("fitnet_criterion", "at_criterion", "gpw_criterion", "lpw_criterion", "nce_criterion"):
        raise NotImplementedError(
            f"ppi_pyg {name}: the multi-label (BCE) variants of the auxiliary losses are not part of the MI355X hot path; "
            "only ppi_pyg's kd_criterion (logit KD) is provided")
    raise AttributeError(name)
