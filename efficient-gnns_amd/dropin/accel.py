"""Opt-out acceleration of two torch modules the reference's own model code calls next to the shimmed operators.

``arxiv_pyg/gnn.py`` builds its models from ``torch_geometric`` convs (shimmed by this directory) AND from ``torch.nn.BatchNorm1d``
(gnn.py:34,68,300-305) and ``torch.nn.Linear`` (the projection heads, gnn.py:296-306).  Those two are torch's own classes: the
name shims cannot reach them, and on the arxiv-shaped problem ATen's BatchNorm kernels for [N, 256] inputs alone take 3.6 ms of
a 12.5 ms epoch (profiles/r04_reference_loop.txt).  ``enable()`` re-points their ``forward`` -- for 2-D float32 GPU inputs only,
everything else takes the original path -- at this package's kernels:

  * ``BatchNorm1d.forward``  -> ``ops.bn_act(x, self, relu=False)``: statistics + apply in two passes (the statistics come out
    of the preceding ``GCNConv``'s aggregation epilogue when that conv produced ``x``), same running-statistics update, backward in
    two passes with the bias gradient of the layer in front formed on the way;
  * ``Linear.forward``       -> ``ops.linear`` (the split-bf16 / f32 MFMA GEMMs of the package).

  * ``torch.optim.Adam(...)`` without an explicit ``fused`` / ``foreach`` choice over float GPU parameters -> ``fused=True``: the same
    update (gnn.py:308-312 passes parameter groups and ``lr`` only) in one launch per group instead of PyTorch's default
    multi-tensor chain (~0.3 ms per step on the headline problem).

Same values within the fp32 tolerance of the package's parity tests; ``disable()`` restores torch's methods.  ``launch.py`` enables
it unless ``--plain-torch-modules`` is given.  Scoped forms for code that must not change torch process-wide: ``with accel.scope():``
(enabled inside the block only) and ``with accel.double_backward():`` (torch's own kernels for a ``create_graph=True`` region: the
package's autograd Functions implement first derivatives only); an ``Adam(differentiable=True)`` is left to torch.
"""
from __future__ import annotations

import torch

_orig: dict = {}


def enabled() -> bool:
    return bool(_orig)


def enable() -> None:
    if _orig:
        return
    from efficient_gnns_amd import ops
    bn_forward, lin_forward = torch.nn.BatchNorm1d.forward, torch.nn.Linear.forward
    _orig.update(bn=bn_forward, lin=lin_forward)

    def fast_bn(self, x):
        if (not _double_backward_wanted(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight is not None and self.bias is not None
                and self.track_running_stats and self.running_mean is not None and x.shape[0] > 1 and ops._bn_shape_ok(ops._rowmajor(x))):
            return ops.bn_act(x, self, relu=False, p=0.0, training=self.training)
        return bn_forward(self, x)

    def fast_linear(self, x):
        if not _double_backward_wanted(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and x.shape[0] > 0:
            return ops.linear(x, self.weight, self.bias)
        return lin_forward(self, x)
    torch.nn.BatchNorm1d.forward = fast_bn
    torch.nn.Linear.forward = fast_linear
    adam_init = torch.optim.Adam.__init__
    _orig["adam"] = adam_init

    def fused_adam_init(self, params, *args, **kwargs):
        # materialise first: the script hands over generators (``model.parameters()``) inside its group dicts
        params = [dict(g, params=list(g["params"])) if isinstance(g, dict) else g for g in list(params)]
        # (fused / foreach are keyword-only in practice; ``differentiable=True`` excludes the fused implementation: left to torch)
        if "fused" not in kwargs and "foreach" not in kwargs and len(args) < 6 and not kwargs.get("differentiable", False):
            flat = [p for g in params for p in (g["params"] if isinstance(g, dict) else [g])]
            if flat and all(isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.float32 for p in flat):
                kwargs["fused"] = True
        adam_init(self, params, *args, **kwargs)
    torch.optim.Adam.__init__ = fused_adam_init


_DOUBLE_BACKWARD = False


class double_backward:
    """``with accel.double_backward():`` -- inside the block BatchNorm1d / Linear take torch's own kernels again: the package's autograd
    Functions implement first derivatives only (a ``create_graph=True`` backward through them raises)."""

    def __enter__(self):
        global _DOUBLE_BACKWARD
        self.prev, _DOUBLE_BACKWARD = _DOUBLE_BACKWARD, True

    def __exit__(self, *exc):
        global _DOUBLE_BACKWARD
        _DOUBLE_BACKWARD = self.prev


def _double_backward_wanted(x) -> bool:
    return _DOUBLE_BACKWARD


class scope:
    """``with accel.scope():`` -- the re-pointed methods only inside the block (restored on exit, also on an exception)."""

    def __enter__(self):
        self.was = enabled()
        enable()
        return self

    def __exit__(self, *exc):
        if not self.was:
            disable()


def disable() -> None:
    if not _orig:
        return
    torch.nn.BatchNorm1d.forward = _orig.pop("bn")
    torch.nn.Linear.forward = _orig.pop("lin")
    torch.optim.Adam.__init__ = _orig.pop("adam")
