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

Round 6 (all scoped to ``enable()`` .. ``disable()``, each with its switch): ``BatchNorm1d.forward`` returns a deferred activation that
absorbs the script's ``F.relu`` / ``F.dropout`` (``LAZY``, efficient_gnns_amd/lazy.py); inference convs seen feeding a BatchNorm are folded
with it (``DEFER_CONV``); equal-option Adam groups are stepped with one fused launch (``MERGE_ADAM_GROUPS``); and ``torch.Tensor.__getitem__``
is guarded so that ONE kind of indexing -- a row gather of a large float32 GPU leaf outside autograd by a 1-D int64 GPU index inside a
grad-enabled region, i.e. ``teacher_out_feat[train_idx]`` of gnn.py:155 -- is deferred into the consuming ``Linear``'s planes GEMMs
(``DEFER_CONST_GATHER``); every other indexing expression takes torch's path unchanged.

Same values within the fp32 tolerance of the package's parity tests; ``disable()`` restores torch's methods.  ``launch.py`` enables
it unless ``--plain-torch-modules`` is given.  Scoped forms for code that must not change torch process-wide: ``with accel.scope():``
(enabled inside the block only) and ``with accel.double_backward():`` (torch's own kernels for a ``create_graph=True`` region: the
package's autograd Functions implement first derivatives only); an ``Adam(differentiable=True)`` is left to torch.
"""
from __future__ import annotations

import os
import weakref

import torch

_orig: dict = {}
_NEXT_BN: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()    # GCNConv -> weakref(BatchNorm1d it was seen to feed); kept outside the modules (pickling)
DEFER_CONV = os.environ.get("EGNN_ACCEL_DEFER_CONV", "1") != "0"   # inference: a GCNConv seen to feed a BatchNorm1d is deferred, so that conv + BN + ReLU run as one folded pass (lazy.LazyFold)
DEFER_CONST_GATHER = os.environ.get("EGNN_ACCEL_DEFER_GATHER", "1") != "0"   # big constant row gathers deferred into the consuming Linear
_BIG_GATHER = 1 << 24          # elements of the gathered-from tensor from which a row gather is worth deferring (64 MB of float32) ...
_MIN_GATHER_ROWS = 16384       # ... and rows gathered (the planes forms of ops.linear_rows start there); tests lower both
MERGE_ADAM_GROUPS = os.environ.get("EGNN_ACCEL_MERGE_ADAM", "1") != "0"   # equal-option Adam groups stepped with one fused launch
LAZY = True     # BatchNorm1d.forward returns a deferred activation (efficient_gnns_amd/lazy.py); False: one launch per torch call, as in round 5


def enabled() -> bool:
    return bool(_orig)


def enable() -> None:
    if _orig:
        return
    from efficient_gnns_amd import ops
    bn_forward, lin_forward = torch.nn.BatchNorm1d.forward, torch.nn.Linear.forward
    _orig.update(bn=bn_forward, lin=lin_forward)

    from efficient_gnns_amd import lazy as L
    from efficient_gnns_amd import nn as PN
    from efficient_gnns_amd.sparse import SparseTensor

    def fast_bn(self, x):
        if LAZY and isinstance(x, L.LazyConv) and x._value is None and not self.training and not torch.is_grad_enabled() \
                and type(self) is torch.nn.BatchNorm1d and self.track_running_stats and self.weight is not None and self.bias is not None:
            return L.LazyFold(x, self)       # inference: the BatchNorm folds into the deferred conv (+ the ReLU that follows)
        x = L.materialise(x)
        prod = getattr(x, "_egnn_producer", None)
        if prod is not None and type(self) is torch.nn.BatchNorm1d:
            _NEXT_BN[prod] = weakref.ref(self)         # learned: this conv feeds this BatchNorm (decides deferral / statistics only)
        if (not _double_backward_wanted(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight is not None and self.bias is not None
                and self.track_running_stats and self.running_mean is not None and x.shape[0] > 1 and ops._bn_shape_ok(ops._rowmajor(x))):
            if LAZY:
                # statistics + running-statistics update now; the apply pass waits for the F.relu / F.dropout / consumer that follow (lazy.py)
                out = L.LazyBnAct.from_bn(self, x, self.training)
                if out is not None:
                    return out
            return ops.bn_act(x, self, relu=False, p=0.0, training=self.training)
        return bn_forward(self, x)

    def fast_linear(self, x):
        if isinstance(x, L.LazyRows) and x._value is None and not _double_backward_wanted(x) and self.weight.dtype == torch.float32 and x.shape[0] > 0:
            return x.linear(self.weight, self.bias)          # Linear(h[idx]) as the gather-fused GEMM (gnn.py:150)
        x = L.materialise(x)
        if not _double_backward_wanted(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and x.shape[0] > 0:
            return ops.linear(x, self.weight, self.bias)
        return lin_forward(self, x)
    torch.nn.BatchNorm1d.forward = fast_bn
    torch.nn.Linear.forward = fast_linear

    # the package's own convs (what the script's `from torch_geometric.nn import GCNConv, SAGEConv` resolves to) as CONSUMERS of a deferred
    # activation: one fused launch; for a GCNConv with a narrow output together with its h @ W (ops.bn_act_linear, gnn.py:47-52)
    gcn_forward, sage_forward = PN.GCNConv.forward, PN.SAGEConv.forward
    _orig.update(gcn=gcn_forward, sage=sage_forward)

    def gcn_consume(self, x, edge_index, *a, **kw):
        if isinstance(x, L.LazyBnAct) and x._value is None and not a and not kw and isinstance(edge_index, SparseTensor) \
                and self.in_channels >= self.out_channels and self._cached_ax is None:
            both = x.materialise_with_linear(self.weight)
            if both is not None:
                out = gcn_forward(self, both[0], edge_index, xw=both[1])
                out._egnn_producer = self
                return out
        x = L.materialise(x)
        nb = _NEXT_BN.get(self)
        bn = nb() if nb is not None else None
        plain = LAZY and not a and not kw and isinstance(edge_index, SparseTensor) and bn is not None and x.is_cuda and x.dim() == 2
        if plain and DEFER_CONV and not torch.is_grad_enabled() and not self.training and not bn.training and not self._uses_memoised_input(x):
            return L.LazyConv(self, x, edge_index, gcn_forward)       # inference: deferred until its consumer is known (lazy.py)
        if plain and self.training and bn.training and self.in_channels >= self.out_channels and bn.running_mean is not None:
            # training: the BatchNorm this conv is known to feed gets its statistics from the aggregation's epilogue (a tagged by-product
            # of the same output; ignored by any other consumer)
            out = gcn_forward(self, x, edge_index, bn_stats_shift=bn.running_mean, want_bn_stats=True)
        else:
            out = gcn_forward(self, x, edge_index, *a, **kw)
        out._egnn_producer = self
        return out

    def sage_consume(self, x, edge_index, *a, **kw):
        return sage_forward(self, L.materialise(x), edge_index, *a, **kw)
    PN.GCNConv.forward = gcn_consume
    PN.SAGEConv.forward = sage_consume
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

    # gnn.py:308-312 hands Adam three parameter groups with the SAME options; the fused implementation launches once per group (latency-bound
    # on 0.5 M parameters).  For the duration of step() the groups are presented as one -- the per-parameter state, ``param_groups`` and
    # ``state_dict()`` of the optimizer stay exactly what the script built (the view is restored before step() returns).
    adam_step = torch.optim.Adam.step
    _orig["adam_step"] = adam_step

    def _same(a, b):
        return a is b if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor) else a == b

    def merged_step(self, closure=None):
        groups = self.param_groups
        if MERGE_ADAM_GROUPS and len(groups) > 1 and all(g.get("fused") for g in groups) and not any(g.get("differentiable") for g in groups):
            first = groups[0]
            keys = [k for k in first if k != "params"]
            if all(set(g) == set(first) and all(_same(g[k], first[k]) for k in keys) for g in groups[1:]):
                merged = dict(first)
                merged["params"] = [p for g in groups for p in g["params"]]
                self.param_groups = [merged]
                try:
                    return adam_step(self, closure)
                finally:
                    self.param_groups = groups
        return adam_step(self, closure)
    torch.optim.Adam.step = merged_step

    # gnn.py:155 ``teacher_proj(teacher_out_feat[train_idx])``: a 273 MB gather of a tensor that never changes, followed by GEMMs on the fresh
    # copy.  A row gather of a LARGE float32 GPU leaf that takes no part in autograd, by a 1-D int64 GPU index, inside a grad-enabled region
    # (the train step) is deferred (lazy.LazyRows): a following Linear runs the gather-fused GEMM on the tensor's once-cut bf16 planes
    # (ops.linear_rows(..., const_input=True): keyed on identity + version, so an in-place change of the tensor re-cuts them); any other
    # consumer gets the gathered rows.  Everything else indexes as before.
    getitem = torch.Tensor.__getitem__
    _orig["getitem"] = getitem

    def deferring_getitem(self, idx):
        if (DEFER_CONST_GATHER and LAZY and type(idx) is torch.Tensor and type(self) is torch.Tensor and idx.dtype == torch.int64 and idx.dim() == 1
                and self.dim() == 2 and self.dtype == torch.float32 and self.is_cuda and idx.is_cuda and not self.requires_grad and self.grad_fn is None
                and self.shape[0] * self.shape[1] >= _BIG_GATHER and idx.numel() >= _MIN_GATHER_ROWS and torch.is_grad_enabled()):
            return L.LazyRows(self, idx, const_base=True)
        return getitem(self, idx)
    torch.Tensor.__getitem__ = deferring_getitem


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
    from efficient_gnns_amd import nn as PN
    PN.GCNConv.forward = _orig.pop("gcn")
    PN.SAGEConv.forward = _orig.pop("sage")
    torch.optim.Adam.__init__ = _orig.pop("adam")
    torch.optim.Adam.step = _orig.pop("adam_step")
    torch.Tensor.__getitem__ = _orig.pop("getitem")
