"""Deferred activations for the reference's UNMODIFIED model code (dropin/accel.py).

``arxiv_pyg/gnn.py:47-51`` spells a hidden layer as four calls of its own::

    x = conv(x, adj_t); x = self.bns[i](x); x = F.relu(x); x = F.dropout(x, p=self.dropout, training=self.training)

and ``gnn.py:150`` reads ``student_proj(model.out_feat[train_idx])``.  The package's own loop runs each of these groups as ONE
launch (``ops.bn_act`` / ``ops.bn_act_linear`` / ``ops.linear_rows``); the script calls them by name, one at a time.  Under
``accel.enable()`` the re-pointed ``BatchNorm1d.forward`` therefore does the part that must happen NOW (batch statistics + the
module's running-statistics update) and returns a ``LazyBnAct``: a tensor-like object (``__torch_function__`` protocol) that

  * absorbs a following ``F.relu`` and ``F.dropout`` (any other torch function, method or operator materialises it first and then
    runs on the real tensor -- nothing is ever silently skipped);
  * is materialised by its first real consumer with ONE fused launch -- by the next ``GCNConv`` together with its narrow ``h @ W``
    (``ops.bn_act_linear``) when the shapes allow, by a sampled criterion as only the sampled rows (``pick``);
  * answers ``lazy[idx]`` for a 1-D int64 index with a ``LazyRows`` that a following ``Linear`` turns into the gather-fused GEMM
    (``ops.linear_rows``: no ``[N_tr, 256]`` copy forward, no zero-fill + scatter backward).

Inference (``model.eval()`` under ``no_grad``): a ``GCNConv`` that has been SEEN feeding a ``BatchNorm1d`` (``accel`` learns the association
from the first forward; it only decides what is deferred) hands out a ``LazyConv``; the eval-mode BatchNorm turns it into a ``LazyFold``
that absorbs the ReLU and is formed as ONE pass -- the BatchNorm folded into the conv's weights, the ReLU in the last kernel's store.
In training the same association gives the BatchNorm its statistics out of the aggregation's epilogue.

Same arithmetic as ``models.train_step`` (the kernels are the same); dropout masks come from the same counter hash (one host seed per
materialised dropout).  Consumers inside the package ask through ``materialise(x)`` / the ``_egnn_materialise`` attribute.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


def materialise(x, pick=None):
    """``x`` as a real tensor (``x[pick]`` when ``pick`` is given): deferred objects are formed now, tensors pass through."""
    m = getattr(x, "_egnn_materialise", None)
    if m is not None:
        return m(pick=pick)
    return x if pick is None else x[pick]


class _TensorLike:
    """Everything a consumer might do with a tensor that this module does not defer: materialise, then delegate."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = cls._absorb(func, args, kwargs)
        if out is not NotImplemented:
            return out
        real = lambda v: v._egnn_materialise() if isinstance(v, _TensorLike) else v   # noqa: E731
        args = tuple(type(a)(real(v) for v in a) if isinstance(a, (list, tuple)) else real(a) for a in args)
        kwargs = {k: real(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)

    @classmethod
    def _absorb(cls, func, args, kwargs):
        return NotImplemented

    def __getattr__(self, name):       # methods / attributes of the real tensor (.t(), .sum(), .requires_grad, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._egnn_materialise(), name)

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self._egnn_materialise())

    def __repr__(self):
        return f"{type(self).__name__}(shape={tuple(self.shape)}, deferred={self._value is None})"

    def dim(self):
        return len(self.shape)

    def size(self, d=None):
        return torch.Size(self.shape) if d is None else self.shape[d]

    dtype = torch.float32
    is_cuda = True


def _binary(name):
    def op(self, other):
        return getattr(self._egnn_materialise(), name)(materialise(other))
    op.__name__ = name
    return op


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__matmul__", "__rmatmul__",
           "__pow__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__"):
    setattr(_TensorLike, _n, _binary(_n))
_TensorLike.__neg__ = lambda self: -self._egnn_materialise()
_TensorLike.__hash__ = object.__hash__


class LazyBnAct(_TensorLike):
    """``dropout(relu?(bn(x)), p)`` with the statistics already taken; see the module docstring."""

    def __init__(self, src, relu=False, p=0.0):
        self._src, self._relu, self._p, self._value = src, relu, p, None

    @staticmethod
    def from_bn(bn, x, training):
        """The part of ``BatchNorm1d.forward`` that cannot wait: statistics (or the producing aggregation's), running-statistics update.
        None when the fused kernels do not take the shape (the caller falls back to torch's forward)."""
        prep = ops._bn_prepare(x, bn, 0.0, training)
        if prep is None:
            return None
        xr, mean, var, use_batch, _, _ = prep
        return LazyBnAct(dict(x=xr, bn=bn, mean=mean, var=var, use_batch=use_batch, training=training))

    shape = property(lambda self: self._src["x"].shape)
    device = property(lambda self: self._src["x"].device)

    @classmethod
    def _absorb(cls, func, args, kwargs):
        x = args[0] if args else None
        if not isinstance(x, LazyBnAct) or x._value is not None:
            return NotImplemented
        if func in (F.relu, torch.relu) and not kwargs.get("inplace", False) and len(args) == 1 and x._p == 0.0:
            return x if x._relu else LazyBnAct(x._src, True, 0.0)
        if func is F.dropout and not kwargs.get("inplace", False) and len(args) <= 3:
            p = args[1] if len(args) > 1 else kwargs.get("p", 0.5)
            training = args[2] if len(args) > 2 else kwargs.get("training", True)
            if not training or p == 0.0:
                return x                          # F.dropout outside training is the identity
            if x._p == 0.0 and 0.0 < p < 1.0:
                return LazyBnAct(x._src, x._relu, float(p))
        return NotImplemented

    def _args(self):
        s = self._src
        drop = self._p if (s["training"] and self._p > 0) else 0.0
        seed = ops._draw_dropout_seed() if drop > 0 else 0
        return s["x"], s["bn"].weight, s["bn"].bias, s["mean"], s["var"], s["bn"].eps, self._relu, drop, seed, s["use_batch"]

    def _egnn_materialise(self, pick=None):
        if self._value is None:
            if pick is not None:   # only the rows a sampled criterion keeps (not cached: the full tensor was never asked for)
                return ops._BnAct.apply(*self._args(), pick)
            v = ops._BnAct.apply(*self._args(), None)
            # two consumers are the rule for the last hidden state (the next conv, and student_proj(out_feat[train_idx])): the
            # row-compact gradient of the second joins the dense one of the first (ops.grad_tap)
            self._value = ops.grad_tap(v) if self._src["training"] else v
        return self._value if pick is None else self._value[pick]

    def materialise_with_linear(self, w):
        """(h, h @ w) in one op for a narrow ``w`` (``ops.bn_act_linear``'s kernels); None when they do not take the shape."""
        s = self._src
        x = s["x"]
        if self._value is not None or not (torch.is_grad_enabled() and s["training"] and s["use_batch"] and type(s["bn"]) is torch.nn.BatchNorm1d
                                           and w.dim() == 2 and w.shape[0] == x.shape[1] and w.shape[1] <= 64 and x.shape[1] % 64 == 0):
            return None
        box = ops._TapBox()
        h, xw = ops._BnActLinear.apply(*self._args(), w, box)
        h._egnn_tap = box
        self._value = h
        return h, xw

    def __getitem__(self, idx):
        if isinstance(idx, torch.Tensor) and idx.dtype == torch.int64 and idx.dim() == 1 and idx.is_cuda:
            return LazyRows(self, idx)
        return self._egnn_materialise()[idx]


class LazyConv(_TensorLike):
    """``conv(x, adj_t)`` of an inference forward (``model.eval()``, ``no_grad``), not run yet.  Only handed out by a ``GCNConv`` that has
    been SEEN to feed a ``BatchNorm1d`` (the association is learned from the first forward, and only decides what is deferred -- any
    consumer forms the same values): the BatchNorm turns it into a ``LazyFold``; everything else runs the conv now."""

    def __init__(self, conv, x, adj_t, run):
        self._conv, self._x, self._adj, self._run, self._value = conv, x, adj_t, run, None

    shape = property(lambda self: torch.Size((self._x.shape[0], self._conv.out_channels)))
    device = property(lambda self: self._x.device)

    def _egnn_materialise(self, pick=None):
        if self._value is None:
            self._value = self._run(self._conv, self._x, self._adj)
            self._value._egnn_producer = self._conv
        return self._value if pick is None else self._value[pick]

    def __getitem__(self, idx):
        return self._egnn_materialise()[idx]


class LazyFold(_TensorLike):
    """``bn(conv(x, adj_t))`` in eval mode, deferred: with a following ``F.relu`` absorbed it is formed by ONE pass -- the BatchNorm folded
    into the conv's weights and bias, the ReLU in the last kernel's store (``GCNConv.forward(eval_bn=)``: what ``models.evaluate`` runs;
    gnn.py:47-49 under ``model.eval()``) -- instead of conv + a normalisation pass + a ReLU pass."""

    def __init__(self, lazy_conv, bn, relu=False):
        self._lc, self._bn, self._relu, self._value = lazy_conv, bn, relu, None

    shape = property(lambda self: self._lc.shape)
    device = property(lambda self: self._lc.device)

    @classmethod
    def _absorb(cls, func, args, kwargs):
        x = args[0] if args else None
        if not isinstance(x, LazyFold) or x._value is not None:
            return NotImplemented
        if func in (F.relu, torch.relu) and not kwargs.get("inplace", False) and len(args) == 1:
            return x if x._relu else LazyFold(x._lc, x._bn, True)
        if func is F.dropout and not kwargs.get("inplace", False) and len(args) <= 3:
            training = args[2] if len(args) > 2 else kwargs.get("training", True)
            p = args[1] if len(args) > 1 else kwargs.get("p", 0.5)
            if not training or p == 0.0:
                return x
        return NotImplemented

    def _egnn_materialise(self, pick=None):
        if self._value is None:
            lc, bn = self._lc, self._bn
            if self._relu and lc._value is None and not torch.is_grad_enabled() and not bn.training:
                self._value = lc._run(lc._conv, lc._x, lc._adj, eval_bn=bn)   # fold + ReLU in the last kernel's store
            else:
                y = ops.bn_act(lc._egnn_materialise(), bn, relu=self._relu, p=0.0, training=False)
                self._value = y
        return self._value if pick is None else self._value[pick]

    def __getitem__(self, idx):
        if isinstance(idx, torch.Tensor) and idx.dtype == torch.int64 and idx.dim() == 1 and idx.is_cuda:
            return LazyRows(self, idx)
        return self._egnn_materialise()[idx]


class LazyRows(_TensorLike):
    """``base[idx]`` (unique 1-D int64 ids) not gathered yet: a ``Linear`` consumes it as ``ops.linear_rows(base, idx, W, b)``."""

    def __init__(self, base, idx, const_base=False):
        self._base, self._idx, self._value, self._const = base, idx, None, const_base

    shape = property(lambda self: torch.Size((self._idx.numel(),) + tuple(self._base.shape[1:])))
    device = property(lambda self: self._idx.device)

    def _egnn_materialise(self, pick=None):
        if self._value is None:
            base = materialise(self._base)
            if pick is not None:
                return ops.take_rows(base, self._idx[pick])
            self._value = ops.take_rows(base, self._idx)
        return self._value if pick is None else self._value[pick]

    def linear(self, weight, bias):
        base = materialise(self._base)
        # ``const_base``: the gathered-from tensor is a leaf outside autograd (accel's deferred constant gather): its rows as once-cut planes
        const = bool(self._const and not base.requires_grad and base.grad_fn is None)
        return ops.linear_rows(base, self._idx, weight, bias, const_input=const)

    def __getitem__(self, idx):
        if isinstance(idx, torch.Tensor) and idx.dtype == torch.int64 and idx.dim() == 1 and idx.is_cuda and self._value is None:
            return LazyRows(self._base, self._idx[idx], self._const)
        return self._egnn_materialise()[idx]
