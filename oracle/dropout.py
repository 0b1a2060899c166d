"""TEST INFRASTRUCTURE (oracle side; never imported by the product package).

Training-mode parity needs the SAME dropout masks on both sides.  The reference draws them from torch's generator inside
``F.dropout`` (/root/reference/arxiv_pyg/gnn.py:50,83: ``F.dropout(x, p=self.dropout, training=self.training)``); the HIP path
draws them from a counter hash of (seed, element index) inside the fused BatchNorm kernels
(efficient-gnns_amd/csrc/bn_common.h ``uniform01``).  Neither generator can reproduce the other, so the parity harness fixes the
masks instead: ``counter_mask`` restates the kernel's hash in NumPy (checked against the kernel's own output in
tests/test_gpu_training_parity.py), and ``injected_dropout`` hands those masks, in call order, to the oracle's ``F.dropout``.
With the masks fixed, multi-step training trajectories of the two sides are directly comparable (SURVEY.md 7.3 "RNG coupling").
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch
import torch.nn.functional as F


_M32 = 0xFFFFFFFF


def _mix32(x: torch.Tensor) -> torch.Tensor:
    """lowbias32 on int64 tensors holding 32-bit values (products wrap mod 2^64; the low 32 bits are exact)."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    return x ^ (x >> 16)


def counter_uniform(seed: int, n: int, C: int) -> np.ndarray:
    """u[r, c] in [0, 1) of element index r * C + c under the 64-bit ``seed`` (bn_common.h ``uniform01``); float32 [n, C]."""
    seed &= 0xFFFFFFFFFFFFFFFF
    idx = torch.arange(n * C, dtype=torch.int64)
    lo, hi = idx & _M32, idx >> 32
    h = _mix32(lo ^ (seed & _M32))
    h = _mix32((h + (seed >> 32) + ((hi * 0x9E3779B9) & _M32)) & _M32)
    return ((h >> 8).to(torch.float32) * (1.0 / 16777216.0)).reshape(n, C).numpy()


def counter_mask(seed: int, n: int, C: int, p: float) -> torch.Tensor:
    """The multiplicative dropout mask of the fused kernels: 1 / (1 - p) where u >= p, else 0 (bn_common.h ``bn_elem``)."""
    u = counter_uniform(seed, n, C)
    keep = u >= np.float32(p)
    return torch.from_numpy(np.where(keep, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32))


@contextlib.contextmanager
def injected_dropout(masks):
    """Within the block every training-mode ``F.dropout(x, p > 0)`` multiplies by the next mask of ``masks`` (an iterable of
    float tensors shaped like x) instead of drawing one; eval-mode / p = 0 calls pass through.  Raises if a call finds no mask
    left or a mask of the wrong shape; the caller checks that every mask was consumed (``left()`` of the yielded object)."""
    queue = list(masks)
    orig = F.dropout

    class _State:
        used = 0

        @staticmethod
        def left():
            return len(queue) - _State.used

    def dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return x
        if _State.used >= len(queue):
            raise AssertionError("injected_dropout: more dropout calls than masks")
        m = queue[_State.used]
        _State.used += 1
        if tuple(m.shape) != tuple(x.shape):
            raise AssertionError(f"injected_dropout: mask {tuple(m.shape)} for activations {tuple(x.shape)}")
        return x * m.to(x.dtype)

    F.dropout = dropout
    try:
        yield _State
    finally:
        F.dropout = orig
