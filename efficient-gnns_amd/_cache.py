"""One memo class for the package's integer preprocessing (edge plans, inverse row maps, composed edge lists, split id maps, ...).

Keyed on the IDENTITY of index tensors (storage address, version counter, shape, dtype) plus extra hashables.  Every entry keeps
its key tensors alive: while it lives the allocator cannot hand the same address to another tensor, so an address match means
the same data.  Eviction is least-recently-used, one entry at a time -- and an entry that a captured hipGraph reads through raw
pointers outlives its eviction: inside ``pinning()`` (GraphedEpoch / ShardedGraphedEpoch hold it open around warm-up + capture)
every value handed out is also appended to the caller's list, which the graph object keeps for its lifetime.
"""
from __future__ import annotations

import contextlib
from collections import OrderedDict

_RECORDERS: list = []


class TensorKeyedCache:
    def __init__(self, capacity: int = 16, max_bytes: int | None = None):
        """``max_bytes``: also bound the sum of the values' tensor bytes (the newest entry always stays, whatever its size)."""
        self.capacity, self.max_bytes, self.store = capacity, max_bytes, OrderedDict()

    @staticmethod
    def _nbytes(value) -> int:
        if hasattr(value, "element_size") and hasattr(value, "numel"):
            return value.numel() * value.element_size()
        if isinstance(value, (tuple, list)):
            return sum(TensorKeyedCache._nbytes(v) for v in value)
        return 0

    @staticmethod
    def _key(tensors, extra):
        return tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype, str(t.device)) for t in tensors) + tuple(extra)

    def get(self, tensors, extra, build):
        key = self._key(tensors, extra)
        hit = self.store.get(key)
        if hit is None:
            hit = self.store[key] = (tuple(tensors), build())
            while len(self.store) > self.capacity:
                self.store.popitem(last=False)
            if self.max_bytes is not None:
                while len(self.store) > 1 and sum(self._nbytes(v[1]) for v in self.store.values()) > self.max_bytes:
                    self.store.popitem(last=False)
        else:
            self.store.move_to_end(key)
        for rec in _RECORDERS:
            rec.append(hit)
        return hit[1]

    def clear(self):
        self.store.clear()

    def __len__(self):
        return len(self.store)


@contextlib.contextmanager
def pinning():
    """Collects every cache value handed out inside the block (keys included): keep the yielded list as long as raw pointers to
    those values may be in use (a captured graph)."""
    rec: list = []
    _RECORDERS.append(rec)
    try:
        yield rec
    finally:
        _RECORDERS.remove(rec)
