"""Host-staged collectives: the sharded step's ``torch.distributed`` calls on GPU tensors over a CPU transport (gloo).

Why it exists.  RCCL refuses two ranks on one device, so on a one-GPU box the node-range sharded step (dist.py) could only
ever run with world_size 1 -- an EMPTY halo.  With this transport W processes share ``cuda:0``: every rank runs the real HIP
kernels on its shard, and each collective of the step (halo ``all_to_all_single``, SyncBN ``all_reduce`` / ``all_gather``,
the G-CRD sample gather with its ``reduce_scatter``, the flat gradient ``all_reduce``) is carried by gloo through (pageable)
host memory: device -> host copy, the collective on the host tensor, host -> device copy, all ordered on the current
stream.  Same collective program, same payloads, same order as the RCCL run -- only the wire differs (``CommTrace`` sees
the calls above this layer).  It is a verification / bring-up transport (eager launches only: a host round trip cannot be
captured into a hipGraph), not a performance path: the multi-GPU product transport is RCCL over xGMI.

``install()`` wraps the collectives in the ``torch.distributed`` namespace (what dist.py / ops.py call); CPU tensors and
groups on a device-capable backend pass through untouched.  ``uninstall()`` restores them.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

_ORIG: dict = {}
_STATS = dict(calls=0, bytes_staged=0)
_STAGE_HOST_TENSORS = False      # tests (CPU, gloo): run host tensors through the staging code as well (its copies are then no-ops)


class _Done:
    """What an ``async_op=True`` call returns: the staged collective has already completed on the host."""

    def wait(self, timeout=None):
        return True

    def is_completed(self):
        return True


def active() -> bool:
    return bool(_ORIG)


def stats() -> dict:
    return dict(_STATS)


def _staged(group, t: torch.Tensor) -> bool:
    """True for a device tensor on a group whose backend cannot take one (gloo): stage through the host."""
    if not (t.is_cuda or _STAGE_HOST_TENSORS):
        return False
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:  # noqa: BLE001  (no default group yet)
        return False


def _down(t: torch.Tensor) -> torch.Tensor:
    """A host copy of a device tensor, ordered behind everything enqueued on the current stream."""
    _STATS["bytes_staged"] += t.numel() * t.element_size()
    h = t.detach().contiguous()
    return h.cpu() if h.is_cuda else h.clone()        # .cpu(): a synchronous device -> host copy on the current stream


def _up(dst: torch.Tensor, src: torch.Tensor) -> None:
    if dst.numel():
        dst.copy_(src.view(dst.shape) if src.shape != dst.shape else src)


def _finish(async_op: bool):
    _STATS["calls"] += 1
    return _Done() if async_op else None


def _all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if not _staged(group, tensor):
        return _ORIG["all_reduce"](tensor, op=op, group=group, async_op=async_op)
    h = _down(tensor)
    _ORIG["all_reduce"](h, op=op, group=group)
    _up(tensor, h)
    return _finish(async_op)


def _broadcast(tensor, src=None, group=None, async_op=False, group_src=None):
    if not _staged(group, tensor):
        kw = {} if group_src is None else {"group_src": group_src}
        return _ORIG["broadcast"](tensor, src=src, group=group, async_op=async_op, **kw)
    h = _down(tensor)
    kw = {} if group_src is None else {"group_src": group_src}
    _ORIG["broadcast"](h, src=src, group=group, **kw)
    _up(tensor, h)
    return _finish(async_op)


def _all_gather_into_tensor(output_tensor, input_tensor, group=None, async_op=False):
    if not _staged(group, input_tensor):
        return _ORIG["all_gather_into_tensor"](output_tensor, input_tensor, group=group, async_op=async_op)
    # gloo wants the output as the flat concatenation of the (flat) inputs, whatever shape the caller gave it ([world, C] for a [C]
    # input is fine over RCCL, refused here); the blocks are contiguous either way
    hin = _down(input_tensor).reshape(-1)
    hout = torch.empty(output_tensor.numel(), dtype=output_tensor.dtype)
    _ORIG["all_gather_into_tensor"](hout, hin, group=group)
    _up(output_tensor, hout)
    return _finish(async_op)


def _all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
    if not _staged(group, input):
        return _ORIG["all_to_all_single"](output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
    hin = _down(input)
    hout = torch.empty(output.shape, dtype=output.dtype)       # (split sizes count rows of dim 0: the shapes stay as given)
    _ORIG["all_to_all_single"](hout, hin, output_split_sizes, input_split_sizes, group=group)
    _up(output, hout)
    return _finish(async_op)


def _reduce_scatter_tensor(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if not _staged(group, input):
        return _ORIG["reduce_scatter_tensor"](output, input, op=op, group=group, async_op=async_op)
    # gloo has no reduce-scatter: the sum of all blocks on the host, this rank's block back to the device (same result as
    # RCCL's reduce_scatter up to the summation order of the ranks' contributions)
    h = _down(input)
    _ORIG["all_reduce"](h, op=op, group=group)
    r = dist.get_rank(group)
    n = output.numel()
    _up(output, h.reshape(-1)[r * n:(r + 1) * n].view(output.shape))
    return _finish(async_op)


_WRAPPERS = dict(all_reduce=_all_reduce, broadcast=_broadcast, all_gather_into_tensor=_all_gather_into_tensor,
                 all_to_all_single=_all_to_all_single, reduce_scatter_tensor=_reduce_scatter_tensor)


def install() -> None:
    """Route the collectives dist.py / ops.py issue on GPU tensors through the host whenever the process group is gloo."""
    if _ORIG:
        return
    for name, fn in _WRAPPERS.items():
        _ORIG[name] = getattr(dist, name)
        setattr(dist, name, fn)


def uninstall() -> None:
    for name, fn in _ORIG.items():
        setattr(dist, name, fn)
    _ORIG.clear()
