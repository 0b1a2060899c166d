"""Audit of the operators a step issues before it is captured into a hipGraph.

Why: torch's multi-block reductions (ATen/native/cuda/Reduce.cuh) zero their inter-block semaphores with a
``cudaMemsetAsync`` -- a MEMSET NODE once captured -- and never reset them in the kernel.  Inside the replayed train step on
this stack (ROCm 7.2, gfx950) such reductions were seen to leave their output unwritten: ``kl_div(..., 'mean')`` over 680 k
edges returned the stale bytes of an earlier workspace, 368.4 instead of 1.9e-5, in EVERY replay when the first launch found
the device idle (profiles/r04_lsp_trace.txt).  The step therefore forms every long sum in this package's own fixed-order
kernels, and ``GraphedEpoch`` refuses to capture a step that still contains a long torch reduction.
"""
from __future__ import annotations

import os
import warnings

import torch
from torch.utils._python_dispatch import TorchDispatchMode

# aten operators that run (or may run) through gpu_reduce_kernel
_REDUCTIONS = {"sum", "mean", "amax", "amin", "max", "min", "prod", "any", "all", "norm", "linalg_vector_norm", "logsumexp", "var",
               "std", "var_mean", "std_mean", "nansum", "argmax", "argmin", "count_nonzero", "mse_loss", "l1_loss", "smooth_l1_loss",
               "huber_loss", "binary_cross_entropy", "binary_cross_entropy_with_logits", "kl_div", "_log_softmax", "_softmax",
               "nll_loss_forward", "dot", "vdot", "trace", "median", "aminmax"}
LONG = 2048   # reduced elements per output from which a reduction may be split over blocks


class LongReductionInCapture(RuntimeError):
    pass


class CaptureAudit(TorchDispatchMode):
    """Records (operator, reduced length, input shape) of every torch reduction over GPU tensors whose reduced length per output
    is >= ``LONG``."""

    def __init__(self, any_device: bool = False):
        super().__init__()
        self.flagged: list = []
        self.any_device = any_device   # CPU tensors count too (the CPU test of this class)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name in _REDUCTIONS:
            src = next((a for a in args if isinstance(a, torch.Tensor)), None)
            res = out[0] if isinstance(out, (tuple, list)) else out
            if src is not None and (src.is_cuda or self.any_device) and isinstance(res, torch.Tensor):
                per_out = src.numel() // max(res.numel(), 1)
                # row-wise softmax / losses reduce along the last dimension only
                if name in ("_log_softmax", "_softmax"):
                    per_out = src.shape[-1] if src.dim() else 1
                if per_out >= LONG:
                    self.flagged.append((name, per_out, tuple(src.shape)))
        return out

    def check(self, what: str) -> None:
        if self.flagged and os.environ.get("EGNN_GRAPH_AUDIT", "1") == "0":
            # the reproducer of the finding (tools/checks/lsp_trace.py VARIANT=refs GRAPH=1 SYNC=1) captures such a step on purpose
            warnings.warn(f"{what}: long torch reductions in a captured step (EGNN_GRAPH_AUDIT=0): {self.flagged[:4]}")
            return
        if self.flagged:
            ops = ", ".join(f"aten.{n} over {k} elements of {s}" for n, k, s in self.flagged[:6])
            raise LongReductionInCapture(
                f"{what}: the step contains long torch reductions ({ops}); their multi-block form relies on a memset node that does not "
                "take effect reliably in replayed hipGraphs on this stack -- use the package's fixed-order kernels (ops.colsum, "
                "ops_edge._LspLoss, ...) or run the step with eager launches")


_HIP_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


def graph_node_kinds(graph: "torch.cuda.CUDAGraph") -> dict:
    """{node type: count, "edges": n, "chain": bool} of a captured graph that was created with ``keep_graph=True`` -- read through the
    HIP runtime (hipGraphGetNodes / hipGraphNodeGetType / hipGraphGetEdges).  The shipped epochs are chains of kernel nodes only; a
    ``memset`` node is what a long torch reduction leaves behind (see the module docstring)."""
    import collections
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    g = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * n.value)()
    hip.hipGraphGetNodes(g, nodes, ctypes.byref(n))
    kinds = collections.Counter()
    for nd in nodes:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
        kinds[_HIP_NODE_TYPES.get(t.value, str(t.value))] += 1
    ne = ctypes.c_size_t(0)
    hip.hipGraphGetEdges(g, None, None, ctypes.byref(ne))
    src, dst = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
    hip.hipGraphGetEdges(g, src, dst, ctypes.byref(ne))
    indeg, outdeg = collections.Counter(dst), collections.Counter(src)
    chain = ne.value == n.value - 1 and all(indeg[x] <= 1 and outdeg[x] <= 1 for x in nodes)
    out = dict(kinds)
    out.update(edges=ne.value, chain=bool(chain))
    return out


def check_captured_graph(graph: "torch.cuda.CUDAGraph", what: str, kernels_only: bool) -> dict:
    """Structural guard, run by ``GraphedEpoch`` / ``ShardedGraphedEpoch`` on every capture (the graph must have been created with
    ``keep_graph=True`` and not be instantiated yet): raises ``LongReductionInCapture`` when the captured graph holds a node of a
    kind the shipped steps never produce.  ``kernels_only``: the single-GPU epoch -- kernel nodes, nothing else.
    Otherwise (the sharded epoch with its captured RCCL collectives): no memset and no host node; RCCL's own device-to-device copies
    (a one-rank all-gather / all-to-all is a memcpy node) and its fork / join edges are allowed.  Returns the node census."""
    kinds = graph_node_kinds(graph)
    bad = [k for k in kinds if k not in ("kernel", "edges", "chain") and (kernels_only or k in ("memset", "host")) and kinds[k]]
    # (the dependency structure is NOT a criterion: autograd adds cross-stream ordering edges when parameters were created on another
    # stream than the capture's -- more edges than a chain, never fewer; ``chain`` is reported, the shipped configurations are chains)
    if bad:
        raise LongReductionInCapture(
            f"{what}: the captured graph holds {', '.join(f'{kinds[k]} {k} node(s)' for k in bad)} "
            f"({kinds}); the shipped steps are kernel nodes only -- a memset / memcpy / host node comes from a torch operator with hidden "
            "scratch traffic (long reductions, sort, index_add_, bincount, ...), whose effect inside replayed hipGraphs is not reliable "
            "on this stack (_audit.py): use the package's kernels for that operator or run the step with eager launches")
    return kinds
