"""Upper bound of ANY source-window schedule of the K = 256 aggregation, measured before a kernel is written.

The gather probe (egnn_probe_gather_lines_f32: the call's line requests with the kernel's slice <-> XCD binding, no values, no
reduction, no output) walks the entry list front to back with all workgroups resident and striding together -- a lock-step sweep.
Handing it the SAME entries SORTED by (row round, source window) therefore replays the gather stream of an ideal window schedule:
R rounds of N / R rows whose partial sums stay on chip, inside a round the sources visited window by window (W windows of N / W
source rows = 21.7 MB / W per XCD slice... of X), every workgroup in the same window at the same time, no row pointers, no per-
(row, window) segment overhead, no Y.  A real kernel cannot gather faster than this.  R = 1, W = 1 is the CSR order (the ceiling
bench.py reports).      python tools/checks/window_gather_probe.py > gpurun_out/.../window_gather_probe.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efficient_gnns_amd as E  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
from efficient_gnns_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
K = 256
d = D.arxiv_like(1.0, seed=0)
adj = E.gcn_norm(d.adj_t.to(dev))
rowptr, col, _ = adj.csr()
n = adj.sparse_sizes()[0]
nnz = col.numel()
rows = torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1])
x = torch.randn(n, K, device=dev)
sink = torch.zeros(1, device=dev)
print(f"# headline graph: n = {n}, stored entries = {nnz}; gather stream = {nnz * K * 4 / 1e9:.2f} GB of 128-byte lines; X = {n * K * 4 / 1e6:.1f} MB "
      f"({n * 128 / 1e6:.1f} MB per XCD slice); grid 8 x 256 workgroups (all resident: one sweep)")
print("# rounds R  windows W  window_MB_per_XCD  best_us  lines_TBs  (loads in flight)")
for R in (1, 2, 3, 5, 8, 16):
    for W in (1, 2, 4, 8, 16, 32):
        key = torch.div(rows * R, n, rounding_mode="floor") * W + torch.div(col * W, n, rounding_mode="floor")
        order = torch.argsort(key, stable=True)
        c32 = col[order].to(torch.int32).contiguous()
        best = None
        for inflight in (4, 8, 16):
            def run():
                _lib.check(lib.egnn_probe_gather_lines_f32(_lib.ptr(x), x.stride(0), n, K, _lib.ptr(c32), nnz, 256, inflight, _lib.ptr(sink),
                                                           _lib.stream()), "egnn_probe_gather_lines_f32")
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 10
            if best is None or us < best[0]:
                best = (us, inflight)
        print(f"{R:9d}  {W:9d}  {n * 128 / W / 1e6:17.2f}  {best[0]:7.1f}  {nnz * K * 4 / best[0] / 1e6:9.2f}  ({best[1]})", flush=True)
