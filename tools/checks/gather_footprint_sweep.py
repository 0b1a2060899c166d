"""Where do the gathered lines of the aggregation come from?  The K = 256 gather stream (egnn_probe_gather_lines_f32: the kernel's
line requests, no reduction, no output) over source matrices of growing footprint, uniform random columns, fixed number of
requests: below the 256 MiB Infinity Cache (MALL) the L2 misses are served by the cache, above it by HBM.  The aggregation's
FETCH_SIZE counter counts both alike (L2-miss fabric requests) -- this sweep is the split.

    python tools/checks/gather_footprint_sweep.py > gpurun_out/.../gather_footprint.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficient_gnns_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
K, nnz = 256, 2_501_771                      # the headline call's width and (with self loops) entry count
g = torch.Generator(device=dev).manual_seed(0)
print(f"# gather stream of {nnz} rows x {K} floats = {nnz * K * 4 / 1e9:.2f} GB of 128-byte line requests per call; uniform random source rows")
print("# footprint_MB  rows  best_us  request_GBs  (grid)")
for n_src in (21_168, 42_336, 84_672, 169_343, 262_144, 338_686, 677_372, 1_354_744):
    x = torch.randn(n_src, K, device=dev)
    col = torch.randint(0, n_src, (nnz,), device=dev, generator=g, dtype=torch.int32)
    sink = torch.zeros(1, device=dev)
    best = None
    for bps, inflight in ((256, 8), (512, 8), (1024, 8), (512, 16), (1024, 4)):
        def run():
            _lib.check(lib.egnn_probe_gather_lines_f32(_lib.ptr(x), x.stride(0), n_src, K, _lib.ptr(col), nnz, bps, inflight, _lib.ptr(sink),
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
            best = (us, bps, inflight)
    print(f"{n_src * K * 4 / 1e6:10.1f}  {n_src:9d}  {best[0]:8.1f}  {nnz * K * 4 / best[0] / 1e3:9.1f}  ({best[1]} x {best[2]})", flush=True)
    del x, col
