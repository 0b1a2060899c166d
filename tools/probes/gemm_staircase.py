"""Does the 128x128-tile GEMM time follow a staircase in M (rounds of 768 resident workgroups)?  (tools only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import efficient_gnns_amd.ops as ops
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
w = torch.randn(256, 256, device=dev)
for tiles_m in (384, 768, 1152, 1200, 1323, 1536, 1900, 2304):
    M = tiles_m * 128
    x = torch.randn(M, 256, device=dev)
    us = t(lambda: ops.gemm_raw(x, w, False, True))
    print(f"M={M:7d} tiles={tiles_m * 2:5d} rounds={tiles_m * 2 / 768:5.2f}  {us:7.1f} us  {2 * M * 256 * 256 / us / 1e6:6.1f} TF/s  us/round={us / (tiles_m * 2 / 768):6.1f}")
