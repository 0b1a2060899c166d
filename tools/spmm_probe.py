#!/usr/bin/env python3
"""SpMM kernel vs memory-system probe: regular-degree synthetic CSRs whose columns are confined to a window."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import efficient_gnns_amd as E
import efficient_gnns_amd.ops as ops
from kernel_bench import timeit

n, deg, K = 169343, 15, 256
x = torch.randn(n, K, device="cuda")
for window in (4096, 32768, 131072, n):
    g = torch.Generator(device="cuda").manual_seed(0)
    col = torch.randint(0, window, (n, deg), device="cuda", generator=g)
    col, _ = torch.sort(col, dim=1)
    rowptr = torch.arange(0, n * deg + 1, deg, device="cuda")
    val = torch.rand(n * deg, device="cuda")
    adj = E.SparseTensor(rowptr=rowptr, col=col.reshape(-1), value=val, sparse_sizes=(n, n))
    for plan in (True, False):
        t = timeit(lambda: ops.spmm_raw(adj, x, "sum", use_plan=plan))
        print(json.dumps(dict(window=window, plan=plan, us=round(t * 1e6, 1), gather_GBs=round(n * deg * K * 4 / t / 1e9, 1))), flush=True)
