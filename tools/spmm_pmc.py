#!/usr/bin/env python3
"""Workload for the PMC passes: a calibration copy of known size, then the K=256 GCN aggregation, a few launches each.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o fetch -- python tools/spmm_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -o write -- python tools/spmm_pmc.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import efficient_gnns_amd as E  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
import efficient_gnns_amd.ops as ops  # noqa: E402

ops._OVERLAP_HEAVY_ROWS = False  # one stream: the three launches of a call appear back to back in the trace
d = D.arxiv_like(1.0, seed=0, with_teacher=False)
gn = E.gcn_norm(d.adj_t.to("cuda"))
x = torch.randn(d.num_nodes, 256, device="cuda")
src = torch.randn(64 * 1024 * 1024, device="cuda")  # 256 MiB
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(src)          # calibration: 268 435 456 B read + 268 435 456 B written per launch
torch.cuda.synchronize()
for _ in range(5):
    ops.spmm_raw(gn, x, "sum")
torch.cuda.synchronize()
print("alg_bytes", gn.spmm_algorithmic_bytes(256), "nnz", gn.nnz())
