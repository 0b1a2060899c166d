#!/usr/bin/env python3
"""Share of the stored entries of the (GCN-normalised) headline adjacency whose SOURCE is among the H highest-degree nodes:
the upper bound on what an LDS-resident table of hub rows could take off the L2 request path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efficient_gnns_amd.data as D  # noqa: E402

n = D.ARXIV["num_nodes"]
src, dst = D.powerlaw_edges(n, D.ARXIV["num_edges"], max_degree=D.ARXIV["max_degree"], seed=0)
key = np.unique(np.concatenate([src * n + dst, dst * n + src, np.arange(n) * n + np.arange(n)]))
col = key % n
deg = np.bincount(col, minlength=n)
order = np.argsort(-deg)
cover = np.cumsum(deg[order]) / key.size
print(f"stored entries {key.size}")
for H in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    print(f"H = {H:6d}: {100 * cover[H - 1]:5.1f} % of the entries, degree at rank H = {deg[order[H - 1]]}")
