#!/usr/bin/env python3
"""Writes the lab's graph files (git-ignored, shipped to the GPU box with the tree): the GCN-normalised CSR (self loops
inserted, values = PyG gcn_norm) of the synthetic ogbn-arxiv-shaped graphs, as the kernels see it.

    python tools/lab/make_graphs.py            # -> tools/lab/data/{chunglu,local}.bin

Layout (little endian): int64 n, nnz, n_comm | int32 rowptr[n+1] | int32 col[nnz] | float32 val[nnz] | int32 comm_ptr[n_comm+1]
(comm_ptr: row ranges of the communities when the ids are in community order; n_comm = 0 otherwise)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import efficient_gnns_amd.data as D  # noqa: E402
import oracle.sparse as OS  # noqa: E402

OUT = os.path.join(ROOT, "tools", "lab", "data")
os.makedirs(OUT, exist_ok=True)
for name, graph in (("chunglu", "chunglu"), ("local", "local-sorted")):
    d = D.arxiv_like(1.0, seed=0, with_teacher=False, graph=graph)
    rp, col, _ = d.adj_t.csr()
    g = OS.gcn_norm_sparse(OS.SparseTensor(rowptr=rp, col=col, sparse_sizes=d.adj_t.sparse_sizes()))
    rp, col, val = g.csr()
    comm_ptr = np.zeros(0, dtype=np.int32)
    if d.community is not None:
        c = d.community.numpy()
        assert (np.diff(c) >= 0).all()
        comm_ptr = np.concatenate([[0], np.nonzero(np.diff(c))[0] + 1, [c.size]]).astype(np.int32)
    with open(os.path.join(OUT, name + ".bin"), "wb") as f:
        np.array([d.num_nodes, col.numel(), max(comm_ptr.size - 1, 0)], dtype=np.int64).tofile(f)
        rp.numpy().astype(np.int32).tofile(f)
        col.numpy().astype(np.int32).tofile(f)
        val.numpy().astype(np.float32).tofile(f)
        comm_ptr.tofile(f)
    print(name, "n", d.num_nodes, "nnz", col.numel(), "communities", max(comm_ptr.size - 1, 0))
