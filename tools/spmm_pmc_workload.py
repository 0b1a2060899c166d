#!/usr/bin/env python3
"""Workload for PMC passes over the K=256 aggregation: (a) the headline Chung-Lu graph, (b) a regular graph whose
columns are confined to 4096 sources (every gather an L2 hit), (c) optional extra graphs given by EGNN_PMC_GRAPHS.
Three launches each, in that order, after a 256 MiB calibration copy (3 launches).  Graphs are cached under /tmp so
that the repeated rocprofv3 passes of one box session do not regenerate them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import efficient_gnns_amd as E  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
import efficient_gnns_amd.ops as ops  # noqa: E402

K = int(os.environ.get("EGNN_PMC_K", "256"))
REDUCE = os.environ.get("EGNN_PMC_REDUCE", "sum")
REPS = int(os.environ.get("EGNN_PMC_REPS", "3"))
which = os.environ.get("EGNN_PMC_GRAPHS", "chunglu,window4096").split(",")


def cached(name, build):
    path = f"/tmp/egnn_pmc_{name}.pt"
    if os.path.exists(path):
        return torch.load(path)
    obj = build()
    torch.save(obj, path)
    return obj


def graph(name):
    if name == "chunglu":
        def build():
            d = D.arxiv_like(1.0, seed=0, with_teacher=False)
            rp, col, _ = d.adj_t.csr()
            return rp, col, d.num_nodes
        rp, col, n = cached(name, build)
        return E.gcn_norm(E.SparseTensor(rowptr=rp.cuda(), col=col.cuda(), sparse_sizes=(n, n)))
    if name.startswith("window"):
        w = int(name[len("window"):])
        n, deg = 169343, 15
        g = torch.Generator(device="cuda").manual_seed(0)
        col, _ = torch.sort(torch.randint(0, w, (n, deg), device="cuda", generator=g), dim=1)
        return E.SparseTensor(rowptr=torch.arange(0, n * deg + 1, deg, device="cuda"), col=col.reshape(-1),
                              value=torch.rand(n * deg, device="cuda"), sparse_sizes=(n, n))
    if name.startswith("local"):
        def build():
            d = D.arxiv_like(1.0, seed=0, with_teacher=False, graph=name)
            rp, col, _ = d.adj_t.csr()
            return rp, col, d.num_nodes
        rp, col, n = cached(name, build)
        return E.gcn_norm(E.SparseTensor(rowptr=rp.cuda(), col=col.cuda(), sparse_sizes=(n, n)))
    if name in ("mag", "mag_reordered"):
        # BASELINE.json configs[4]: the MAG-shaped graph (N = 1 939 743, 42.2 M entries), value-less adjacency, mean aggregation
        def build():
            d = D.mag_like(1.0, seed=0)
            rp, col, _ = d.adj_t.csr()
            return rp, col, d.num_nodes
        rp, col, n = cached("mag", build)
        adj = E.SparseTensor(rowptr=rp.cuda(), col=col.cuda(), sparse_sizes=(n, n))
        if name == "mag_reordered":
            from efficient_gnns_amd.sparse import community_order
            adj = adj.permute(community_order(adj))
        return adj
    raise SystemExit(f"unknown graph {name}")


src = torch.randn(64 * 1024 * 1024, device="cuda")  # 256 MiB
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
for name in which:
    adj = graph(name)
    x = torch.randn(adj.sparse_size(1), K, device="cuda")
    ops.spmm_raw(adj, x, REDUCE)   # warm-up (plan construction)
    torch.cuda.synchronize()
    torch.zeros(7777, device="cuda")   # marker dispatch between graphs (grid of 7777 elements)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.spmm_raw(adj, x, REDUCE)
    e1.record()
    torch.cuda.synchronize()
    print("us_per_call", round(e0.elapsed_time(e1) * 1e3 / REPS, 1), end=" ")
    print("graph", name, "nnz", adj.nnz(), "alg_bytes", adj.spmm_algorithmic_bytes(K), flush=True)
