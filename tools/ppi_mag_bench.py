#!/usr/bin/env python3
"""Secondary BASELINE.json configs that bench.py does not time (tools only; numbers go to profiles/ and DESIGN.md):

  configs[0]  PPI, 2-layer GCN-256 student, logit KD, one optimisation step per graph (ppi_pyg/gnn.py:185-274) --
              timed WITH the frozen GAT teacher (TeacherNet) forward inside every step, as the reference does
              (:208-209), and WITHOUT it (teacher logits precomputed), on the GPU and with the CPU oracle;
  configs[4]  MAG-shaped SAGE-mean aggregation (mag_pyg/gnn.py:162): the mean-SpMM at N = 1.94 M / 42 M entries.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (cap_cpu_threads)
import efficient_gnns_amd as E  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
import efficient_gnns_amd.models as PM  # noqa: E402
import efficient_gnns_amd.ops as ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mag-scale", type=float, default=1.0)
ap.add_argument("--cpu-graphs", type=int, default=4, help="PPI graphs timed with the CPU oracle (0 = skip)")
args = ap.parse_args()
bench.cap_cpu_threads()
dev = torch.device("cuda", 0)


def sync():
    torch.cuda.synchronize()


# ---------------------------------------------------------------- PPI
train, _, _ = D.ppi_like(seed=0)
gpu_graphs = [(g.x.to(dev), g.edge_index.to(dev), g.y.to(dev), g.teacher_logits.to(dev)) for g in train]
torch.manual_seed(0)
student = PM.GCN(50, 256, 121, 2, 0.0, cached=False).to(dev)
teacher = PM.TeacherNet(50, 121).to(dev).eval().requires_grad_(False)
opt = torch.optim.Adam(student.parameters(), lr=0.005)


def ppi_epoch(with_teacher):
    student.train()
    tot = 0.0
    for x, ei, y, tl in gpu_graphs:
        if with_teacher:
            with torch.no_grad():
                tl = teacher(x, ei)
        loss, _, _ = E.ppi_kd_criterion(student(x, ei), y, tl, 0.5, 1.0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        tot += loss.item()                      # the reference accumulates loss.item() per batch (gnn.py:262-266)
    return tot


res = {}
for with_teacher in (False, True):
    for _ in range(2):
        ppi_epoch(with_teacher)
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ppi_epoch(with_teacher)
    sync()
    res["ppi_epoch_ms_with_teacher" if with_teacher else "ppi_epoch_ms_teacher_precomputed"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
x, ei, _, _ = gpu_graphs[0]
for _ in range(3):
    with torch.no_grad():
        teacher(x, ei)
sync()
t0 = time.perf_counter()
for _ in range(20):
    with torch.no_grad():
        teacher(x, ei)
sync()
res["teacher_forward_ms_graph0"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
res["graph0_nodes_edges"] = [int(x.shape[0]), int(ei.shape[1])]

if args.cpu_graphs > 0:
    import oracle.criterion as OC
    import oracle.models as OM
    torch.manual_seed(0)
    ostu, otea = OM.GCN(50, 256, 121, 2, 0.0, cached=False), OM.TeacherNet(50, 121).eval()
    oopt = torch.optim.Adam(ostu.parameters(), lr=0.005)
    sub = train[:args.cpu_graphs]
    for with_teacher in (False, True):
        t0 = time.perf_counter()
        for g in sub:
            tl = g.teacher_logits
            if with_teacher:
                with torch.no_grad():
                    tl = otea(g.x, g.edge_index)
            loss = OC.ppi_kd_criterion(ostu(g.x, g.edge_index), g.y, tl, 0.5, 1.0)[0]
            oopt.zero_grad()
            loss.backward()
            oopt.step()
        dt = time.perf_counter() - t0
        nodes = sum(g.num_nodes for g in sub)
        allnodes = sum(g.num_nodes for g in train)
        res["cpu_oracle_ppi_epoch_ms_" + ("with_teacher" if with_teacher else "teacher_precomputed") + "_extrapolated"] = round(
            dt * allnodes / nodes * 1e3, 1)
    res["cpu_threads"] = torch.get_num_threads()
    res["cpu_sample"] = f"{args.cpu_graphs} of 20 training graphs, scaled by node count"
print(json.dumps({"what": "ppi_2l_gcn_kd", **res}))

# ---------------------------------------------------------------- MAG-shaped mean aggregation
t0 = time.perf_counter()
d = D.mag_like(scale=args.mag_scale, seed=0)
gen_s = time.perf_counter() - t0
adj = d.adj_t.to(dev)
out = {"what": "mag_mean_spmm", "N": d.num_nodes, "nnz": adj.nnz(), "graph_generation_s": round(gen_s, 1)}
for K in (128, 256):
    xg = torch.randn(d.num_nodes, K, device=dev)
    for _ in range(2):
        ops.spmm_raw(adj, xg, "mean")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.spmm_raw(adj, xg, "mean")
    e1.record()
    sync()
    us = e0.elapsed_time(e1) / 5 * 1e3
    alg = adj.spmm_algorithmic_bytes(K)
    out[f"K{K}_us"] = round(us, 1)
    out[f"K{K}_alg_GBs"] = round(alg / us / 1e3, 1)
    out[f"K{K}_gather_GBs"] = round(adj.nnz() * K * 4 / us / 1e3, 1)
    del xg
print(json.dumps(out))
