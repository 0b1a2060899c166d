#!/usr/bin/env python3
"""Per-parameter gradient errors of ONE full-size optimisation step: GPU path vs the fp32 CPU oracle vs a float64 run of the same
oracle (the conditioning of every gradient tensor: how far two correct fp32 implementations may be apart).

    python tools/checks/grad_parity_probe.py [--scale 1.0] [--configs gcn:kd,gcn:nce,sage:lpw:cosine] [--cpu-only]
"""
import argparse
import copy
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
import oracle.models as OM  # noqa: E402
import oracle.sparse as OS  # noqa: E402
import oracle.utils as OU  # noqa: E402


def run_oracle(model, sp, tp, dc, mode, hp, edge, dtype, seed):
    model, sp, tp = (copy.deepcopy(m).to(dtype) if m is not None else None for m in (model, sp, tp))
    groups = [{"params": model.parameters(), "lr": 0.01}]
    for p in (sp, tp):
        if p is not None:
            groups.append({"params": p.parameters(), "lr": 0.01})
    opt = torch.optim.Adam(groups)
    orig = OS.SparseTensor.fill_value
    if dtype == torch.float64:
        OS.SparseTensor.fill_value = lambda self, fill, dtype=torch.float64: orig(self, fill, dtype)
    try:
        np.random.seed(seed)
        x = dc.x.to(dtype)
        tf, tl = dc.teacher_out_feat.to(dtype), dc.teacher_logits.to(dtype)
        losses = OM.train_step(model, x, dc.adj_t, dc.y, dc.split_idx["train"], opt, mode, hp, tf, tl, sp, tp, edge)
    finally:
        OS.SparseTensor.fill_value = orig
    named = [(f"model.{k}", v) for k, v in model.named_parameters()]
    for tag, m in (("student_proj", sp), ("teacher_proj", tp)):
        if m is not None:
            named += [(f"{tag}.{k}", v) for k, v in m.named_parameters()]
    return losses, {k: v.grad.detach().double() for k, v in named if v.grad is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--configs", default="gcn:kd,gcn:nce,sage:lpw:cosine")
    ap.add_argument("--cpu-only", action="store_true")
    a = ap.parse_args()
    data = D.arxiv_like(a.scale, seed=0)
    rowptr, col, _ = data.adj_t.csr()
    dc = types.SimpleNamespace(**vars(data))
    dc.adj_t = OS.SparseTensor(rowptr=rowptr, col=col, sparse_sizes=data.adj_t.sparse_sizes())
    dev = None if a.cpu_only else torch.device("cuda", 0)
    d = None if a.cpu_only else bench.to_device(data, dev)
    for cfg in a.configs.split(","):
        parts = cfg.split(":")
        gnn, mode = parts[0], parts[1]
        hp = {**bench.HP, **bench.MODE_HP.get(mode, {})}
        if len(parts) > 2:
            hp["kernel"] = parts[2]
        if a.scale < 1.0:
            hp["max_samples"] = min(hp["max_samples"], 512)
        args = types.SimpleNamespace(gnn=gnn, training=mode, seed=0)
        bench.seed_all(17)
        om, osp, otp, _ = bench.build_problem(OM, dc, "cpu", args, hp, dropout=0.0)
        edge = None
        if mode == "lpw":
            edge = OU.subgraph(dc.split_idx["train"], torch.stack(dc.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=dc.num_nodes)[0]
        l32, g32 = run_oracle(om, osp, otp, dc, mode, hp, edge, torch.float32, 17)
        l64, g64 = run_oracle(om, osp, otp, dc, mode, hp, edge, torch.float64, 17)
        gg = None
        if not a.cpu_only:
            import efficient_gnns_amd.models as PM
            pm, psp, ptp, popt = bench.build_problem(PM, d, dev, args, hp, dropout=0.0)
            pm.load_state_dict(om.state_dict())
            for x, y in ((psp, osp), (ptp, otp)):
                if x is not None:
                    x.load_state_dict(y.state_dict())
            edge_p = None
            if mode == "lpw":
                from efficient_gnns_amd.utils import subgraph
                edge_p = subgraph(d.split_idx["train"], torch.stack(d.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
            np.random.seed(17)
            lg = PM.train_step(pm, d.x, d.adj_t, d.y, d.split_idx["train"], popt, mode, hp, d.teacher_out_feat, d.teacher_logits, psp, ptp, edge_p)
            named = [(f"model.{k}", v) for k, v in pm.named_parameters()]
            for tag, m in (("student_proj", psp), ("teacher_proj", ptp)):
                if m is not None:
                    named += [(f"{tag}.{k}", v) for k, v in m.named_parameters()]
            gg = {k: v.grad.detach().double().cpu() for k, v in named if v.grad is not None}
            print(json.dumps(dict(config=cfg, losses_gpu=lg, losses_f32=l32, losses_f64=l64)))
        else:
            print(json.dumps(dict(config=cfg, losses_f32=l32, losses_f64=l64)))
        top = max(float(v.abs().max()) for v in g64.values())
        print(f"{'tensor':34s} {'max|g64|':>10s} {'rms|g64|':>10s} {'f32-f64 /max':>13s} {'gpu-f64 /max':>13s} {'gpu-f32 /max':>13s}   (top {top:.3e})")
        for k, r in g64.items():
            sc = float(r.abs().max()) + 1e-300
            e32 = float((g32[k] - r).abs().max()) / sc
            line = f"{k:34s} {sc:10.3e} {float(r.pow(2).mean().sqrt()):10.3e} {e32:13.3e}"
            if gg is not None:
                line += f" {float((gg[k] - r).abs().max()) / sc:13.3e} {float((gg[k] - g32[k]).abs().max()) / sc:13.3e}"
            print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
