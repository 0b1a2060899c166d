"""TEST INFRASTRUCTURE: training-mode parity of the HIP path against the CPU oracle with the dropout masks fixed.

The product's fused BatchNorm kernels draw dropout masks from a counter hash (csrc/bn_common.h); the reference / oracle use
``F.dropout`` (arxiv_pyg/gnn.py:48-50,80-83).  This harness records the seeds of the GPU run, rebuilds the masks on the host
(oracle/dropout.py) and injects them into the oracle's ``F.dropout``, so that MULTI-STEP trajectories with dropout > 0 --
the regime the benchmark times -- are compared value by value (eager launches and hipGraph replays alike).
Only tests/ and bench.py's parity leg import this module.
"""
from __future__ import annotations

import contextlib
import copy
import types

import numpy as np
import torch

from . import models as OM
from . import sparse as OS
from . import utils as OU
from .dropout import counter_mask, injected_dropout

MASK64 = 0xFFFFFFFFFFFFFFFF


@contextlib.contextmanager
def recorded_seeds(ops):
    """Every per-call dropout seed ``ops._draw_dropout_seed`` hands out inside the block, in call order."""
    seeds: list = []
    orig = ops._draw_dropout_seed

    def draw():
        s = orig()
        seeds.append(s)
        return s
    ops._draw_dropout_seed = draw
    try:
        yield seeds
    finally:
        ops._draw_dropout_seed = orig


def oracle_data(data):
    """The problem on the oracle's own SparseTensor (CPU)."""
    rowptr, col, _ = data.adj_t.csr()
    dc = types.SimpleNamespace(**vars(data))
    dc.adj_t = OS.SparseTensor(rowptr=rowptr, col=col, sparse_sizes=data.adj_t.sparse_sizes())
    return dc


def build_pair(PM, data, d, device, gnn, mode, hp, hidden, layers, dropout, lr, seed):
    """Oracle and product models / heads / Adam with identical initial weights; ONE parameter group on both sides (the reference's
    three groups carry the same hyper-parameters, gnn.py:308-312), same parameter order."""
    torch.manual_seed(seed)
    def make(M, dev, fused):
        Net = M.GCN if gnn == "gcn" else M.SAGE
        model = Net(data.num_features, hidden, data.num_classes, layers, dropout).to(dev)
        sp = tp = None
        if mode in ("nce", "gpw", "fitnet"):
            sp = M.make_projection(hidden, hp["proj_dim"]).to(dev)
            tp = M.make_projection(data.teacher_out_feat.shape[1], hp["proj_dim"]).to(dev)
        params = [p for m in (model, sp, tp) if m is not None for p in m.parameters()]
        opt = torch.optim.Adam(params, lr=lr, **(dict(fused=True, capturable=True) if fused else {}))
        return model, sp, tp, opt
    om, osp, otp, oopt = make(OM, "cpu", False)
    pm, psp, ptp, popt = make(PM, device, True)
    pm.load_state_dict(om.state_dict())
    for a, b in ((psp, osp), (ptp, otp)):
        if a is not None:
            a.load_state_dict(b.state_dict())
    return (om, osp, otp, oopt), (pm, psp, ptp, popt)


def sync_oracle_to(product, oracle):
    """Weights, BatchNorm buffers and Adam state of the product side -> the oracle side (after a GraphedEpoch's warm-up steps)."""
    (pm, psp, ptp, popt), (om, osp, otp, oopt) = product, oracle
    for a, b in ((pm, om), (psp, osp), (ptp, otp)):
        if a is not None:
            b.load_state_dict({k: v.detach().cpu() for k, v in a.state_dict().items()})
    sd = popt.state_dict()
    state = {i: {k: (v.detach().cpu().to(torch.float32) if k == "step" else v.detach().cpu()) if torch.is_tensor(v) else v
                 for k, v in st.items()} for i, st in sd["state"].items()}
    groups = copy.deepcopy(oopt.state_dict()["param_groups"])
    oopt.load_state_dict({"state": state, "param_groups": groups})


def edges_of(data, d, mode, subgraph_fn):
    if mode != "lpw":
        return None, None
    eo = OU.subgraph(data.split_idx["train"], torch.stack(data.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=data.num_nodes)[0]
    ep = subgraph_fn(d.split_idx["train"], torch.stack(d.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
    return eo, ep


def oracle_steps(oracle, dc, mode, hp, edge_o, masks_by_step, numpy_seed):
    """The oracle's train steps with the given masks per step; one np.random seed before the first step."""
    om, osp, otp, oopt = oracle
    np.random.seed(numpy_seed)
    out = []
    for masks in masks_by_step:
        with injected_dropout(masks) as st:
            out.append(OM.train_step(om, dc.x, dc.adj_t, dc.y, dc.split_idx["train"], oopt, mode, hp, dc.teacher_out_feat,
                                     dc.teacher_logits, osp, otp, edge_o))
        if st.left() != 0:
            raise AssertionError(f"{st.left()} dropout masks of a step were not consumed by the oracle")
    return out


def masks_for(seeds, n, hidden, p):
    return [counter_mask(s & MASK64, n, hidden, p) for s in seeds]


def rel_errors(got, ref):
    """max over steps and the three loss terms of |got - ref| / max(|ref|, 1e-12); exact zeros must match exactly."""
    worst = 0.0
    for g, r in zip(got, ref):
        for a, b in zip(g, r):
            worst = max(worst, (abs(a - b) / abs(b)) if b != 0 else (0.0 if a == 0 else float("inf")))
    return worst


def trajectory(PM, ops, data, d, device, gnn, mode, hp, steps, graph, subgraph_fn, hidden=256, layers=3, dropout=0.5, lr=0.01,
               seed=0, warmup=2, sync_before_first_replay=True):
    """``steps`` optimisation steps of the product path (eager ``train_step`` or ``GraphedEpoch`` replays) and of the oracle with
    the product's dropout masks injected, from the same state, the same np.random draws.  Returns dict(got, ref, max_rel)."""
    oracle, product = build_pair(PM, data, d, device, gnn, mode, hp, hidden, layers, dropout, lr, seed)
    pm, psp, ptp, popt = product
    dc = oracle_data(data)
    edge_o, edge_p = edges_of(data, d, mode, subgraph_fn)
    n, n_drop = data.num_nodes, layers - 1
    np_seed = seed + 101
    got, seeds_by_step = [], []
    if not graph:
        np.random.seed(np_seed)
        for _ in range(steps):
            with recorded_seeds(ops) as seeds:
                got.append(PM.train_step(pm, d.x, d.adj_t, d.y, d.split_idx["train"], popt, mode, hp, d.teacher_out_feat,
                                         d.teacher_logits, psp, ptp, edge_p))
            assert len(seeds) == n_drop, (len(seeds), n_drop)
            seeds_by_step.append(list(seeds))
    else:
        with recorded_seeds(ops) as seeds:
            ge = PM.GraphedEpoch(pm, d.x, d.adj_t, d.y, d.split_idx["train"], popt, mode, hp, d.teacher_out_feat, d.teacher_logits,
                                 psp, ptp, edge_index=edge_p, split_idx=d.split_idx, warmup=warmup)
        assert len(seeds) == n_drop * (warmup + 1), (len(seeds), n_drop, warmup)
        host_seeds = seeds[-n_drop:]                     # the call-site seeds baked into the captured launches
        sync_oracle_to(product, oracle)                   # the warm-up steps moved weights, buffers and Adam state
        np.random.seed(np_seed)
        ge.redraw()
        if sync_before_first_replay:
            torch.cuda.synchronize()                      # the condition under which long torch reductions in a replay went stale
        for _ in range(steps):
            dev_seed = int(ge._seed_dev.item())           # the per-step device seed the NEXT replay adds to the baked ones
            losses, _ = ge.step()
            got.append(tuple(losses))
            seeds_by_step.append([(h + dev_seed) & MASK64 for h in host_seeds])
    ref = oracle_steps(oracle, dc, mode, hp, edge_o, [masks_for(s, n, hidden, dropout) for s in seeds_by_step], np_seed)
    return dict(got=[list(map(float, g)) for g in got], ref=[list(map(float, r)) for r in ref], max_rel=rel_errors(got, ref))
