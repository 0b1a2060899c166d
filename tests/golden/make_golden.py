#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON FILES.

Run only in the build container (needs /root/reference; the GPU box has no copy):

    python tests/golden/make_golden.py

What is executed for real (imported from /root/reference, unmodified):
  * arxiv_pyg/criterion.py   -> kd / fitnet / at / gpw / lpw / nce criteria
  * ppi_pyg/criterion.py     -> multi-label kd_criterion (criterion.npz) and the BCE-flavoured fitnet / at / gpw / lpw / nce
                                criteria (criterion_ppi.npz)
  * arxiv_pyg/gnn.py         -> GCN, SAGE, ProjectionGCD, train(), test()
  * arxiv_pyg/gnn_kd_and_aux.py -> train() (KD + aux combination rule)
  * ppi_pyg/gnn.py           -> GAT and TeacherNet (the frozen teacher the PPI student step runs, :208-209), GCN,
                                train() (kd with the teacher forward inside every step, supervised) and test() (micro-F1)
  * mag_pyg/gnn.py           -> RGCNConv (a MessagePassing subclass of the reference's own), RGCN.forward / .inference
  * arxiv_dgl/models.py      -> GATConv / GAT (the arxiv GAT teacher: eval-mode forward that produces the features/ and logits/
                                artefacts; ``dgl`` message-passing built-ins shimmed from the DGL docs), with the label-reuse loop
                                of arxiv_dgl/gat.py:151-166 restated inline (gat.py imports matplotlib / ogb at module top)

What is shimmed (third-party packages that are neither vendored by the reference nor installable
here -- SURVEY.md 8c): ``torch_geometric`` (GCNConv, SAGEConv, utils.softmax, utils.subgraph,
transforms.ToSparseTensor -> the oracle restatements), ``ogb`` (Evaluator -> SURVEY 9.10),
``torch.utils.tensorboard`` (unused by the functions called).  The goldens therefore PIN the
reference's own code (loss formulas, model structure, train/eval step) and leave the PyG operator
semantics pinned only by tests/test_oracle_known_answers.py.
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True   # importing from the read-only /root/reference must not leave __pycache__ there
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import oracle.nn as onn  # noqa: E402
import oracle.sparse as osp  # noqa: E402
import oracle.utils as outils  # noqa: E402


class _Evaluator:
    def __init__(self, name=None):
        self.name = name

    def eval(self, d):
        yt, yp = d["y_true"].cpu().numpy(), d["y_pred"].cpu().numpy()
        return {"acc": float((yt == yp).mean())}


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    utils = mod("torch_geometric.utils", softmax=outils.softmax, subgraph=outils.subgraph,
                to_dense_adj=None, negative_sampling=None, add_self_loops=None, to_undirected=None)
    hetero = mod("torch_geometric.utils.hetero", group_hetero_graph=None)
    utils.hetero = hetero
    nn_ = mod("torch_geometric.nn", GCNConv=onn.GCNConv, SAGEConv=onn.SAGEConv, GATConv=onn.GATConv,
              MessagePassing=onn.MessagePassing)
    mod("torch_geometric.datasets", PPI=None)
    mod("torch_geometric.data", DataLoader=None, Data=None, GraphSAINTRandomWalkSampler=None)
    tr = mod("torch_geometric.transforms", ToSparseTensor=osp.ToSparseTensor)
    mod("torch_geometric", utils=utils, nn=nn_, transforms=tr)
    mod("torch_sparse", SparseTensor=osp.SparseTensor)
    npp = mod("ogb.nodeproppred", PygNodePropPredDataset=None, Evaluator=_Evaluator)
    mod("ogb", nodeproppred=npp)
    mod("torch.utils.tensorboard", SummaryWriter=object)


def load_ref(relpath, name):
    path = os.path.join(REF, relpath)
    d = os.path.dirname(path)
    sys.path.insert(0, d)  # `from logger import Logger`, `from criterion import *`
    try:
        for stale in ("criterion", "logger"):
            sys.modules.pop(stale, None)
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.path.remove(d)
    return m


def t2n(t):
    return t.detach().cpu().numpy().copy()  # copy: parameters are later updated in place


# ------------------------------------------------------------------------------------------------
def criterion_inputs(seed=0, n=48, C=7, P=12, Ds=10, Dt=17, E=160):
    g = torch.Generator().manual_seed(seed)
    d = dict(
        logits=torch.randn(n, C, generator=g) * 2,
        labels=torch.randint(0, C, (n,), generator=g),
        teacher_logits=torch.randn(n, C, generator=g) * 3,
        feat_p=torch.randn(n, P, generator=g),
        tfeat_p=torch.relu(torch.randn(n, P, generator=g)) + 0.01,
        feat=torch.relu(torch.randn(n, Ds, generator=g)) * 0.7,
        tfeat=torch.relu(torch.randn(n, Dt, generator=g)) * 0.5,
    )
    # directed edge list over the n nodes, CSR (row-major) order like adj_t.coo(); some nodes get
    # no incoming edge, a few get many
    src = torch.randint(0, n, (E,), generator=g)
    dst = torch.cat([torch.randint(0, n // 2, (E - 30,), generator=g), torch.full((30,), 3)])
    key = torch.unique(src * n + dst)
    d["edge_index"] = torch.stack([key // n, key % n])
    return d


def run_criterion_case(fn, tensors_req, *args, np_seed=None, **kw):
    """Call ``fn`` with grad-enabled clones; returns losses and grads of ``loss`` wrt inputs."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in tensors_req.items()}
    if np_seed is not None:
        np.random.seed(np_seed)
    out = fn(leaves, *args, **kw)
    loss, loss_cls, loss_aux = out
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True, retain_graph=True)
    aux_grads = torch.autograd.grad(loss_aux, list(leaves.values()), allow_unused=True)
    rec = {"loss": t2n(loss), "loss_cls": t2n(loss_cls), "loss_aux": t2n(loss_aux)}
    for (k, _), g, ga in zip(leaves.items(), grads, aux_grads):
        rec["grad_" + k] = t2n(g) if g is not None else np.zeros(0, np.float32)
        rec["auxgrad_" + k] = t2n(ga) if ga is not None else np.zeros(0, np.float32)
    return rec


def make_criterion_goldens():
    ref = load_ref("arxiv_pyg/criterion.py", "ref_arxiv_criterion")
    refp = load_ref("ppi_pyg/criterion.py", "ref_ppi_criterion")
    d = criterion_inputs()
    out = {"in_" + k: t2n(v) for k, v in d.items()}
    cases = {}

    cases["kd"] = run_criterion_case(
        lambda L: ref.kd_criterion(L["logits"], d["labels"], d["teacher_logits"], 0.9, 4.0),
        {"logits": d["logits"]})
    cases["kd_a05_T1"] = run_criterion_case(
        lambda L: ref.kd_criterion(L["logits"], d["labels"], d["teacher_logits"], 0.5, 1.0),
        {"logits": d["logits"]})
    cases["fitnet"] = run_criterion_case(
        lambda L: ref.fitnet_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], 1000),
        {"logits": d["logits"], "feat": d["feat_p"], "tfeat": d["tfeat_p"]})
    cases["at"] = run_criterion_case(
        lambda L: ref.at_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], 1000),
        {"logits": d["logits"], "feat": d["feat"], "tfeat": d["tfeat"]})
    for kern in ("cosine", "poly", "l2", "rbf"):
        # full (no subsampling) and subsampled (S=32 < n=48, np seed 123)
        for tag, S, seed in (("full", 8192, None), ("sub", 32, 123)):
            cases[f"gpw_{kern}_{tag}"] = run_criterion_case(
                lambda L: ref.gpw_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], kern, 2.0, S),
                {"logits": d["logits"], "feat": d["feat_p"], "tfeat": d["tfeat_p"]}, np_seed=seed)
        for crit in ("kld", "mse"):
            cases[f"lpw_{kern}_{crit}"] = run_criterion_case(
                lambda L: ref.lpw_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], d["edge_index"],
                                            kern, 100, crit),
                {"logits": d["logits"], "feat": d["feat"], "tfeat": d["tfeat"]})
    for tag, S, seed in (("full", 8192, None), ("sub", 32, 7)):
        cases[f"nce_{tag}"] = run_criterion_case(
            lambda L: ref.nce_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], 0.1, 0.075, S),
            {"logits": d["logits"], "feat": d["feat_p"], "tfeat": d["tfeat_p"]}, np_seed=seed)

    # PPI multi-label KD
    g = torch.Generator().manual_seed(5)
    pl = torch.randn(40, 11, generator=g)
    py = (torch.rand(40, 11, generator=g) < 0.3).float()
    pt = torch.randn(40, 11, generator=g) * 2
    out["in_ppi_logits"], out["in_ppi_labels"], out["in_ppi_teacher"] = t2n(pl), t2n(py), t2n(pt)
    cases["ppi_kd"] = run_criterion_case(lambda L: refp.kd_criterion(L["logits"], py, pt, 0.5, 1.0),
                                         {"logits": pl})
    for name, rec in cases.items():
        for k, v in rec.items():
            out[f"{name}__{k}"] = v
    np.savez_compressed(os.path.join(HERE, "criterion.npz"), **out)
    print("criterion.npz:", len(cases), "cases")


def ppi_criterion_cases(ref_module, d, py):
    """(name -> callable(leaves)) of the PPI auxiliary criteria on the shared toy inputs; ``ref_module`` = the reference's
    ppi_pyg/criterion.py (generator) or a module with the same six names (tests)."""
    cases = {
        "ppi_fitnet": (lambda L: ref_module.fitnet_criterion(L["logits"], py, L["feat"], L["tfeat"], 1000), "p", None),
        "ppi_at": (lambda L: ref_module.at_criterion(L["logits"], py, L["feat"], L["tfeat"], 1000), "raw", None),
        "ppi_nce_full": (lambda L: ref_module.nce_criterion(L["logits"], py, L["feat"], L["tfeat"], 0.1, 0.075, 8192), "p", None),
        "ppi_nce_sub": (lambda L: ref_module.nce_criterion(L["logits"], py, L["feat"], L["tfeat"], 0.1, 0.075, 32), "p", 7),
    }
    for kern in ("cosine", "rbf"):
        cases[f"ppi_gpw_{kern}_sub"] = (lambda L, kern=kern: ref_module.gpw_criterion(L["logits"], py, L["feat"], L["tfeat"], kern, 2.0, 32), "p", 123)
        cases[f"ppi_lpw_{kern}_kld"] = (lambda L, kern=kern: ref_module.lpw_criterion(L["logits"], py, L["feat"], L["tfeat"], d["edge_index"],
                                                                                       kern, 100, "kld"), "raw", None)
    return cases


def make_ppi_criterion_goldens():
    """The auxiliary criteria of the reference's ppi_pyg/criterion.py:21-146 (multi-label BCE classification term) -- a file of
    its own, so that criterion.npz keeps regenerating bit for bit."""
    refp = load_ref("ppi_pyg/criterion.py", "ref_ppi_criterion_aux")
    d = criterion_inputs()
    g = torch.Generator().manual_seed(6)
    n, C = d["logits"].shape
    py = (torch.rand(n, C, generator=g) < 0.3).float()
    out = {"in_ppi_labels": t2n(py)}
    n_cases = 0
    for name, (fn, which, seed) in ppi_criterion_cases(refp, d, py).items():
        leaves = {"logits": d["logits"], "feat": d["feat_p"] if which == "p" else d["feat"], "tfeat": d["tfeat_p"] if which == "p" else d["tfeat"]}
        rec = run_criterion_case(fn, leaves, np_seed=seed)
        for k, v in rec.items():
            out[f"{name}__{k}"] = v
        n_cases += 1
    np.savez_compressed(os.path.join(HERE, "criterion_ppi.npz"), **out)
    print("criterion_ppi.npz:", n_cases, "cases")


# ------------------------------------------------------------------------------------------------
def tiny_graph(seed=1, n=72, F_in=8, C=5, Dt=20, E=260):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (E,), generator=g)
    dst = torch.cat([torch.randint(0, n, (E - 40,), generator=g), torch.full((40,), 5)])
    keep = src != dst
    edge_index = torch.stack([src[keep], dst[keep]])
    perm = torch.randperm(n, generator=g)
    return dict(
        edge_index=edge_index, x=torch.randn(n, F_in, generator=g),
        y=torch.randint(0, C, (n, 1), generator=g),
        train_idx=perm[:40].clone(), valid_idx=perm[40:55].clone(), test_idx=perm[55:].clone(),
        teacher_out_feat=torch.relu(torch.randn(n, Dt, generator=g)),
        teacher_logits=torch.randn(n, C, generator=g) * 3,
    )


def make_train_goldens():
    ref = load_ref("arxiv_pyg/gnn.py", "ref_arxiv_gnn")
    ref2 = load_ref("arxiv_pyg/gnn_kd_and_aux.py", "ref_arxiv_gnn_kd_and_aux")
    G = tiny_graph()
    n, F_in = G["x"].shape
    C, Dt, H, P, L = 5, G["teacher_out_feat"].shape[1], 16, 12, 3
    out = {"in_" + k: t2n(v) for k, v in G.items()}
    out["hp"] = np.array([H, P, L, C], dtype=np.int64)

    data = types.SimpleNamespace(x=G["x"], y=G["y"], edge_index=G["edge_index"], num_nodes=n)
    data = osp.ToSparseTensor()(data)
    data.adj_t = data.adj_t.to_symmetric()
    rowptr, col, _ = data.adj_t.csr()
    out["adj_rowptr"], out["adj_col"] = t2n(rowptr), t2n(col)
    split_idx = {"train": G["train_idx"], "valid": G["valid_idx"], "test": G["test_idx"]}
    ei = torch.stack(data.adj_t.coo()[:2])
    edge_index_tr = outils.subgraph(G["train_idx"], ei, relabel_nodes=True)[0]
    out["train_subgraph_edge_index"] = t2n(edge_index_tr)

    def args_for(mode, **kw):
        a = dict(training=mode, alpha=0.9, kd_T=4.0, beta=0.5, nce_T=0.075, max_samples=24, proj_dim=P, kernel="rbf")
        a.update(kw)
        return argparse.Namespace(**a)

    runs = [
        ("gcn", "supervised", {}, ref), ("gcn", "kd", {}, ref),
        ("gcn", "nce", dict(beta=0.1), ref), ("gcn", "nce", dict(beta=0.1, max_samples=8192), ref),
        ("gcn", "gpw", dict(kernel="cosine", beta=100.0), ref), ("gcn", "gpw", dict(kernel="rbf", beta=10.0), ref),
        ("gcn", "fitnet", dict(beta=10.0), ref), ("gcn", "at", dict(beta=10.0), ref),
        ("gcn", "gcd", dict(beta=0.1), ref),
        ("sage", "lpw", dict(kernel="rbf", beta=100.0), ref), ("sage", "lpw", dict(kernel="cosine", beta=100.0), ref),
        ("sage", "kd", {}, ref), ("sage", "nce", dict(beta=0.1), ref),
        ("gcn", "nce", dict(beta=0.01, nce_T=0.05), ref2), ("sage", "lpw", dict(kernel="l2", beta=1.0), ref2),
    ]
    names = []
    for i, (gnn, mode, kw, R) in enumerate(runs):
        tag = f"run{i:02d}"
        names.append(f"{tag}:{gnn}:{mode}:{'kdaux' if R is ref2 else 'aux'}:" +
                     ",".join(f"{k}={v}" for k, v in sorted(kw.items())))
        a = args_for(mode, **kw)
        torch.manual_seed(100 + i)
        np.random.seed(100 + i)
        Model = R.GCN if gnn == "gcn" else R.SAGE
        model = Model(F_in, H, C, L, 0.0)  # dropout 0: no device-RNG coupling in the goldens
        sp = tp = None
        if mode in ("nce", "gpw", "fitnet"):
            sp = torch.nn.Sequential(torch.nn.Linear(H, P), torch.nn.BatchNorm1d(P), torch.nn.ReLU())
            tp = torch.nn.Sequential(torch.nn.Linear(Dt, P), torch.nn.BatchNorm1d(P), torch.nn.ReLU())
        elif mode == "gcd":
            sp, tp = R.ProjectionGCD(H, P), R.ProjectionGCD(Dt, P)
        params = [{"params": model.parameters(), "lr": 0.01}]
        if sp is not None:
            params += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
        opt = torch.optim.Adam(params)
        for k, v in model.state_dict().items():
            out[f"{tag}__init__model.{k}"] = t2n(v)
        if sp is not None:
            for k, v in sp.state_dict().items():
                out[f"{tag}__init__sproj.{k}"] = t2n(v)
            for k, v in tp.state_dict().items():
                out[f"{tag}__init__tproj.{k}"] = t2n(v)
        logits0, accs0 = R.test(model, data, split_idx, _Evaluator())  # eval at the initial state
        out[f"{tag}__eval0_logits"] = t2n(logits0)
        out[f"{tag}__eval0_accs"] = np.array(accs0, dtype=np.float64)
        losses = []
        for _ in range(3):
            losses.append(R.train(model, data, G["train_idx"], opt, a, G["teacher_out_feat"], G["teacher_logits"],
                                  sp, tp, edge_index_tr if mode == "lpw" else None))
        # NOTE: biases that feed a BatchNorm have a mathematically zero gradient; under Adam their
        # update is rounding noise / (|noise| + eps), so they (and running_mean, and therefore the
        # post-training eval logits) are not reproducible across implementations.  Tests compare
        # eval0_* tightly and exclude those entries from the final-state comparison.
        logits, accs = R.test(model, data, split_idx, _Evaluator())
        out[f"{tag}__losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}__eval_logits"] = t2n(logits)
        out[f"{tag}__eval_accs"] = np.array(accs, dtype=np.float64)
        for k, v in model.state_dict().items():
            out[f"{tag}__final__model.{k}"] = t2n(v)
    out["run_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "train_arxiv.npz"), **out)
    print("train_arxiv.npz:", len(runs), "runs")


def make_train_dropout_goldens():
    """The training regime the benchmark times: the reference's own GCN / SAGE + train() with dropout 0.5.  The masks come from
    torch's CPU generator inside the reference's F.dropout calls (gnn.py:50,83); `torch.manual_seed(step_seed)` right before the
    first step fixes them, and the oracle -- which calls F.dropout in the same order on the same shapes -- draws the SAME masks
    from the same seed.  Pins the oracle's dropout-mode trajectory (oracle/training_parity.py injects the HIP path's masks into it)."""
    ref = load_ref("arxiv_pyg/gnn.py", "ref_arxiv_gnn")
    G = tiny_graph(seed=3, n=96, E=420)
    n, F_in = G["x"].shape
    C, Dt, H, P, L = 5, G["teacher_out_feat"].shape[1], 16, 12, 3
    out = {"in_" + k: t2n(v) for k, v in G.items()}
    out["hp"] = np.array([H, P, L, C], dtype=np.int64)
    data = types.SimpleNamespace(x=G["x"], y=G["y"], edge_index=G["edge_index"], num_nodes=n)
    data = osp.ToSparseTensor()(data)
    data.adj_t = data.adj_t.to_symmetric()
    rowptr, col, _ = data.adj_t.csr()
    out["adj_rowptr"], out["adj_col"] = t2n(rowptr), t2n(col)
    ei = torch.stack(data.adj_t.coo()[:2])
    edge_index_tr = outils.subgraph(G["train_idx"], ei, relabel_nodes=True)[0]
    out["train_subgraph_edge_index"] = t2n(edge_index_tr)
    runs = [("gcn", "nce", dict(beta=0.1)), ("gcn", "kd", {}), ("gcn", "gpw", dict(kernel="cosine", beta=100.0)),
            ("sage", "lpw", dict(kernel="cosine", beta=100.0)), ("sage", "nce", dict(beta=0.1))]
    names = []
    for i, (gnn, mode, kw) in enumerate(runs):
        tag = f"run{i:02d}"
        names.append(f"{tag}:{gnn}:{mode}:aux:" + ",".join(f"{k}={v}" for k, v in sorted(kw.items())))
        a = dict(training=mode, alpha=0.9, kd_T=4.0, beta=0.5, nce_T=0.075, max_samples=24, proj_dim=P, kernel="rbf")
        a.update(kw)
        a = argparse.Namespace(**a)
        torch.manual_seed(300 + i)
        Model = ref.GCN if gnn == "gcn" else ref.SAGE
        model = Model(F_in, H, C, L, 0.5)
        sp = tp = None
        if mode in ("nce", "gpw"):
            sp = torch.nn.Sequential(torch.nn.Linear(H, P), torch.nn.BatchNorm1d(P), torch.nn.ReLU())
            tp = torch.nn.Sequential(torch.nn.Linear(Dt, P), torch.nn.BatchNorm1d(P), torch.nn.ReLU())
        params = [{"params": model.parameters(), "lr": 0.01}]
        if sp is not None:
            params += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
        opt = torch.optim.Adam(params)
        for k, v in model.state_dict().items():
            out[f"{tag}__init__model.{k}"] = t2n(v)
        if sp is not None:
            for k, v in sp.state_dict().items():
                out[f"{tag}__init__sproj.{k}"] = t2n(v)
            for k, v in tp.state_dict().items():
                out[f"{tag}__init__tproj.{k}"] = t2n(v)
        torch.manual_seed(700 + i)            # the step seed: fixes the dropout masks of the 4 steps below
        np.random.seed(700 + i)
        losses = [ref.train(model, data, G["train_idx"], opt, a, G["teacher_out_feat"], G["teacher_logits"], sp, tp,
                            edge_index_tr if mode == "lpw" else None) for _ in range(4)]
        out[f"{tag}__losses"] = np.array(losses, dtype=np.float64)
        for k, v in model.state_dict().items():
            out[f"{tag}__final__model.{k}"] = t2n(v)
    out["run_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "train_arxiv_dropout.npz"), **out)
    print("train_arxiv_dropout.npz:", len(runs), "runs")


def make_ppi_teacher_goldens():
    """The reference's own GAT / TeacherNet bodies (skip connections, ELU, head averaging, out_feat) in eval mode on a
    small multi-graph-free input; GATConv itself is the oracle restatement (shim)."""
    ref = load_ref("ppi_pyg/gnn.py", "ref_ppi_gnn")
    g = torch.Generator().manual_seed(11)
    n, F_in, C = 53, 9, 7
    src = torch.randint(0, n - 3, (300,), generator=g)      # the last three nodes only have their self loops
    dst = torch.cat([torch.randint(0, n - 3, (260,), generator=g), torch.full((40,), 2)])
    src[:15] = dst[:15]                                      # self loops in the input (GATConv replaces them)
    edge_index = torch.stack([src, dst])
    x = torch.randn(n, F_in, generator=g)
    out = {"in_x": t2n(x), "in_edge_index": t2n(edge_index)}
    torch.manual_seed(3)
    m = ref.GAT(F_in, 6, C, 3, 0.5, heads=2)
    m.eval()
    with torch.no_grad():
        y = m(x, edge_index)
    for k, v in m.state_dict().items():
        out["gat_param__" + k] = t2n(v)
    out["gat_logits"], out["gat_out_feat"] = t2n(y), t2n(m.out_feat)
    # TeacherNet is hard-wired to 4 x 256 hidden units (2.3 M parameters): its weights are rebuilt from the seed by the
    # tests (same module construction order), with a few checksums to detect a different RNG stream
    torch.manual_seed(4)
    t = ref.TeacherNet(F_in, C)
    t.eval()
    with torch.no_grad():
        yt = t(x, edge_index)
    out["teachernet_logits"], out["teachernet_out_feat_sum"] = t2n(yt), t2n(t.out_feat.double().sum())
    out["teachernet_param_sums"] = np.array([float(v.double().sum()) for v in t.state_dict().values()])
    out["teachernet_keys"] = np.array(list(t.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "ppi_teacher.npz"), **out)
    print("ppi_teacher.npz: GAT + TeacherNet forward")


class _Batch:
    """What the PPI loops need from a PyG batch: x / edge_index / y and .to(device)."""

    def __init__(self, x, edge_index, y):
        self.x, self.edge_index, self.y = x, edge_index, y

    def to(self, device):
        return self


def ppi_toy(seed=31, n_graphs=3, F_in=9, C=7):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n_graphs):
        n = 40 + 11 * i
        src = torch.randint(0, n, (6 * n,), generator=g)
        dst = torch.randint(0, n, (6 * n,), generator=g)
        key = torch.unique(torch.cat([src * n + dst, dst * n + src]))       # symmetric, like PPI
        out.append(_Batch(torch.randn(n, F_in, generator=g), torch.stack([key // n, key % n]),
                          (torch.rand(n, C, generator=g) < 0.3).float()))
    return out


def make_ppi_train_goldens():
    """The reference's own ppi_pyg train() / test() loops (kd with the teacher forward inside each step; supervised)."""
    ref = load_ref("ppi_pyg/gnn.py", "ref_ppi_gnn_train")
    graphs = ppi_toy()
    F_in, C = graphs[0].x.shape[1], graphs[0].y.shape[1]
    out = {}
    for i, b in enumerate(graphs):
        out[f"in_x{i}"], out[f"in_ei{i}"], out[f"in_y{i}"] = t2n(b.x), t2n(b.edge_index), t2n(b.y)
    torch.manual_seed(5)
    teacher = ref.GAT(F_in, 6, C, 3, 0.0, heads=2)
    for k, v in teacher.state_dict().items():
        out["teacher__" + k] = t2n(v)
    for mode in ("kd", "supervised"):
        torch.manual_seed(6)
        model = ref.GCN(F_in, 16, C, 2, 0.0)
        for k, v in model.state_dict().items():
            out[f"{mode}_init__" + k] = t2n(v)
        opt = torch.optim.Adam(model.parameters(), lr=0.005)
        args = types.SimpleNamespace(training=mode, alpha=0.5, kd_T=1.0)
        recs = [ref.train(model, teacher if mode == "kd" else None, None, graphs, opt, args, "cpu") for _ in range(3)]
        out[f"{mode}_epoch_losses"] = np.array(recs, dtype=np.float64)
        out[f"{mode}_f1"] = np.array(ref.test(model, None, graphs, "cpu"), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ppi_train.npz"), **out)
    print("ppi_train.npz: kd + supervised epochs, micro-F1")


def mag_toy(seed=21):
    """Three node types (0 has features, 1 and 2 get embeddings), four relations incl. a reverse pair."""
    g = torch.Generator().manual_seed(seed)
    sizes = {0: 23, 1: 17, 2: 9}
    rels = {("a", "r0", "b"): (0, 1, 60), ("b", "r1", "a"): (1, 0, 60), ("a", "r2", "c"): (0, 2, 30), ("c", "r3", "a"): (2, 0, 45)}
    key2int = {"a": 0, "b": 1, "c": 2}
    edge_index_dict = {}
    for i, (k, (s, t, e)) in enumerate(rels.items()):
        row = torch.randint(0, sizes[s], (e,), generator=g)
        col = torch.randint(0, sizes[t], (e,), generator=g)
        row[-1], col[-1] = sizes[s] - 1, sizes[t] - 1          # the inferred SparseTensor sizes (mag:151) equal the true ones
        edge_index_dict[k] = (row, col)
        key2int[k] = i
    x0 = torch.randn(sizes[0], 8, generator=g)
    # grouped homogeneous form (what group_hetero_graph would give): offsets per node type
    off = {0: 0, 1: sizes[0], 2: sizes[0] + sizes[1]}
    ei = torch.cat([torch.stack([r + off[rels[k][0]], c + off[rels[k][1]]]) for k, (r, c) in edge_index_dict.items()], dim=1)
    et = torch.cat([torch.full((rels[k][2],), key2int[k]) for k in rels])
    node_type = torch.cat([torch.full((sizes[t],), t) for t in (0, 1, 2)])
    local_idx = torch.cat([torch.arange(sizes[t]) for t in (0, 1, 2)])
    return sizes, edge_index_dict, key2int, x0, ei, et, node_type, local_idx


def make_mag_rgcn_goldens():
    ref = load_ref("mag_pyg/gnn.py", "ref_mag_gnn")
    sizes, edge_index_dict, key2int, x0, ei, et, node_type, local_idx = mag_toy()
    torch.manual_seed(9)
    m = ref.RGCN(8, 12, 5, 2, 0.5, sizes, [0], 4)
    m.eval()
    out = {"in_x0": t2n(x0), "in_edge_index": t2n(ei), "in_edge_type": t2n(et), "in_node_type": t2n(node_type),
           "in_local_idx": t2n(local_idx)}
    for k, (r, c) in edge_index_dict.items():
        out["in_rel__" + "|".join(k)] = t2n(torch.stack([r, c]))
    for k, v in m.state_dict().items():
        out["param__" + k] = t2n(v)
    with torch.no_grad():
        y = m({0: x0}, ei, et, node_type, local_idx)
        inf = m.inference({0: x0}, edge_index_dict, key2int)
    out["forward_logits"], out["forward_out_feat"] = t2n(y), t2n(m.out_feat)
    for j, v in inf.items():
        out[f"inference__{j}"] = t2n(v)
    np.savez_compressed(os.path.join(HERE, "mag_rgcn.npz"), **out)
    print("mag_rgcn.npz: RGCN forward + inference")


# ------------------------------------------------------------------------------------------------
# arxiv_dgl: the GAT teacher that produces the student's input artefacts
class _DGLGraph:
    """The slice of a homogeneous ``dgl.DGLGraph`` that arxiv_dgl/models.py touches, restated from the DGL 0.5/0.6 docs
    (third party, not installable here): edges (src[e], dst[e]); ``srcdata`` / ``dstdata`` / ``ndata`` are ONE frame on a
    non-block graph; ``apply_edges(f)`` / ``update_all(msg, reduce)`` run the built-in functions below."""
    is_block = False

    def __init__(self, src, dst, n):
        self.src, self.dst, self.n = src, dst, n
        self.ndata = {}
        self.srcdata = self.dstdata = self.ndata
        self.edata = {}

    def local_scope(self):
        import contextlib

        @contextlib.contextmanager
        def scope():
            nd, ed = dict(self.ndata), dict(self.edata)
            try:
                yield
            finally:
                self.ndata.clear(), self.ndata.update(nd)
                self.edata.clear(), self.edata.update(ed)
        return scope()

    def in_degrees(self):
        return torch.bincount(self.dst, minlength=self.n)

    def out_degrees(self):
        return torch.bincount(self.src, minlength=self.n)

    def number_of_edges(self):
        return self.src.numel()

    def number_of_dst_nodes(self):
        return self.n

    def apply_edges(self, f):
        f(self)

    def update_all(self, msg, red):
        msg(self)
        red(self)


def install_dgl_shim():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def u_add_v(a, b, out):
        return lambda g: g.edata.__setitem__(out, g.srcdata[a][g.src] + g.dstdata[b][g.dst])

    def copy_u(a, out):
        return lambda g: g.edata.__setitem__(out, g.srcdata[a][g.src])

    def u_mul_e(a, b, out):
        return lambda g: g.edata.__setitem__(out, g.srcdata[a][g.src] * g.edata[b])

    def fsum(m, out):
        def red(g):
            e = g.edata[m]
            g.dstdata[out] = torch.zeros((g.n,) + tuple(e.shape[1:]), dtype=e.dtype).index_add_(0, g.dst, e)
        return red

    def edge_softmax(graph, e, eids=None):
        dst = graph.dst if eids is None else graph.dst[eids]
        shape = (graph.n,) + tuple(e.shape[1:])
        idx = dst.view(-1, *([1] * (e.dim() - 1))).expand_as(e)
        mx = torch.full(shape, float("-inf"), dtype=e.dtype).scatter_reduce(0, idx, e, "amax", include_self=True)
        ex = torch.exp(e - mx[dst])
        return ex / torch.zeros(shape, dtype=e.dtype).index_add_(0, dst, ex)[dst]

    fn = mod("dgl.function", u_add_v=u_add_v, copy_u=copy_u, u_mul_e=u_mul_e, sum=fsum)
    base = mod("dgl._ffi.base", DGLError=RuntimeError)
    ffi = mod("dgl._ffi", base=base)
    putils = mod("dgl.nn.pytorch.utils", Identity=torch.nn.Identity)
    pyt = mod("dgl.nn.pytorch", utils=putils, GraphConv=None)
    dnn = mod("dgl.nn", pytorch=pyt)
    dops = mod("dgl.ops", edge_softmax=edge_softmax)
    dut = mod("dgl.utils", expand_as_pair=lambda x: x if isinstance(x, tuple) else (x, x))
    mod("dgl", function=fn, _ffi=ffi, nn=dnn, ops=dops, utils=dut)


def make_arxiv_gat_goldens():
    """arxiv_dgl/models.py's own ``GAT`` / ``GATConv`` (eval-mode forward: the producer of features/ and logits/ artefacts)
    in the three configurations of gat.py:375-389 (use_norm, no_attn_dst, label reuse) on a toy graph, plus the label-reuse
    evaluation loop of gat.py:151-183 (restated inline: gat.py itself imports matplotlib / ogb datasets at module top)."""
    install_dgl_shim()
    models = load_ref("arxiv_dgl/models.py", "ref_arxiv_dgl_models")
    g = torch.Generator().manual_seed(41)
    n, F_in, C, hidden, heads = 57, 11, 6, 5, 3
    # bidirected edges without duplicates, self loops replaced (gat.py:56-71)
    a = torch.rand(n, n, generator=g) < 0.08
    a = (a | a.t())
    a.fill_diagonal_(True)
    dst, src = torch.nonzero(a, as_tuple=True)        # row-major: edges grouped by destination, sources ascending
    graph = _DGLGraph(src, dst, n)
    x = torch.randn(n, F_in, generator=g)
    labels = torch.randint(0, C, (n, 1), generator=g)
    perm = torch.randperm(n, generator=g)
    train_idx, val_idx, test_idx = perm[:30], perm[30:42], perm[42:]
    out = {"in_src": t2n(src), "in_dst": t2n(dst), "in_x": t2n(x), "in_labels": t2n(labels), "in_train": t2n(train_idx),
           "in_val": t2n(val_idx), "in_test": t2n(test_idx)}
    for name, cfg in (("norm_noattn", dict(use_attn_dst=False, use_symmetric_norm=True, n_label_iters=1)),
                      ("plain", dict(use_attn_dst=True, use_symmetric_norm=False, n_label_iters=0)),
                      ("norm_attn", dict(use_attn_dst=True, use_symmetric_norm=True, n_label_iters=2))):
        torch.manual_seed(7)
        model = models.GAT(F_in + C, C, hidden, 3, heads, F.relu, dropout=0.75, input_drop=0.25, attn_drop=0.0, edge_drop=0.3,
                           use_attn_dst=cfg["use_attn_dst"], use_symmetric_norm=cfg["use_symmetric_norm"])
        with torch.no_grad():   # non-trivial BatchNorm running statistics and last-layer bias
            for bn in model.norms:
                bn.running_mean.normal_(0, 0.3, generator=g)
                bn.running_var.uniform_(0.5, 1.5, generator=g)
                bn.weight.uniform_(0.5, 1.5, generator=g)
                bn.bias.normal_(0, 0.2, generator=g)
            model.bias_last.bias.normal_(0, 0.5, generator=g)
        model.eval()
        for k, v in model.state_dict().items():
            out[f"{name}__param__{k}"] = t2n(v)
        with torch.no_grad():
            # gat.py:104-107 add_labels + :151-168 evaluate(), with the reference's model object
            onehot = torch.zeros([n, C])
            onehot[train_idx, labels[train_idx, 0]] = 1
            feat = torch.cat([x, onehot], dim=-1)
            pred = model(graph, feat)
            unlabel_idx = torch.cat([val_idx, test_idx])
            for _ in range(cfg["n_label_iters"]):
                feat[unlabel_idx, -C:] = F.softmax(pred[unlabel_idx], dim=-1)
                pred = model(graph, feat)
            conv0 = model.convs[0](graph, feat)
        out[f"{name}__pred"], out[f"{name}__feat"], out[f"{name}__conv0"] = t2n(pred), t2n(model.feat), t2n(conv0)
    np.savez_compressed(os.path.join(HERE, "arxiv_gat.npz"), **out)
    print("arxiv_gat.npz: arxiv_dgl GAT teacher forward x 3 configs (+ label reuse)")


if __name__ == "__main__":
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    torch.set_num_threads(1)  # reproducible reduction order in the recorded numbers
    install_shims()
    make_criterion_goldens()
    make_train_goldens()
    make_train_dropout_goldens()
    make_ppi_teacher_goldens()
    make_mag_rgcn_goldens()
    make_ppi_train_goldens()
    make_arxiv_gat_goldens()
    make_ppi_criterion_goldens()
