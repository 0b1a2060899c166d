"""Oracle vs the golden vectors produced by the reference's own files (tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle.criterion as OC
import oracle.models as OM
import oracle.sparse as OS
import oracle.utils as OU
from conftest import ARXIV_GAT_CONFIGS, arxiv_gat_case, as_t, mag_rgcn_case, ppi_train_case

RT, AT = 1e-6, 1e-7


def _check(rec, G, name):
    for k, v in rec.items():
        ref = G[f"{name}__{k}"]
        np.testing.assert_allclose(v, ref, rtol=2e-5, atol=1e-7, err_msg=f"{name}:{k}")


def _run(fn, leaves, np_seed=None):
    L = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
    if np_seed is not None:
        np.random.seed(np_seed)
    loss, loss_cls, loss_aux = fn(L)
    g = torch.autograd.grad(loss, list(L.values()), allow_unused=True, retain_graph=True)
    ga = torch.autograd.grad(loss_aux, list(L.values()), allow_unused=True)
    rec = {"loss": loss.detach().numpy(), "loss_cls": loss_cls.detach().numpy(), "loss_aux": loss_aux.detach().numpy()}
    for (k, _), a, b in zip(L.items(), g, ga):
        rec["grad_" + k] = a.numpy() if a is not None else np.zeros(0, np.float32)
        rec["auxgrad_" + k] = b.numpy() if b is not None else np.zeros(0, np.float32)
    return rec


def criterion_cases(G, C=OC, dev="cpu"):
    """(name, fn(leaves)->triple, leaves, np_seed) for every golden criterion case."""
    d = {k[3:]: as_t(G[k], dev) for k in G.files if k.startswith("in_")}
    cases = [
        ("kd", lambda L: C.kd_criterion(L["logits"], d["labels"], d["teacher_logits"], 0.9, 4.0), {"logits": d["logits"]}, None),
        ("kd_a05_T1", lambda L: C.kd_criterion(L["logits"], d["labels"], d["teacher_logits"], 0.5, 1.0), {"logits": d["logits"]}, None),
        ("fitnet", lambda L: C.fitnet_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], 1000),
         {"logits": d["logits"], "feat": d["feat_p"], "tfeat": d["tfeat_p"]}, None),
        ("at", lambda L: C.at_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], 1000),
         {"logits": d["logits"], "feat": d["feat"], "tfeat": d["tfeat"]}, None),
    ]
    for kern in ("cosine", "poly", "l2", "rbf"):
        for tag, S, seed in (("full", 8192, None), ("sub", 32, 123)):
            cases.append((f"gpw_{kern}_{tag}",
                          lambda L, kern=kern, S=S: C.gpw_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], kern, 2.0, S),
                          {"logits": d["logits"], "feat": d["feat_p"], "tfeat": d["tfeat_p"]}, seed))
        for crit in ("kld", "mse"):
            cases.append((f"lpw_{kern}_{crit}",
                          lambda L, kern=kern, crit=crit: C.lpw_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"],
                                                                         d["edge_index"], kern, 100, crit),
                          {"logits": d["logits"], "feat": d["feat"], "tfeat": d["tfeat"]}, None))
    for tag, S, seed in (("full", 8192, None), ("sub", 32, 7)):
        cases.append((f"nce_{tag}", lambda L, S=S: C.nce_criterion(L["logits"], d["labels"], L["feat"], L["tfeat"], 0.1, 0.075, S),
                      {"logits": d["logits"], "feat": d["feat_p"], "tfeat": d["tfeat_p"]}, seed))
    return cases, d


def test_criteria_match_reference(golden_criterion):
    G = golden_criterion
    cases, _ = criterion_cases(G)
    assert len(cases) == 22
    for name, fn, leaves, seed in cases:
        _check(_run(fn, leaves, seed), G, name)


def ppi_aux_cases(G, Gp, mod, dev="cpu"):
    """(name, fn(leaves) -> triple, leaves, np_seed) of tests/golden/criterion_ppi.npz for a module with ppi_*_criterion functions."""
    import types
    d = {k[3:]: as_t(G[k], dev) for k in G.files if k.startswith("in_") and not k.startswith("in_ppi")}
    py = as_t(Gp["in_ppi_labels"], dev)
    names = types.SimpleNamespace(fitnet_criterion=mod.ppi_fitnet_criterion, at_criterion=mod.ppi_at_criterion,
                                  gpw_criterion=mod.ppi_gpw_criterion, lpw_criterion=mod.ppi_lpw_criterion, nce_criterion=mod.ppi_nce_criterion)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mk_golden_cases", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)          # only its case table is used (nothing under /root/reference is touched at import)
    out = []
    for name, (fn, which, seed) in mg.ppi_criterion_cases(names, d, py).items():
        leaves = {"logits": d["logits"], "feat": d["feat_p"] if which == "p" else d["feat"], "tfeat": d["tfeat_p"] if which == "p" else d["tfeat"]}
        out.append((name, fn, leaves, seed))
    return out


def test_ppi_auxiliary_criteria_match_reference(golden_criterion, golden_criterion_ppi):
    """The BCE-flavoured fitnet / at / gpw / lpw / nce of /root/reference/ppi_pyg/criterion.py:21-146 (goldens produced by that
    file itself) against the oracle's restatement: losses and gradients."""
    cases = ppi_aux_cases(golden_criterion, golden_criterion_ppi, OC)
    assert len(cases) == 8
    for name, fn, leaves, seed in cases:
        _check(_run(fn, leaves, seed), golden_criterion_ppi, name)


def test_ppi_kd_matches_reference(golden_criterion):
    G = golden_criterion
    pl, py, pt = (as_t(G["in_ppi_" + k]) for k in ("logits", "labels", "teacher"))
    _check(_run(lambda L: OC.ppi_kd_criterion(L["logits"], py, pt, 0.5, 1.0), {"logits": pl}), G, "ppi_kd")


def test_loss_kd_only_alias(golden_criterion):
    G = golden_criterion
    d = {k[3:]: as_t(G[k]) for k in G.files if k.startswith("in_")}
    v = OC.loss_kd_only(d["logits"], d["labels"], d["teacher_logits"], 0.9, 4.0)
    np.testing.assert_allclose(v.numpy(), G["kd__loss_aux"], rtol=1e-6)


def build_graph(G):
    n = G["in_x"].shape[0]
    adj = OS.to_sparse_tensor(as_t(G["in_edge_index"]), n).to_symmetric()
    return adj


def test_graph_structure_matches(golden_train):
    G = golden_train
    adj = build_graph(G)
    rowptr, col, _ = adj.csr()
    assert np.array_equal(rowptr.numpy(), G["adj_rowptr"]) and np.array_equal(col.numpy(), G["adj_col"])
    ei = torch.stack(adj.coo()[:2])
    sub = OU.subgraph(as_t(G["in_train_idx"]), ei, relabel_nodes=True)[0]
    assert np.array_equal(sub.numpy(), G["train_subgraph_edge_index"])


def parse_run(name):
    tag, gnn, mode, flavour, kw = name.split(":")
    hp = dict(alpha=0.9, kd_T=4.0, beta=0.5, nce_T=0.075, max_samples=24, kernel="rbf")
    for item in filter(None, kw.split(",")):
        k, v = item.split("=")
        hp[k] = v if k == "kernel" else (int(v) if k == "max_samples" else float(v))
    return tag, gnn, mode, flavour == "kdaux", hp


def run_training(G, name, M, build_adj, dev="cpu", make_proj=None, subgraph_fn=None):
    """Re-run one golden training run with module set ``M`` (oracle or product)."""
    tag, gnn, mode, kdaux, hp = parse_run(name)
    H, P, L, C = (int(v) for v in G["hp"])
    x, y = as_t(G["in_x"], dev), as_t(G["in_y"], dev)
    tr = as_t(G["in_train_idx"], dev)
    split = {k: as_t(G[f"in_{k}_idx"], dev) for k in ("train", "valid", "test")}
    tfeat, tlog = as_t(G["in_teacher_out_feat"], dev), as_t(G["in_teacher_logits"], dev)
    adj = build_adj(G)
    model = (M.GCN if gnn == "gcn" else M.SAGE)(x.shape[1], H, C, L, 0.0).to(dev)
    model.load_state_dict({k.split("model.", 1)[1]: as_t(G[k], dev) for k in G.files if k.startswith(f"{tag}__init__model.")})
    sp = tp = None
    if mode in ("nce", "gpw", "fitnet"):
        sp, tp = M.make_projection(H, P).to(dev), M.make_projection(tfeat.shape[1], P).to(dev)
    elif mode == "gcd":
        sp, tp = M.ProjectionGCD(H, P).to(dev), M.ProjectionGCD(tfeat.shape[1], P).to(dev)
    if sp is not None:
        sp.load_state_dict({k.split("sproj.", 1)[1]: as_t(G[k], dev) for k in G.files if k.startswith(f"{tag}__init__sproj.")})
        tp.load_state_dict({k.split("tproj.", 1)[1]: as_t(G[k], dev) for k in G.files if k.startswith(f"{tag}__init__tproj.")})
    groups = [{"params": model.parameters(), "lr": 0.01}]
    if sp is not None:
        groups += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
    opt = torch.optim.Adam(groups)
    ei = as_t(G["train_subgraph_edge_index"], dev) if mode == "lpw" else None
    logits0, accs0 = M.evaluate(model, x, adj, y, split)
    np.random.seed(100 + int(tag[3:]))
    losses = [M.train_step(model, x, adj, y, tr, opt, mode, hp, tfeat, tlog, sp, tp, ei, kd_and_aux=kdaux) for _ in range(3)]
    return model, np.array(losses), logits0, np.array(accs0)


def noise_driven(key, n_layers):
    """Pre-BatchNorm biases / running means: zero true gradient, Adam turns rounding noise into steps."""
    if "running_mean" in key or "num_batches_tracked" in key:
        return True
    for i in range(n_layers - 1):
        if key in (f"convs.{i}.bias", f"convs.{i}.lin_l.bias"):
            return True
    return False


def test_train_and_eval_match_reference(golden_train):
    G = golden_train
    for name in G["run_names"]:
        name = str(name)
        tag = name.split(":")[0]
        model, losses, logits0, accs0 = run_training(G, name, OM, build_graph)
        np.testing.assert_allclose(logits0.numpy(), G[f"{tag}__eval0_logits"], rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(accs0, G[f"{tag}__eval0_accs"], err_msg=name)
        np.testing.assert_allclose(losses, G[f"{tag}__losses"], rtol=1e-5, atol=1e-7, err_msg=name)
        for k, v in model.state_dict().items():
            if noise_driven(k, int(G["hp"][2])):
                continue
            np.testing.assert_allclose(v.numpy(), G[f"{tag}__final__model.{k}"], rtol=1e-4, atol=1e-6, err_msg=f"{name}:{k}")


def test_training_with_dropout_matches_reference(golden_train_dropout):
    """The regime the benchmark times: the reference's own GCN / SAGE + train() with dropout 0.5 (tests/golden/train_arxiv_dropout.npz,
    generated from arxiv_pyg/gnn.py by make_golden.py::make_train_dropout_goldens).  The oracle calls F.dropout in the reference's order
    on the same shapes, so the same torch seed gives it the same masks: four steps' losses and the final weights must agree.  This is
    the oracle the GPU trajectories are compared with after the HIP path's masks are injected (oracle/training_parity.py)."""
    G = golden_train_dropout
    for name in G["run_names"]:
        name = str(name)
        tag, gnn, mode, kdaux, hp = parse_run(name)
        H, P, L, C = (int(v) for v in G["hp"])
        x, y = as_t(G["in_x"]), as_t(G["in_y"])
        tr = as_t(G["in_train_idx"])
        tfeat, tlog = as_t(G["in_teacher_out_feat"]), as_t(G["in_teacher_logits"])
        adj = build_graph(G)
        model = (OM.GCN if gnn == "gcn" else OM.SAGE)(x.shape[1], H, C, L, 0.5)
        model.load_state_dict({k.split("model.", 1)[1]: as_t(G[k]) for k in G.files if k.startswith(f"{tag}__init__model.")})
        sp = tp = None
        groups = [{"params": model.parameters(), "lr": 0.01}]
        if mode in ("nce", "gpw"):
            sp, tp = OM.make_projection(H, P), OM.make_projection(tfeat.shape[1], P)
            sp.load_state_dict({k.split("sproj.", 1)[1]: as_t(G[k]) for k in G.files if k.startswith(f"{tag}__init__sproj.")})
            tp.load_state_dict({k.split("tproj.", 1)[1]: as_t(G[k]) for k in G.files if k.startswith(f"{tag}__init__tproj.")})
            groups += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
        opt = torch.optim.Adam(groups)
        ei = as_t(G["train_subgraph_edge_index"]) if mode == "lpw" else None
        i = int(tag[3:])
        torch.manual_seed(700 + i)
        np.random.seed(700 + i)
        losses = [OM.train_step(model, x, adj, y, tr, opt, mode, hp, tfeat, tlog, sp, tp, ei, kd_and_aux=kdaux) for _ in range(4)]
        np.testing.assert_allclose(np.array(losses), G[f"{tag}__losses"], rtol=2e-5, atol=1e-7, err_msg=name)
        assert len({tuple(l) for l in losses}) == 4
        for k, v in model.state_dict().items():
            if noise_driven(k, L):
                continue
            np.testing.assert_allclose(v.numpy(), G[f"{tag}__final__model.{k}"], rtol=2e-4, atol=2e-6, err_msg=f"{name}:{k}")


def test_oracle_gat_and_teachernet_match_reference_bodies(golden_ppi_teacher):
    """The PPI teacher models (ppi_pyg/gnn.py GAT :86-117, TeacherNet :23-47) executed from the reference's own file when
    the golden was made; the oracle restatement with the same weights must reproduce logits and out_feat."""
    import oracle.models as OM
    G = golden_ppi_teacher
    x, ei = torch.from_numpy(G["in_x"]), torch.from_numpy(G["in_edge_index"])
    m = OM.GAT(x.shape[1], 6, G["gat_logits"].shape[1], 3, 0.5, heads=2)
    m.load_state_dict({k[len("gat_param__"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("gat_param__")})
    m.eval()
    with torch.no_grad():
        y = m(x, ei)
    np.testing.assert_allclose(y.numpy(), G["gat_logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m.out_feat.numpy(), G["gat_out_feat"], rtol=1e-5, atol=1e-6)
    torch.manual_seed(4)   # TeacherNet (2.3 M parameters): weights rebuilt from the seed the golden used
    t = OM.TeacherNet(x.shape[1], G["teachernet_logits"].shape[1])
    assert list(t.state_dict().keys()) == list(G["teachernet_keys"])
    sums = np.array([float(v.double().sum()) for v in t.state_dict().values()])
    if not np.allclose(sums, G["teachernet_param_sums"], rtol=1e-9, atol=1e-9):
        pytest.skip("this torch build draws a different init stream than the one the golden was made with")
    t.eval()
    with torch.no_grad():
        yt = t(x, ei)
    np.testing.assert_allclose(yt.numpy(), G["teachernet_logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(t.out_feat.double().sum()), float(G["teachernet_out_feat_sum"]), rtol=1e-6)


def test_oracle_rgcn_matches_reference_body(golden_mag_rgcn):
    """mag_pyg/gnn.py RGCNConv / RGCN.forward / RGCN.inference executed from the reference's own file (MessagePassing and
    SparseTensor shimmed by the oracle); the oracle restatement with the same weights reproduces them."""
    G = golden_mag_rgcn
    sizes, edge_index_dict, key2int, params, args = mag_rgcn_case(G)
    m = OM.RGCN(8, 12, 5, 2, 0.5, sizes, [0], 4)
    m.load_state_dict(params)
    m.eval()
    with torch.no_grad():
        y = m(*args)
        inf = m.inference(args[0], edge_index_dict, key2int)
    np.testing.assert_allclose(y.numpy(), G["forward_logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m.out_feat.numpy(), G["forward_out_feat"], rtol=1e-5, atol=1e-6)
    for j, v in inf.items():
        np.testing.assert_allclose(v.numpy(), G[f"inference__{j}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("mode", ["kd", "supervised"])
def test_oracle_ppi_epochs_match_reference_train_loop(golden_ppi_train, mode):
    """ppi_pyg/gnn.py train() (:185-274; kd = frozen GAT teacher forward inside every step) and test() (:277-288) executed
    from the reference's own file; the oracle's ppi_train_epoch / ppi_test reproduce the three epoch records and micro-F1."""
    G = golden_ppi_train
    graphs, teacher_sd, init = ppi_train_case(G)
    F_in, Cn = graphs[0].x.shape[1], graphs[0].y.shape[1]
    teacher = OM.GAT(F_in, 6, Cn, 3, 0.0, heads=2)
    teacher.load_state_dict(teacher_sd)
    model = OM.GCN(F_in, 16, Cn, 2, 0.0, cached=False)
    model.load_state_dict(init[mode])
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    hp = dict(alpha=0.5, kd_T=1.0)
    recs = [OM.ppi_train_epoch(model, teacher if mode == "kd" else None, graphs, opt, mode, hp) for _ in range(3)]
    np.testing.assert_allclose(np.array(recs), G[f"{mode}_epoch_losses"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(OM.ppi_test(model, graphs), float(G[f"{mode}_f1"]), atol=2e-3)


@pytest.mark.parametrize("name", sorted(ARXIV_GAT_CONFIGS))
def test_oracle_arxiv_gat_teacher_matches_reference_bodies(golden_arxiv_gat, name):
    """oracle.ArxivGAT / DGLGATConv / teacher_evaluate against the reference's own arxiv_dgl/models.py GAT (eval-mode forward,
    label reuse as in gat.py:151-166) recorded by tests/golden/make_golden.py: predictions (= the logits artefact), the last
    hidden features (= the features artefact) and the raw first layer."""
    G = golden_arxiv_gat
    model, adj, x, labels, (tr, va, te), C, iters = arxiv_gat_case(G, OM, OS.SparseTensor, name)
    model.eval()
    pred, feat = OM.teacher_evaluate(model, adj, x, labels, tr, va, te, C, use_labels=True, n_label_iters=iters)
    np.testing.assert_allclose(pred.numpy(), G[f"{name}__pred"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(feat.numpy(), G[f"{name}__feat"], rtol=2e-5, atol=1e-6)
    # first layer on the final (label-reused) input features
    f = OM.add_labels(x, labels, tr, C)
    p = model(adj, f)
    un = torch.cat([va, te])
    for _ in range(iters):
        f[un, -C:] = torch.softmax(p[un], dim=-1)
        p = model(adj, f)
    with torch.no_grad():
        np.testing.assert_allclose(model.convs[0](adj, f).numpy(), G[f"{name}__conv0"], rtol=2e-5, atol=1e-6)
