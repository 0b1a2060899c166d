"""The reference's OWN ``arxiv_pyg/gnn.py`` (train()/test(), GCN/SAGE classes) running unchanged on top of
``efficient-gnns_amd/dropin``.  Needs /root/reference, so it runs in the build container only (and needs a GPU for
the kernels): on the GPU box the reference is absent and the test skips -- tests/golden/ covers the same path there."""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT

REF = "/root/reference/arxiv_pyg/gnn.py"
DROPIN = os.path.join(ROOT, "efficient-gnns_amd", "dropin")
_SHIMMED = ("criterion", "torch_geometric", "torch_geometric.nn", "torch_geometric.utils", "torch_geometric.transforms", "torch_sparse")


def _load_reference_gnn():
    for name in ("ogb", "ogb.nodeproppred", "torch.utils.tensorboard", "logger"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["ogb.nodeproppred"].PygNodePropPredDataset = None
    sys.modules["ogb.nodeproppred"].Evaluator = object
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    sys.modules["logger"].Logger = object
    sys.path.insert(0, DROPIN)
    for stale in _SHIMMED:
        sys.modules.pop(stale, None)
    spec = importlib.util.spec_from_file_location("ref_gnn_dropin", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dropin_names_resolve_on_cpu():
    """Import-level drop-in check (no GPU needed): every name the reference imports resolves to the new package."""
    sys.path.insert(0, DROPIN)
    try:
        for stale in _SHIMMED:
            sys.modules.pop(stale, None)
        import criterion as C
        import torch_geometric.transforms as T
        from torch_geometric.nn import GCNConv, SAGEConv
        from torch_geometric.utils import softmax, subgraph, to_dense_adj, negative_sampling, add_self_loops  # noqa: F401
        from torch_sparse import SparseTensor
        import efficient_gnns_amd as E
        assert GCNConv is E.GCNConv and SAGEConv is E.SAGEConv and SparseTensor is E.SparseTensor
        assert T.ToSparseTensor is E.ToSparseTensor
        for fn in ("kd_criterion", "fitnet_criterion", "at_criterion", "gpw_criterion", "lpw_criterion", "nce_criterion"):
            assert callable(getattr(C, fn))
        import inspect
        assert str(inspect.signature(C.nce_criterion)) == "(logits, labels, feat, teacher_feat, beta=0.5, nce_T=0.075, max_samples=8192)"
        assert str(inspect.signature(C.kd_criterion)) == "(logits, labels, teacher_logits, alpha=0.9, T=4)"
        assert str(inspect.signature(C.lpw_criterion)) == "(logits, labels, feat, teacher_feat, edge_index, kernel='cosine', beta=100, criterion='kld')"
        assert str(inspect.signature(C.gpw_criterion)) == "(logits, labels, feat, teacher_feat, kernel='cosine', beta=1, max_samples=8192)"
        assert str(inspect.signature(C.fitnet_criterion)) == "(logits, labels, feat, teacher_feat, beta=1000)"
        assert str(inspect.signature(C.at_criterion)) == "(logits, labels, feat, teacher_feat, beta=1000)"
    finally:
        sys.path.remove(DROPIN)
        for stale in _SHIMMED:
            sys.modules.pop(stale, None)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is only present in the build container")
def test_reference_train_loop_runs_unchanged_on_the_new_operators():
    import efficient_gnns_amd.data as D
    ref = _load_reference_gnn()
    dev = torch.device("cuda")
    d = D.arxiv_like(scale=0.01, seed=2)
    data = types.SimpleNamespace(x=d.x.to(dev), y=d.y.to(dev), adj_t=d.adj_t.to(dev))
    split_idx = d.split_idx
    train_idx = split_idx["train"].to(dev)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ref.GCN(d.num_features, 64, d.num_classes, 3, 0.5).to(dev)
    sp = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.BatchNorm1d(32), torch.nn.ReLU()).to(dev)
    tp = torch.nn.Sequential(torch.nn.Linear(750, 32), torch.nn.BatchNorm1d(32), torch.nn.ReLU()).to(dev)
    opt = torch.optim.Adam([{"params": model.parameters()}, {"params": sp.parameters()}, {"params": tp.parameters()}], lr=0.01)
    args = argparse.Namespace(training="nce", alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=256, kernel="rbf")

    class Ev:
        def eval(self, dd):
            return {"acc": float((dd["y_true"].cpu().numpy() == dd["y_pred"].cpu().numpy()).mean())}
    tf, tl = d.teacher_out_feat.to(dev), d.teacher_logits.to(dev)
    l0 = ref.train(model, data, train_idx, opt, args, tf, tl, sp, tp, None)
    for _ in range(5):
        l1 = ref.train(model, data, train_idx, opt, args, tf, tl, sp, tp, None)
    out, accs = ref.test(model, data, split_idx, Ev())
    assert all(np.isfinite(l0)) and all(np.isfinite(l1)) and l1[0] < l0[0], (l0, l1)
    assert out.shape == (d.num_nodes, d.num_classes) and all(0 <= a <= 1 for a in accs)
