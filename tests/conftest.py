import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore", message=".*reduction: 'mean' divides.*")
warnings.filterwarnings("ignore", message=".*Sparse CSR tensor support is in beta.*")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_criterion():
    return np.load(os.path.join(GOLDEN, "criterion.npz"))


@pytest.fixture(scope="session")
def golden_train_dropout():
    return np.load(os.path.join(GOLDEN, "train_arxiv_dropout.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_criterion_ppi():
    return np.load(os.path.join(GOLDEN, "criterion_ppi.npz"))


@pytest.fixture(scope="session")
def golden_ppi_teacher():
    return np.load(os.path.join(GOLDEN, "ppi_teacher.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_mag_rgcn():
    return np.load(os.path.join(GOLDEN, "mag_rgcn.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_ppi_train():
    return np.load(os.path.join(GOLDEN, "ppi_train.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_arxiv_gat():
    return np.load(os.path.join(GOLDEN, "arxiv_gat.npz"), allow_pickle=False)


ARXIV_GAT_CONFIGS = {"norm_noattn": dict(use_attn_dst=False, use_symmetric_norm=True, n_label_iters=1),
                     "plain": dict(use_attn_dst=True, use_symmetric_norm=False, n_label_iters=0),
                     "norm_attn": dict(use_attn_dst=True, use_symmetric_norm=True, n_label_iters=2)}


def arxiv_gat_case(G, M, SparseTensor, name, device="cpu"):
    """Model (``M.ArxivGAT`` with the golden's parameters), message graph and inputs of tests/golden/arxiv_gat.npz."""
    import torch.nn.functional as F
    cfg = ARXIV_GAT_CONFIGS[name]
    x, labels = as_t(G["in_x"], device), as_t(G["in_labels"], device)
    n, C = x.shape[0], 6
    adj = SparseTensor(row=as_t(G["in_dst"], device), col=as_t(G["in_src"], device), sparse_sizes=(n, n))
    model = M.ArxivGAT(x.shape[1] + C, C, 5, 3, 3, F.relu, dropout=0.75, input_drop=0.25, attn_drop=0.0, edge_drop=0.3,
                       use_attn_dst=cfg["use_attn_dst"], use_symmetric_norm=cfg["use_symmetric_norm"]).to(device)
    pre = f"{name}__param__"
    sd = {k[len(pre):]: as_t(G[k], device) for k in G.files if k.startswith(pre)}
    missing = model.load_state_dict(sd, strict=True)
    idx = tuple(as_t(G[k], device) for k in ("in_train", "in_val", "in_test"))
    return model, adj, x, labels, idx, C, cfg["n_label_iters"]


@pytest.fixture(scope="session")
def golden_train():
    return np.load(os.path.join(GOLDEN, "train_arxiv.npz"), allow_pickle=False)


def as_t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def mag_rgcn_case(G, device="cpu"):
    """Inputs of tests/golden/mag_rgcn.npz in the shapes RGCN.forward / .inference take."""
    sizes = {0: 23, 1: 17, 2: 9}
    rel_names = [k[len("in_rel__"):] for k in G.files if k.startswith("in_rel__")]
    key2int = {"a": 0, "b": 1, "c": 2}
    edge_index_dict = {}
    for i, name in enumerate(rel_names):
        key = tuple(name.split("|"))
        r, c = as_t(G["in_rel__" + name], device)
        edge_index_dict[key] = (r, c)
        key2int[key] = i
    params = {k[len("param__"):]: as_t(G[k], device) for k in G.files if k.startswith("param__")}
    args = ({0: as_t(G["in_x0"], device)}, as_t(G["in_edge_index"], device), as_t(G["in_edge_type"], device),
            as_t(G["in_node_type"], device), as_t(G["in_local_idx"], device))
    return sizes, edge_index_dict, key2int, params, args


def ppi_train_case(G, device="cpu"):
    """Batch graphs and parameter dicts of tests/golden/ppi_train.npz."""
    import types
    graphs = []
    i = 0
    while f"in_x{i}" in G.files:
        graphs.append(types.SimpleNamespace(x=as_t(G[f"in_x{i}"], device), edge_index=as_t(G[f"in_ei{i}"], device),
                                            y=as_t(G[f"in_y{i}"], device)))
        i += 1
    sd = lambda prefix: {k[len(prefix):]: as_t(G[k], device) for k in G.files if k.startswith(prefix)}  # noqa: E731
    return graphs, sd("teacher__"), {m: sd(f"{m}_init__") for m in ("kd", "supervised")}
