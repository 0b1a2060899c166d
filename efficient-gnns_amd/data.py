"""Seeded synthetic datasets with the shapes of the reference's datasets (none is on disk and there is no
network).  Integer / host-side generation only -- tensors are moved to the GPU by the caller.

  * ``arxiv_like``  ogbn-arxiv: N=169 343, 1 166 243 directed power-law edges (no self loops, no
                    multi-edges), x [N,128], 40 classes, split 90 941 / 29 799 / 48 603, GAT-teacher
                    artefacts [N,750] (non-negative, post-ReLU) and [N,40]
                    (/root/reference/arxiv_pyg/gnn.py:236-279)
  * ``ppi_like``    PPI: 20+2+2 graphs, x [n,50], 121 labels (/root/reference/ppi_pyg/gnn.py:301-310)
  * ``mag_like``    ogbn-mag grouped homogeneous graph (/root/reference/mag_pyg/gnn.py:322-346)
"""
from __future__ import annotations

import types

import numpy as np
import torch

ARXIV = dict(num_nodes=169_343, num_edges=1_166_243, num_features=128, num_classes=40,
             split=(90_941, 29_799, 48_603), teacher_dim=750, max_degree=13_000)


def powerlaw_edges(n: int, e: int, gamma: float = 2.2, max_degree: int | None = None, seed: int = 0) -> np.ndarray:
    """Directed edge list [2, e] (source, target): power-law in-degree (Chung-Lu weights w_i ~ (i+i0)^-1/(gamma-1)),
    near-uniform out-degree, no self loops, no duplicate edges, node ids randomly permuted."""
    rng = np.random.default_rng(seed)
    alpha = 1.0 / (gamma - 1.0)
    ranks = np.arange(n, dtype=np.float64)

    def probs(i0):
        w = (ranks + i0) ** (-alpha)
        return w / w.sum()

    i0 = 1.0
    if max_degree is not None:  # bisect the offset so that the expected hub in-degree ~ max_degree
        lo, hi = 1e-3, 1e4
        for _ in range(60):
            mid = (lo * hi) ** 0.5
            if probs(mid)[0] * e > max_degree:
                lo = mid
            else:
                hi = mid
        i0 = hi
    p = probs(i0)
    perm = rng.permutation(n)
    keys = np.empty(0, dtype=np.int64)
    need = e
    while need > 0:
        m = int(need * 1.1) + 16
        dst = perm[rng.choice(n, size=m, p=p)]
        src = rng.integers(0, n, size=m)
        ok = src != dst
        k = src[ok].astype(np.int64) * n + dst[ok]
        keys = np.unique(np.concatenate([keys, k]))
        need = e - keys.size
    if keys.size > e:
        keys = np.sort(rng.choice(keys, size=e, replace=False))
    return np.stack([keys // n, keys % n])


def community_edges(n: int, e: int, gamma: float = 2.2, max_degree: int | None = None, seed: int = 0,
                    mean_community: int = 400, mu: float = 0.25, shuffle_ids: bool = True):
    """Directed edge list [2, e] with the SAME in-degree law as ``powerlaw_edges`` (Chung-Lu weights, same hub bisection)
    plus COMMUNITY STRUCTURE (a degree-corrected stochastic block model): nodes belong to communities of log-normal size
    (mean ``mean_community``); the source of an edge is drawn from the target's community with probability 1 - mu and from
    the whole graph otherwise -- hubs whose in-degree exceeds half their community draw the surplus globally, so the degree
    law survives.  Citation graphs look like this (papers cite inside their sub-field); the plain Chung-Lu graph is the
    locality-free worst case.  ``shuffle_ids`` permutes the node ids, as real datasets come, so that locality has to be
    DISCOVERED by a reorder pass (``SparseTensor.reorder``).  Returns (edges [2,e], community [n] of the returned ids)."""
    rng = np.random.default_rng(seed)
    alpha = 1.0 / (gamma - 1.0)
    ranks = np.arange(n, dtype=np.float64)

    def probs(i0):
        w = (ranks + i0) ** (-alpha)
        return w / w.sum()

    i0 = 1.0
    if max_degree is not None:
        lo, hi = 1e-3, 1e4
        for _ in range(60):
            mid = (lo * hi) ** 0.5
            if probs(mid)[0] * e > max_degree:
                lo = mid
            else:
                hi = mid
        i0 = hi
    p = probs(i0)
    # communities over a hidden order in which degree ranks are scattered (hubs spread over communities)
    sizes = []
    left = n
    while left > 0:
        s = int(min(left, max(32, rng.lognormal(np.log(mean_community) - 0.125, 0.5))))
        sizes.append(s)
        left -= s
    sizes = np.array(sizes, dtype=np.int64)
    comm_start = np.concatenate([[0], np.cumsum(sizes)])
    comm_of_pos = np.repeat(np.arange(sizes.size), sizes)
    pos_of_rank = rng.permutation(n)                 # degree rank r sits at hidden position pos_of_rank[r]
    exp_deg = p * e
    keys = np.empty(0, dtype=np.int64)
    need = e
    while need > 0:
        m = int(need * 1.15) + 16
        r = rng.choice(n, size=m, p=p)               # target by degree rank
        dst = pos_of_rank[r]
        c = comm_of_pos[dst]
        p_local = (1.0 - mu) * np.minimum(1.0, 0.5 * sizes[c] / np.maximum(exp_deg[r], 1.0))
        local = rng.random(m) < p_local
        src = np.where(local, comm_start[c] + (rng.random(m) * sizes[c]).astype(np.int64), rng.integers(0, n, size=m))
        ok = src != dst
        k = src[ok].astype(np.int64) * n + dst[ok]
        keys = np.unique(np.concatenate([keys, k]))
        need = e - keys.size
    if keys.size > e:
        keys = np.sort(rng.choice(keys, size=e, replace=False))
    src, dst = keys // n, keys % n
    if shuffle_ids:
        relabel = rng.permutation(n)
        src, dst = relabel[src], relabel[dst]
        community = np.empty(n, dtype=np.int64)
        community[relabel] = comm_of_pos
    else:
        community = comm_of_pos
    return np.stack([src, dst]), community


TEACHER_FEAT_SCALE = 1.0 / 16.0


def arxiv_like(scale: float = 1.0, seed: int = 0, with_teacher: bool = True, graph: str = "chunglu",
               teacher_feat_scale: float = TEACHER_FEAT_SCALE):
    """Synthetic ogbn-arxiv-shaped node-classification problem (CPU tensors).  ``scale`` < 1 shrinks N and E
    proportionally (test sizes); scale=1 is the BASELINE.json workload.  ``graph``: 'chunglu' (the headline workload of
    SURVEY 8d: power law, no locality at all) | 'local' (same degree law + community structure, ids shuffled: locality
    must be found by ``SparseTensor.reorder``) | 'local-sorted' (the same graph with ids already in community order).
    ``teacher_feat_scale``: the [N,750] teacher features are relu(N(0,1)) * scale.  At scale 1 two rows are ~22 apart
    (||a-b||^2 ~ 510), so the rbf similarities exp(-||a-b||^2 / 2) of the LSP / GSP losses (criterion.py:78,111) underflow to
    exactly 0 and those configurations compare nothing; 1/16 puts the median ||a-b||^2 at ~2 (similarities ~0.37, edge
    distributions far from uniform).  Non-negative with ~50 % zeros either way; the projection heads start
    with a BatchNorm, so G-CRD / GSP-cosine see the same normalised inputs at any scale."""
    from .transforms import to_sparse_tensor
    n = max(64, int(round(ARXIV["num_nodes"] * scale)))
    e = max(128, int(round(ARXIV["num_edges"] * scale)))
    md = max(8, int(ARXIV["max_degree"] * min(1.0, scale * 4)))
    community = None
    if graph == "chunglu":
        ei = torch.from_numpy(powerlaw_edges(n, e, max_degree=md, seed=seed))
    elif graph in ("local", "local-sorted"):
        edges, community = community_edges(n, e, max_degree=md, seed=seed, shuffle_ids=(graph == "local"))
        ei = torch.from_numpy(edges)
    else:
        raise ValueError(f"unknown synthetic graph '{graph}'")
    g = torch.Generator().manual_seed(seed)
    d = types.SimpleNamespace()
    d.num_nodes, d.num_features, d.num_classes = n, ARXIV["num_features"], ARXIV["num_classes"]
    d.edge_index = ei
    d.community = None if community is None else torch.from_numpy(community)
    d.x = torch.randn(n, d.num_features, generator=g)
    d.y = torch.randint(0, d.num_classes, (n, 1), generator=g)
    tr, va, te = ARXIV["split"]
    n_tr = int(round(n * tr / ARXIV["num_nodes"]))
    n_va = int(round(n * va / ARXIV["num_nodes"]))
    perm = torch.randperm(n, generator=g)
    d.split_idx = {"train": perm[:n_tr].clone(), "valid": perm[n_tr:n_tr + n_va].clone(), "test": perm[n_tr + n_va:].clone()}
    d.adj_t = to_sparse_tensor(ei, n).to_symmetric()  # gnn.py:237-240
    d.edge_index = None
    if with_teacher:
        d.teacher_out_feat = torch.relu(torch.randn(n, ARXIV["teacher_dim"], generator=g)) * teacher_feat_scale
        d.teacher_logits = torch.randn(n, d.num_classes, generator=g) * 3.0
    return d


def ppi_like(seed: int = 0, n_train: int = 20, total_train_nodes: int = 44_906, feats: int = 50, labels: int = 121):
    """List of per-graph namespaces (x, y, edge_index symmetric LongTensor[2,E], teacher_logits)."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(600, 3500, size=n_train).astype(np.float64)
    sizes = np.maximum(64, np.round(sizes * total_train_nodes / sizes.sum())).astype(np.int64)
    sizes = list(sizes) + [3257, 3257, 2762, 2762]  # 2 validation (6 514) + 2 test (5 524) graphs
    graphs = []
    for gi, n in enumerate(sizes):
        n = int(n)
        e = int(n * 13.5)
        ei = powerlaw_edges(n, e, gamma=2.6, max_degree=max(8, n // 8), seed=seed * 1000 + gi)
        both = np.unique(np.concatenate([ei[0] * n + ei[1], ei[1] * n + ei[0]]))
        g = torch.Generator().manual_seed(seed * 1000 + gi)
        d = types.SimpleNamespace()
        d.edge_index = torch.from_numpy(np.stack([both // n, both % n]))
        d.x = torch.randn(n, feats, generator=g)
        d.y = (torch.rand(n, labels, generator=g) < 0.3).float()
        d.teacher_logits = torch.randn(n, labels, generator=g) * 2.0
        d.num_nodes = n
        graphs.append(d)
    return graphs[:n_train], graphs[n_train:n_train + 2], graphs[n_train + 2:]


def mag_like(scale: float = 1.0, seed: int = 0, feats: int = 128):
    """Homogeneous MAG-shaped graph: N=1 939 743, 42 182 144 directed edges after reverse edges."""
    from .transforms import to_sparse_tensor
    n = max(256, int(round(1_939_743 * scale)))
    e = max(512, int(round(21_091_072 * scale)))
    ei = torch.from_numpy(powerlaw_edges(n, e, gamma=2.3, max_degree=max(16, int(30_000 * min(1.0, scale * 4))), seed=seed))
    g = torch.Generator().manual_seed(seed)
    d = types.SimpleNamespace()
    d.num_nodes, d.num_features, d.num_classes = n, feats, 349
    d.x = torch.randn(n, feats, generator=g)
    d.adj_t = to_sparse_tensor(ei, n).to_symmetric()
    return d


# ------------------------------------------------------------------------------------------------
# teacher artefacts on disk: what the teacher run leaves for the student run
# ------------------------------------------------------------------------------------------------
def teacher_artifact_paths(root: str, expt_name: str, seed: int):
    """The reference's layout: ``arxiv_dgl/gat.py:245-251`` writes ``features/<expt>/<seed>.pt`` ([N, 750] fp32 hidden
    features of the last GAT layer) and ``logits/<expt>/<seed>.pt`` ([N, 40] fp32); ``arxiv_pyg/gnn.py:278-279`` reads
    them back with ``torch.load`` (expt ``gat-3L250x3h``)."""
    import os
    return (os.path.join(root, "features", expt_name, f"{seed}.pt"), os.path.join(root, "logits", expt_name, f"{seed}.pt"))


def save_teacher_artifacts(root: str, expt_name: str, seed: int, out_feat: torch.Tensor, logits: torch.Tensor) -> None:
    import os
    f_path, l_path = teacher_artifact_paths(root, expt_name, seed)
    for path, t in ((f_path, out_feat), (l_path, logits)):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(t.detach().to("cpu", torch.float32).contiguous(), path)


def load_teacher_artifacts(root: str, expt_name: str, seed: int, num_nodes: int | None = None, device="cpu"):
    """(teacher_out_feat, teacher_logits) as ``gnn.py:278-279`` loads them; on a GPU the features come back behind a
    16-byte-aligned row pitch (``ops.pad_pitch``) so that the fused gather-GEMM of the projection head takes its float4 path."""
    f_path, l_path = teacher_artifact_paths(root, expt_name, seed)
    feat = torch.load(f_path, map_location="cpu")
    logits = torch.load(l_path, map_location="cpu")
    if feat.dim() != 2 or logits.dim() != 2 or feat.shape[0] != logits.shape[0]:
        raise ValueError(f"teacher artefacts disagree: features {tuple(feat.shape)}, logits {tuple(logits.shape)}")
    if num_nodes is not None and feat.shape[0] != num_nodes:
        raise ValueError(f"teacher artefacts are for {feat.shape[0]} nodes, the graph has {num_nodes}")
    feat, logits = feat.to(device, torch.float32), logits.to(device, torch.float32)
    if feat.is_cuda:
        from .ops import pad_pitch
        feat = pad_pitch(feat)
    return feat, logits
