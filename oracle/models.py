"""ORACLE (test infrastructure): student models and the train / eval step of the reference.

Restates /root/reference/arxiv_pyg/gnn.py:23-99 (``GCN``, ``SAGE``, ``ProjectionGCD``), :102-195
(``train``) and :198-218 (``test``) on top of the oracle convs, plus the KD+aux combination rule of
/root/reference/arxiv_pyg/gnn_kd_and_aux.py:114-181.  ``state_dict`` keys match the reference
(``convs.i.*``, ``bns.i.*``).  Pinned against the reference's own ``gnn.py`` by
``tests/golden/make_golden.py``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import criterion as C
from .nn import GCNConv, SAGEConv


class _Student(nn.Module):
    """[conv -> BatchNorm1d -> ReLU -> dropout] x (L-1) -> conv; exposes ``out_feat`` (gnn.py:45-53)."""

    def __init__(self, make_conv, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__()
        dims = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        self.convs = nn.ModuleList(make_conv(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.bns = nn.ModuleList(nn.BatchNorm1d(hidden_channels) for _ in range(num_layers - 1))
        self.dropout = dropout
        self.out_feat = None

    def reset_parameters(self):
        for m in list(self.convs) + list(self.bns):
            m.reset_parameters()

    def forward(self, x, adj_t):
        for conv, bn in zip(self.convs[:-1], self.bns):
            x = F.dropout(F.relu(bn(conv(x, adj_t))), p=self.dropout, training=self.training)
            self.out_feat = x  # post-dropout activations of the last hidden layer (gnn.py:51)
        return self.convs[-1](x, adj_t)


class GCN(_Student):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, cached=True):
        super().__init__(lambda a, b: GCNConv(a, b, cached=cached),
                         in_channels, hidden_channels, out_channels, num_layers, dropout)


class SAGE(_Student):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__(SAGEConv, in_channels, hidden_channels, out_channels, num_layers, dropout)


class ProjectionGCD(nn.Module):
    """Linear + non-cached GCNConv + BN + ReLU on the full graph (gnn.py:88-99)."""

    def __init__(self, hidden_channels, proj_dim):
        super().__init__()
        self.lin = nn.Linear(hidden_channels, proj_dim)
        self.conv = GCNConv(hidden_channels, proj_dim)
        self.bn = nn.BatchNorm1d(proj_dim)

    def forward(self, x, adj_t):
        return F.relu(self.bn(self.lin(x) + self.conv(x, adj_t)))


def make_projection(in_dim, proj_dim):
    """Linear + BN + ReLU head (gnn.py:296-306)."""
    return nn.Sequential(nn.Linear(in_dim, proj_dim), nn.BatchNorm1d(proj_dim), nn.ReLU())


def distill_loss(mode, model, out, labels, train_idx, teacher_out_feat, teacher_logits, hp,
                 student_proj=None, teacher_proj=None, edge_index=None, adj_t=None, kd_and_aux=False):
    """The loss branch of ``train()`` (gnn.py:112-189); ``kd_and_aux`` = gnn_kd_and_aux.py:114-181."""
    if mode == "supervised":
        loss = F.cross_entropy(out, labels)
        return loss, loss, loss * 0
    if mode == "kd":
        return C.kd_criterion(out, labels, teacher_logits[train_idx], hp["alpha"], hp["kd_T"])
    if mode in ("fitnet", "gpw", "nce"):
        f = student_proj(model.out_feat[train_idx])
        t = teacher_proj(teacher_out_feat[train_idx])
    elif mode in ("at", "lpw"):
        f, t = model.out_feat[train_idx], teacher_out_feat[train_idx]
    elif mode == "gcd":
        f = student_proj(model.out_feat, adj_t)[train_idx]
        t = teacher_proj(teacher_out_feat, adj_t)[train_idx]
    else:
        raise NotImplementedError(mode)
    if mode == "fitnet":
        res = C.fitnet_criterion(out, labels, f, t, hp["beta"])
    elif mode == "at":
        res = C.at_criterion(out, labels, f, t, hp["beta"])
    elif mode == "gpw":
        res = C.gpw_criterion(out, labels, f, t, hp["kernel"], hp["beta"], hp["max_samples"])
    elif mode == "lpw":
        res = C.lpw_criterion(out, labels, f, t, edge_index, hp["kernel"], hp["beta"])
    else:  # nce, gcd
        res = C.nce_criterion(out, labels, f, t, hp["beta"], hp["nce_T"], hp["max_samples"])
    if not kd_and_aux:
        return res
    loss_aux = res[2]
    loss, loss_cls, _ = C.kd_criterion(out, labels, teacher_logits[train_idx], hp["alpha"], hp["kd_T"])
    return loss + hp["beta"] * loss_aux, loss_cls, loss_aux


def train_step(model, x, adj_t, y, train_idx, optimizer, mode, hp, teacher_out_feat=None, teacher_logits=None,
               student_proj=None, teacher_proj=None, edge_index=None, kd_and_aux=False):
    """One optimisation step = one epoch of full-graph training (gnn.py:102-195)."""
    model.train()
    for p in (student_proj, teacher_proj):
        if p is not None:
            p.train()
    out = model(x, adj_t)[train_idx]
    labels = y.squeeze(1)[train_idx]
    loss, loss_cls, loss_aux = distill_loss(mode, model, out, labels, train_idx, teacher_out_feat, teacher_logits,
                                            hp, student_proj, teacher_proj, edge_index, adj_t, kd_and_aux)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.item(), loss_cls.item(), loss_aux.item()


def accuracy(y_true, y_pred):
    """``ogb`` Evaluator('ogbn-arxiv')['acc']: mean of y_true == y_pred via NumPy."""
    return float((y_true.cpu().numpy() == y_pred.cpu().numpy()).mean())


@torch.no_grad()
def evaluate(model, x, adj_t, y, split_idx):
    """Eval forward + argmax + 3 accuracies (gnn.py:198-218)."""
    model.eval()
    out = model(x, adj_t)
    y_pred = out.argmax(dim=-1, keepdim=True)
    accs = tuple(accuracy(y[split_idx[k]], y_pred[split_idx[k]]) for k in ("train", "valid", "test"))
    return out, accs


class GAT(nn.Module):
    """The PPI teacher (/root/reference/ppi_pyg/gnn.py:86-117): GATConv + linear skip per layer, ELU, dropout; the last
    layer averages its heads.  ``out_feat`` = last hidden (post-dropout), read by the feature-distillation losses."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, heads=4):
        super().__init__()
        from .nn import GATConv
        self.convs = nn.ModuleList([GATConv(in_channels, hidden_channels, heads=heads)])
        self.lins = nn.ModuleList([nn.Linear(in_channels, hidden_channels * heads)])
        for _ in range(num_layers - 2):
            self.convs.append(GATConv(heads * hidden_channels, hidden_channels, heads=heads))
            self.lins.append(nn.Linear(hidden_channels * heads, hidden_channels * heads))
        self.convs.append(GATConv(heads * hidden_channels, out_channels, heads=heads, concat=False))
        self.lins.append(nn.Linear(hidden_channels * heads, out_channels))
        self.dropout = dropout
        self.out_feat = None

    def reset_parameters(self):
        for m in list(self.convs) + list(self.lins):
            m.reset_parameters()

    def forward(self, x, adj_t):
        for conv, lin in zip(self.convs[:-1], self.lins[:-1]):
            x = conv(x, adj_t) + lin(x)
            x = F.elu(x)
            x = F.dropout(x, p=self.dropout, training=self.training)
            self.out_feat = x
        return self.convs[-1](x, adj_t) + self.lins[-1](x)


class TeacherNet(nn.Module):
    """The PPI teacher checkpointed by the reference (/root/reference/ppi_pyg/gnn.py:23-47): 4 x 256 GAT layers with
    linear skips, ELU, a 6-head averaging output layer; ``out_feat`` = second hidden.  Same attribute names (state_dict
    keys ``conv1.*``, ``lin1.*``, ...) so that the reference's ``checkpoint.pt`` loads."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        from .nn import GATConv
        self.conv1 = GATConv(in_channels, 256, heads=4)
        self.lin1 = nn.Linear(in_channels, 4 * 256)
        self.conv2 = GATConv(4 * 256, 256, heads=4)
        self.lin2 = nn.Linear(4 * 256, 4 * 256)
        self.conv3 = GATConv(4 * 256, out_channels, heads=6, concat=False)
        self.lin3 = nn.Linear(4 * 256, out_channels)
        self.out_feat = None

    def reset_parameters(self):
        for m in (self.conv1, self.conv2, self.conv3, self.lin1, self.lin2, self.lin3):
            m.reset_parameters()

    def forward(self, x, edge_index):
        x = F.elu(self.conv1(x, edge_index) + self.lin1(x))
        x = F.elu(self.conv2(x, edge_index) + self.lin2(x))
        self.out_feat = x
        return self.conv3(x, edge_index) + self.lin3(x)


class RGCN(nn.Module):
    """/root/reference/mag_pyg/gnn.py:71-168: R-GCN over the grouped heterogeneous graph; ``forward`` on a (sampled)
    homogeneous subgraph, ``inference`` full-batch per relation with ``adj_t.matmul(x, reduce='mean')`` (:151,162)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, num_nodes_dict, x_types, num_edge_types):
        super().__init__()
        from .nn import RGCNConv
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, hidden_channels, out_channels
        self.num_layers, self.dropout = num_layers, dropout
        node_types = list(num_nodes_dict.keys())
        self.num_node_types, self.num_edge_types = len(node_types), num_edge_types
        self.emb_dict = nn.ParameterDict({f"{key}": nn.Parameter(torch.empty(num_nodes_dict[key], in_channels))
                                          for key in sorted(set(node_types).difference(set(x_types)))})
        dims = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        self.convs = nn.ModuleList(RGCNConv(a, b, self.num_node_types, num_edge_types) for a, b in zip(dims[:-1], dims[1:]))
        self.out_feat = None
        self.reset_parameters()

    def reset_parameters(self):
        for emb in self.emb_dict.values():
            nn.init.xavier_uniform_(emb)
        for conv in self.convs:
            conv.reset_parameters()

    def group_input(self, x_dict, node_type, local_node_idx):
        h = torch.zeros((node_type.size(0), self.in_channels), device=node_type.device)
        for key, x in x_dict.items():
            mask = node_type == key
            h[mask] = x[local_node_idx[mask]]
        for key, emb in self.emb_dict.items():
            mask = node_type == int(key)
            h[mask] = emb[local_node_idx[mask]]
        return h

    def forward(self, x_dict, edge_index, edge_type, node_type, local_node_idx):
        x = self.group_input(x_dict, node_type, local_node_idx)
        for i, conv in enumerate(self.convs):
            x = conv(x, edge_index, edge_type, node_type)
            if i != self.num_layers - 1:
                x = F.dropout(F.relu(x), p=0.5, training=self.training)
                self.out_feat = x
        return x

    def inference(self, x_dict, edge_index_dict, key2int):
        from .sparse import SparseTensor, matmul
        x_dict = dict(x_dict)
        for key, emb in self.emb_dict.items():
            x_dict[int(key)] = emb
        adj_t_dict = {}
        for key, (row, col) in edge_index_dict.items():
            n_dst, n_src = x_dict[key2int[key[-1]]].shape[0], x_dict[key2int[key[0]]].shape[0]
            adj_t_dict[key] = SparseTensor(row=col, col=row, sparse_sizes=(n_dst, n_src))
        for i, conv in enumerate(self.convs):
            out_dict = {j: conv.root_lins[j](x) for j, x in x_dict.items()}
            for keys, adj_t in adj_t_dict.items():
                tmp = matmul(adj_t, x_dict[key2int[keys[0]]], "mean")
                out_dict[key2int[keys[-1]]] = out_dict[key2int[keys[-1]]] + conv.rel_lins[key2int[keys]](tmp)
            if i != self.num_layers - 1:
                out_dict = {j: F.relu(v) for j, v in out_dict.items()}
            x_dict = out_dict
        return x_dict


def ppi_train_epoch(model, teacher_model, graphs, optimizer, mode, hp):
    """One PPI epoch (/root/reference/ppi_pyg/gnn.py:185-274): one optimisation step per batch graph; in ``kd`` mode the
    frozen teacher's forward runs inside every step (:208-209).  ``graphs``: objects with x / edge_index / y.
    Returns the epoch means (loss, loss_cls, loss_aux) like the reference."""
    model.train()
    if teacher_model is not None:
        teacher_model.eval()
    tot = [0.0, 0.0, 0.0]
    for g in graphs:
        out = model(g.x, g.edge_index)
        if mode == "supervised":
            loss = F.binary_cross_entropy_with_logits(out, g.y)
            loss_cls, loss_aux = loss, loss * 0
        elif mode == "kd":
            with torch.no_grad():
                teacher_out = teacher_model(g.x, g.edge_index)
            loss, loss_cls, loss_aux = C.ppi_kd_criterion(out, g.y, teacher_out, hp["alpha"], hp["kd_T"])
        else:
            raise NotImplementedError(mode)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        for i, v in enumerate((loss, loss_cls, loss_aux)):
            tot[i] += v.detach().item()
    return tuple(v / max(1, len(graphs)) for v in tot)


@torch.no_grad()
def ppi_test(model, graphs):
    """Micro-F1 over all nodes and labels of ``graphs`` (/root/reference/ppi_pyg/gnn.py:277-288: predictions = logits > 0,
    sklearn ``f1_score(average='micro')``, 0 when nothing is predicted positive)."""
    model.eval()
    tp = fp = fn = 0.0
    for g in graphs:
        pred = (model(g.x, g.edge_index) > 0).float()
        y = g.y
        tp += float((pred * y).sum())
        fp += float((pred * (1 - y)).sum())
        fn += float(((1 - pred) * y).sum())
    if tp + fp == 0:
        return 0
    return 2 * tp / (2 * tp + fp + fn)


# ------------------------------------------------------------------------------------------------
# the arxiv GAT teacher that PRODUCES the artefacts of the hot path
# ------------------------------------------------------------------------------------------------
class ElementWiseLinear(nn.Module):
    """arxiv_dgl/models.py:11-45 (only the bias-only, in-place form the GAT uses)."""

    def __init__(self, size, weight=True, bias=True, inplace=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(size)) if weight else None
        self.bias = nn.Parameter(torch.zeros(size)) if bias else None
        self.inplace = inplace

    def forward(self, x):
        if self.weight is not None:
            x = x * self.weight
        if self.bias is not None:
            x = x + self.bias
        return x


class ArxivGAT(nn.Module):
    """arxiv_dgl/models.py:239-313: [GATConv(residual) -> flatten heads -> BatchNorm1d -> activation -> dropout] x (L-1), a
    one-head... (n_heads of the hidden layers, 1 head for the last), mean over the heads of the last layer, ``bias_last``.
    ``self.feat`` = the last hidden features (what ``gat.py:250-251`` saves as the teacher's feature artefact)."""

    def __init__(self, in_feats, n_classes, n_hidden, n_layers, n_heads, activation, dropout=0.0, input_drop=0.0, attn_drop=0.0,
                 edge_drop=0.0, use_attn_dst=True, use_symmetric_norm=False):
        super().__init__()
        from .nn import DGLGATConv
        self.in_feats, self.n_hidden, self.n_classes, self.n_layers, self.num_heads = in_feats, n_hidden, n_classes, n_layers, n_heads
        self.convs, self.norms = nn.ModuleList(), nn.ModuleList()
        for i in range(n_layers):
            in_hidden = n_heads * n_hidden if i > 0 else in_feats
            out_hidden = n_hidden if i < n_layers - 1 else n_classes
            num_heads = n_heads if i < n_layers - 1 else 1
            self.convs.append(DGLGATConv(in_hidden, out_hidden, num_heads=num_heads, attn_drop=attn_drop, edge_drop=edge_drop,
                                         use_attn_dst=use_attn_dst, use_symmetric_norm=use_symmetric_norm, residual=True))
            if i < n_layers - 1:
                self.norms.append(nn.BatchNorm1d(n_heads * out_hidden))
        self.bias_last = ElementWiseLinear(n_classes, weight=False, bias=True, inplace=True)
        self.input_drop, self.dropout, self.activation = nn.Dropout(input_drop), nn.Dropout(dropout), activation
        self.feat = None

    def forward(self, graph, feat):
        h = self.input_drop(feat)
        for i in range(self.n_layers):
            h = self.convs[i](graph, h)
            if i < self.n_layers - 1:
                h = h.flatten(1)
                h = self.norms[i](h)
                h = self.activation(h)
                h = self.dropout(h)
                self.feat = h
        return self.bias_last(h.mean(1))


def add_labels(feat, labels, idx, n_classes):
    """gat.py:104-107: one-hot labels of ``idx`` appended to the features (zeros elsewhere)."""
    onehot = torch.zeros([feat.shape[0], n_classes], dtype=feat.dtype, device=feat.device)
    onehot[idx, labels[idx, 0]] = 1
    return torch.cat([feat, onehot], dim=-1)


@torch.no_grad()
def teacher_evaluate(model, graph, feat, labels, train_idx, val_idx, test_idx, n_classes, use_labels=True, n_label_iters=0):
    """The producer of the teacher artefacts, gat.py:151-183 (``evaluate``): eval-mode forward with the train labels as input
    features and ``n_label_iters`` label-reuse rounds (predicted soft labels written back for the unlabelled nodes).
    Returns (pred [N, C] = what ``logits/<expt>/<seed>.pt`` holds, model.feat [N, heads*hidden] = ``features/...``)."""
    model.eval()
    if use_labels:
        feat = add_labels(feat, labels, train_idx, n_classes)
    pred = model(graph, feat)
    if n_label_iters > 0:
        unlabel_idx = torch.cat([val_idx, test_idx])
        for _ in range(n_label_iters):
            feat[unlabel_idx, -n_classes:] = F.softmax(pred[unlabel_idx], dim=-1)
            pred = model(graph, feat)
    return pred, model.feat
