"""Student models and the train / eval step, mirroring the reference's L1/L2 code on the new operators.

``GCN`` / ``SAGE`` / ``ProjectionGCD`` follow /root/reference/arxiv_pyg/gnn.py:23-99 (same constructor
signatures, ``convs`` / ``bns`` state_dict keys, ``.out_feat`` side channel); ``train_step`` follows
``train()`` (:102-195) and the KD+aux rule of gnn_kd_and_aux.py:114-181; ``evaluate`` follows ``test()``
(:198-218).  The reference's own ``gnn.py`` also runs unchanged on top of ``efficient-gnns_amd/dropin``.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _cache, _lib
from . import criterion as C
from . import ops
from .nn import DGLGATConv, GATConv, GCNConv, RGCNConv, SAGEConv
from .sparse import SparseTensor


_EVAL_BN_FOLD = True   # (no environment switch any more; tests flip the attribute to compare the fold with the separate normalisation pass)
# Fusions measured in round 3 (profiles/r03_bench_line_r02_paths.json: all five off = 130.5 vs 144.3 epochs/s); module attributes, no
# environment switches -- tests flip them to compare each fused form with its composed one.
_TRAIN_ROWS = True        # [train_idx] row picks inside the CE / KD kernels
_LSP_FULL_ROWS = True     # LSP on the full tensors through composed edge ids
_FUSED_TAIL = True        # ops.bn_act_linear for the last hidden layer of a GCN
_SAMPLED_HEADS = True     # projection heads form only the rows a sampled criterion keeps


class _Student(nn.Module):
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

    # ---- the layer plan ---------------------------------------------------------------------------------------------------
    # Every hidden layer is conv -> BatchNorm -> ReLU -> dropout (gnn.py:47-50).  WHICH kernels run it depends on five facts that
    # hold for a whole forward pass (mode, grad mode, device, adjacency kind, module types); ``_layer_form`` names the pair
    # (conv form, activation form) of a layer and ``forward`` only executes it:
    #   conv form   "fold"   test(): eval-mode BatchNorm folded into the conv's weights, ReLU in the last kernel's store (no act.)
    #               "stats"  training GCNConv on a plain adjacency: BatchNorm statistics from the aggregation's store epilogue
    #               "plain"  conv(x, adj_t)
    #   act. form   "tail"       last hidden layer of a GCN: BN + ReLU + dropout + gradient tap + the output conv's h @ W as ONE op
    #               "tail_sync"  the same on node-range shards (dist.SyncBatchNorm1d: all-rank statistics)
    #               "bn_act"     fused BN + ReLU + dropout kernels (torch.nn.BatchNorm1d)
    #               "sync"       dist.SyncBatchNorm1d.fused_act
    #               "torch"      another norm module (CPU tensors: the gloo tests' stand-in switch only)
    def _layer_form(self, li, x, adj_t):
        conv, bn, last = self.convs[li], self.bns[li], self.convs[-1]
        training, grad, gpu = self.training, torch.is_grad_enabled(), x.is_cuda
        plain_adj, sharded_adj = isinstance(adj_t, SparseTensor), hasattr(adj_t, "gcn_normalized")
        gcn, torch_bn, sync_bn = isinstance(conv, GCNConv), isinstance(bn, nn.BatchNorm1d), hasattr(bn, "fused_act")
        if (_EVAL_BN_FOLD and not training and not grad and gcn and gpu and not conv._uses_memoised_input(x)
                and ((type(bn) is nn.BatchNorm1d and bn.track_running_stats and plain_adj) or (sharded_adj and sync_bn))):
            return "fold", None
        conv_form = "stats" if (gcn and torch_bn and gpu and training and plain_adj) else "plain"
        act = "bn_act" if (torch_bn and gpu) else ("sync" if sync_bn else "torch")
        if (li == len(self.bns) - 1 and _FUSED_TAIL and training and grad and gpu and isinstance(last, GCNConv)
                and last.in_channels >= last.out_channels):
            if plain_adj and not sharded_adj and type(bn) is nn.BatchNorm1d and last.out_channels <= 64 and last.in_channels % 64 == 0:
                act = "tail"
            elif sharded_adj and hasattr(bn, "fused_act_linear"):
                act = "tail_sync"
        return conv_form, act

    def forward(self, x, adj_t):
        for li, (conv, bn) in enumerate(zip(self.convs[:-1], self.bns)):
            conv_form, act = self._layer_form(li, x, adj_t)
            if conv_form == "fold":
                x = self.out_feat = conv(x, adj_t, eval_bn=bn)
                continue
            x = conv(x, adj_t, bn_stats_shift=bn.running_mean, want_bn_stats=True) if conv_form == "stats" else conv(x, adj_t)
            if act in ("tail", "tail_sync"):
                # (h, h @ W_out) in one op whose backward is one pass over the [N, hidden] tensors (ops._BnActLinear); None = the fused
                # kernels do not take this shape: the composed form below
                both = (ops.bn_act_linear(x, bn, self.convs[-1].weight, relu=True, p=self.dropout, training=True) if act == "tail"
                        else bn.fused_act_linear(x, self.convs[-1].weight, True, self.dropout, True))
                if both is not None:
                    self.out_feat, xw = both
                    return self.convs[-1](self.out_feat, adj_t, xw=xw)
                act = "bn_act" if act == "tail" else "sync"
            if act == "bn_act":
                x = ops.bn_act(x, bn, relu=True, p=self.dropout, training=self.training)
            elif act == "sync":
                x = bn.fused_act(x, True, self.dropout, self.training)
            else:                      # another norm module: torch's own operators (on the GPU)
                _lib.require_gpu(x)
                x = F.dropout(F.relu(bn(x)), p=self.dropout, training=self.training)
            self.out_feat = x
        if self.training and x.is_cuda:
            # the last hidden state has two consumers (the last conv and student_proj(out_feat[train_idx]), gnn.py:150): the projection's
            # row-compact input gradient is added into the conv's dense one instead of autograd's zero-fill + scatter + full-size add
            x = self.out_feat = ops.grad_tap(x)
        return self.convs[-1](x, adj_t)


class GCN(_Student):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, cached=True):
        super().__init__(lambda a, b: GCNConv(a, b, cached=cached), in_channels, hidden_channels, out_channels,
                         num_layers, dropout)


class SAGE(_Student):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, aggr="mean"):
        super().__init__(lambda a, b: SAGEConv(a, b, aggr=aggr), in_channels, hidden_channels, out_channels,
                         num_layers, dropout)


class ProjectionGCD(nn.Module):
    def __init__(self, hidden_channels, proj_dim):
        super().__init__()
        self.lin = nn.Linear(hidden_channels, proj_dim)
        self.conv = GCNConv(hidden_channels, proj_dim)
        self.bn = nn.BatchNorm1d(proj_dim)

    def forward(self, x, adj_t):
        return F.relu(self.bn(self.lin(x) + self.conv(x, adj_t)))


class ProjectionHead(nn.Sequential):
    """Linear + BatchNorm1d + ReLU (gnn.py:296-306) with the reference's Sequential state_dict keys (0.*, 1.*);
    the forward runs the MFMA GEMM and the fused BN + ReLU kernels."""

    def __init__(self, in_dim, proj_dim):
        super().__init__(nn.Linear(in_dim, proj_dim), nn.BatchNorm1d(proj_dim), nn.ReLU())

    def forward(self, x):
        lin, bn = self[0], self[1]
        if hasattr(bn, "fused_act"):                            # dist.SyncBatchNorm1d
            return bn.fused_act(ops.linear(x, lin.weight, lin.bias), True, 0.0, self.training)
        _lib.require_gpu(x)
        if not isinstance(bn, nn.BatchNorm1d):
            return super().forward(x)
        return ops.bn_act(ops.linear(x, lin.weight, lin.bias), bn, relu=True, p=0.0, training=self.training)

    def forward_rows(self, x, idx, pick=None, const_input=False):
        """``self(x[idx])`` for unique row ids: the gather is fused into the GEMM's operand load (gnn.py:150-156).
        ``pick`` (unique ids into ``idx``): ``self(x[idx])[pick]`` -- what a sampled criterion keeps of the head's output
        (criterion.py:62-65,134-137); the BatchNorm statistics span all of ``idx``, only the picked rows are normalised and stored."""
        lin, bn = self[0], self[1]
        y = ops.linear_rows(x, idx, lin.weight, lin.bias, const_input=const_input)   # const_input: see ops.linear_rows (the teacher head)
        if hasattr(bn, "fused_act"):                            # dist.SyncBatchNorm1d: all-rank statistics, only the picked rows formed
            return bn.fused_act(y, True, 0.0, self.training, pick=pick)
        if not isinstance(bn, nn.BatchNorm1d):
            y = self[2](bn(y))
            return y if pick is None else y[pick]
        return ops.bn_act(y, bn, relu=True, p=0.0, training=self.training, pick=pick)


def make_projection(in_dim, proj_dim):
    return ProjectionHead(in_dim, proj_dim)


_ROW_CACHE = _cache.TensorKeyedCache(capacity=8)
_CACHE_CONST_ROWS = os.environ.get("EGNN_CACHE_CONST_ROWS", "0") == "1"  # opt-in (the reference re-gathers every step)


def _const_rows(t, idx):
    """``t[idx]`` for a constant tensor (teacher artefacts): the reference re-gathers 273 MB every step (gnn.py:155);
    the rows do not change within a run, so (opt-in) the gather is done once per (tensor, index) identity + version."""
    if t.requires_grad or not _CACHE_CONST_ROWS:
        return t[idx]
    return _ROW_CACHE.get((t, idx), (), lambda: t[idx])


_GLOBAL_EDGES = _cache.TensorKeyedCache(capacity=8)


def _global_edges(edge_index, train_idx):
    """``train_idx[edge_index]``: the edge list of the train-induced subgraph (relabelled, gnn.py:274) in node ids of the full graph.
    Built once per (edge list, index) identity + version; the entry keeps both alive (_cache.py)."""
    return _GLOBAL_EDGES.get((edge_index, train_idx), (), lambda: train_idx[edge_index].contiguous())


def distill_loss(mode, model, out, labels, train_idx, teacher_out_feat, teacher_logits, hp,
                 student_proj=None, teacher_proj=None, edge_index=None, adj_t=None, kd_and_aux=False, rows=None):
    """``rows`` = None: ``out`` / ``labels`` are the compact train rows (gnn.py:109-110).  ``rows`` = train_idx: they are the FULL
    logits / labels and the criteria pick the rows inside their kernels (criterion.py's ``rows`` keyword)."""
    kd_teacher = (lambda: teacher_logits) if rows is not None else (lambda: _const_rows(teacher_logits, train_idx))
    if mode == "supervised":
        loss = ops.cross_entropy(out, labels, rows)
        return loss, loss, loss * 0
    if mode == "kd":
        return C.rows_kd_criterion(out, labels, kd_teacher(), hp["alpha"], hp["kd_T"], rows=rows)
    picked = False
    if mode in ("fitnet", "gpw", "nce"):
        if hasattr(student_proj, "forward_rows") and hasattr(teacher_proj, "forward_rows") and not _CACHE_CONST_ROWS:
            pick = None
            if mode in ("gpw", "nce") and _SAMPLED_HEADS:
                # the criterion's one host draw (criterion.py:62-65,134-137), made here: the heads then form only the sampled rows
                pick = C._sample_rows(train_idx.numel(), hp["max_samples"], model.out_feat.device)
                picked = pick is not None
            f = student_proj.forward_rows(model.out_feat, train_idx, pick=pick)   # proj(feat[train_idx]) without the copies
            t = teacher_proj.forward_rows(teacher_out_feat, train_idx, pick=pick, const_input=True)   # the teacher's features never change (gnn.py:155)
        else:
            f = student_proj(ops.take_rows(model.out_feat, train_idx))
            t = teacher_proj(_const_rows(teacher_out_feat, train_idx))
    elif mode == "lpw" and _LSP_FULL_ROWS and edge_index is not None:
        # LSP reads rows only through the edge list (criterion.py:100-104): the train-subgraph ids of gnn.py:274 are composed with
        # train_idx once, and the edge kernels then address the FULL feature tensors -- feat[train_idx] / teacher_feat[train_idx] (366 MB of
        # copies per step) and the zero-fill + scatter of their backward never exist; the loss is a mean over the same edges.
        f, t, edge_index = model.out_feat, teacher_out_feat, _global_edges(edge_index, train_idx)
    elif mode in ("at", "lpw"):
        f, t = ops.take_rows(model.out_feat, train_idx), _const_rows(teacher_out_feat, train_idx)
    elif mode == "gcd":
        f = student_proj(model.out_feat, adj_t)[train_idx]
        t = teacher_proj(teacher_out_feat, adj_t)[train_idx]
    else:
        raise NotImplementedError(mode)
    if mode == "fitnet":
        res = C.rows_fitnet_criterion(out, labels, f, t, hp["beta"], rows=rows)
    elif mode == "at":
        res = C.rows_at_criterion(out, labels, f, t, hp["beta"], rows=rows)
    elif mode == "gpw":
        res = C.rows_gpw_criterion(out, labels, f, t, hp["kernel"], hp["beta"], hp["max_samples"], rows=rows, presampled=picked)
    elif mode == "lpw":
        res = C.rows_lpw_criterion(out, labels, f, t, edge_index, hp["kernel"], hp["beta"], rows=rows)
    else:
        res = C.rows_nce_criterion(out, labels, f, t, hp["beta"], hp["nce_T"], hp["max_samples"], rows=rows, presampled=picked)
    if not kd_and_aux:
        return res
    loss_aux = res[2]
    loss, loss_cls, _ = C.rows_kd_criterion(out, labels, kd_teacher(), hp["alpha"], hp["kd_T"], rows=rows)
    return loss + hp["beta"] * loss_aux, loss_cls, loss_aux


_ONES: dict = {}


def _one_like(t):
    """A cached 0-dim 1.0 on ``t``'s device: ``loss.backward(gradient=...)`` without autograd's per-step ones_like fill launch.  Never
    evicted (4 bytes per device; captured graphs read it through a raw pointer)."""
    key = (str(t.device), t.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=t.dtype, device=t.device)
    return one


def train_step_tensors(model, x, adj_t, y, train_idx, optimizer, mode, hp, teacher_out_feat=None, teacher_logits=None,
                       student_proj=None, teacher_proj=None, edge_index=None, kd_and_aux=False, out=None):
    """One full-graph optimisation step (gnn.py:102-195) WITHOUT the host reads: returns the device tensor
    [loss, loss_cls, loss_aux] (written into ``out`` -- float32 [3] -- when given).  (``train_step`` adds the reads; ``GraphedEpoch``
    captures this function.)"""
    model.train()
    for p in (student_proj, teacher_proj):
        if p is not None:
            p.train()
    logits = model(x, adj_t)
    if _TRAIN_ROWS and logits.is_cuda and y.dim() == 2 and y.shape[1] == 1 and y.dtype == torch.int64:
        # gnn.py:109-110 `out = model(...)[train_idx]`, `y.squeeze(1)[train_idx]` (and `teacher_logits[train_idx]`): the row picks
        # happen inside the CE / KD kernels (the criteria's `rows` keyword), their backward writes the dense logits gradient
        loss, loss_cls, loss_aux = distill_loss(mode, model, logits, y.view(-1), train_idx, teacher_out_feat, teacher_logits, hp,
                                                student_proj, teacher_proj, edge_index, adj_t, kd_and_aux, rows=train_idx)
    else:
        out = ops.take_rows(logits, train_idx)   # == model(...)[train_idx] (split ids are unique)
        labels = y.squeeze(1)[train_idx]
        loss, loss_cls, loss_aux = distill_loss(mode, model, out, labels, train_idx, teacher_out_feat, teacher_logits, hp,
                                                student_proj, teacher_proj, edge_index, adj_t, kd_and_aux)
    optimizer.zero_grad()
    loss.backward(gradient=_one_like(loss) if loss.is_cuda and loss.dim() == 0 else None)
    optimizer.step()
    return torch.stack([loss.detach(), loss_cls.detach(), loss_aux.detach()], out=out)


def train_step(model, x, adj_t, y, train_idx, optimizer, mode, hp, teacher_out_feat=None, teacher_logits=None,
               student_proj=None, teacher_proj=None, edge_index=None, kd_and_aux=False):
    """One full-graph optimisation step (= one training epoch of the reference); the three losses as Python floats (the
    reference's three ``.item()`` reads, here one device->host copy)."""
    vals = train_step_tensors(model, x, adj_t, y, train_idx, optimizer, mode, hp, teacher_out_feat, teacher_logits,
                              student_proj, teacher_proj, edge_index, kd_and_aux)
    a, b, c = vals.tolist()
    return a, b, c


def accuracy(y_true, y_pred) -> float:
    """``ogb`` Evaluator('ogbn-arxiv')['acc'] = mean(y_true == y_pred); compared on the device, one scalar read."""
    return float((y_true == y_pred).float().mean())


@torch.no_grad()
def evaluate_tensors(model, x, adj_t, y, split_idx, accs_out=None):
    """``test()`` (gnn.py:198-218) without the host read: (logits, device tensor of the three accuracies -- ``accs_out``, float64 [3],
    when given)."""
    model.eval()
    out = model(x, adj_t)
    if y.dtype == torch.int64 and y.numel() == out.shape[0]:
        return out, ops.split_accuracy(out, y, split_idx, out=accs_out)   # argmax + the three Evaluator accuracies in one pass
    y_pred = out.argmax(dim=-1, keepdim=True)
    hit = (y_pred == y).view(-1).to(torch.float32)
    return out, torch.stack([hit[split_idx[k]].mean() for k in ("train", "valid", "test")])


@torch.no_grad()
def evaluate(model, x, adj_t, y, split_idx):
    _lib.require_gpu(x)
    out, accs = evaluate_tensors(model, x, adj_t, y, split_idx)   # the three Evaluator accuracies with ONE device->host read instead of three
    return out, tuple(accs.tolist())


class GraphedEpoch:
    """One epoch of the reference loop (gnn.py:333-340: ``train()`` then ``test()``) captured ONCE as a hipGraph and
    replayed: the ~180 kernel launches of an epoch are enqueued by one call, the GPU never waits for the host between
    them.  Same kernels, same arithmetic, same RNG coupling as ``train_step`` + ``evaluate``:
      * the G-CRD / GSP row sample is still ONE ``np.random.choice`` per step on the host (criterion.py:63,135); it is
        uploaded into a static device buffer the captured kernels read (``criterion._ROW_SAMPLER``);
      * dropout masks change every replay: the kernels add a per-step seed that lives in device memory
        (``ops._DROPOUT_SEED_DEV``), drawn from torch's host generator like the eager path;
      * losses / accuracies are read back once per epoch from static output tensors.
    Needs ``torch.optim.Adam(..., capturable=True)`` (fused or not).  ``split_idx=None`` captures the train step only."""

    def __init__(self, model, x, adj_t, y, train_idx, optimizer, mode, hp, teacher_out_feat=None, teacher_logits=None,
                 student_proj=None, teacher_proj=None, edge_index=None, split_idx=None, kd_and_aux=False, warmup=3):
        if not x.is_cuda:
            raise ValueError("GraphedEpoch needs GPU tensors")
        self.mode, self.hp = mode, hp
        self.n_train = train_idx.numel()
        S = hp.get("max_samples", 0) if mode in ("nce", "gpw") else 0
        self.n_pick = S if 0 < S < self.n_train else 0
        dev = x.device
        self._pick_dev = torch.zeros(max(self.n_pick, 1), dtype=torch.int64, device=dev)
        self._pick_host = torch.zeros(max(self.n_pick, 1), dtype=torch.int64).pin_memory()
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._seed_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        args = (model, x, adj_t, y, train_idx, optimizer, mode, hp, teacher_out_feat, teacher_logits, student_proj, teacher_proj,
                edge_index, kd_and_aux)

        # the epoch's six scalars side by side in ONE 40-byte device buffer (three float32 losses at byte 0, three float64 accuracies at
        # byte 16), written by the kernels that produce them: one device->host copy per epoch, no gather launch
        self._res_bytes = torch.zeros(40, dtype=torch.uint8, device=dev)
        res_losses, res_accs = self._res_bytes[:12].view(torch.float32), self._res_bytes[16:].view(torch.float64)

        def body():
            losses = train_step_tensors(*args, out=res_losses)
            if split_idx is None:
                return losses, None, None
            out, accs = evaluate_tensors(model, x, adj_t, y, split_idx, accs_out=res_accs)
            return losses, out, accs
        self._body = body
        # every cached structure the captured launches read through raw pointers (edge plans, composed edge lists, inverse row maps,
        # normalised adjacencies) stays alive with this object, whatever the caches evict later (_cache.pinning)
        self._pin_ctx = _cache.pinning()
        self._pinned = self._pin_ctx.__enter__()
        try:
            self._capture(body, dev, warmup)
        finally:
            self._pin_ctx.__exit__(None, None, None)

    def _capture(self, body, dev, warmup):
        # warm-up on a side stream (allocator, optimizer state, cached structures), as graph capture requires
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), self._installed():
            for i in range(warmup):
                self._refresh()
                if i == warmup - 1:
                    # the step about to be captured must not contain a long torch reduction (see _audit.py: their semaphore memset
                    # node was seen not to take effect in replays -- outputs silently stale)
                    from ._audit import CaptureAudit
                    with CaptureAudit() as audit:
                        body()
                    audit.check("GraphedEpoch")
                else:
                    body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # The graph is KEPT after capture and read back through the HIP runtime before it is instantiated: the shipped epochs are chains
        # of kernel nodes only.  A memset / memcpy / host node is what an ATen operator with hidden scratch traffic leaves behind (the
        # semaphore memset of a long reduction, sort / index_add_ / bincount scratch, zero_() on some paths) -- the class of node that was
        # seen not to take effect in replays (_audit.py).  Structural and always on: it also catches operators CaptureAudit's name
        # list does not know.
        from ._audit import check_captured_graph
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        with self._installed():
            self._refresh()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(self.graph):
                self.losses, self.out, self.accs = body()
        self.node_kinds = check_captured_graph(self.graph, "GraphedEpoch", kernels_only=True)
        self.graph.instantiate()
        torch.cuda.synchronize(dev)
        self._refresh()                                        # randomness of the first replay

    class _Install:
        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            self.prev = (C._ROW_SAMPLER, ops._DROPOUT_SEED_DEV)
            o = self.o

            def sampler(n, S, device):
                if S >= n:
                    return None
                if n != o.n_train or S != o.n_pick:
                    raise RuntimeError(f"GraphedEpoch was captured for {o.n_pick} of {o.n_train} rows, the criterion asks for {S} of {n}")
                return o._pick_dev
            C._ROW_SAMPLER = sampler
            ops._DROPOUT_SEED_DEV = o._seed_dev

        def __exit__(self, *exc):
            C._ROW_SAMPLER, ops._DROPOUT_SEED_DEV = self.prev

    def _installed(self):
        return GraphedEpoch._Install(self)

    def _draw(self):
        """The per-step host randomness (the reference's one ``np.random.choice`` per step, the dropout seed), drawn into
        pinned staging buffers."""
        import numpy as np
        if getattr(self, "_uploaded", None) is not None:
            self._uploaded.synchronize()     # the previous upload has read the pinned buffers (it sits in front of the replay: microseconds)
        if self.n_pick:
            self._pick_host.copy_(torch.from_numpy(np.random.choice(self.n_train, self.n_pick, replace=False)))
        self._seed_host.random_()
        self._seed_host.bitwise_and_(0x3FFFFFFFFFFFFFFF)

    def _upload(self):
        """Staging buffers -> the static device buffers the captured kernels read (stream-ordered: after the last replay)."""
        if self.n_pick:
            self._pick_dev.copy_(self._pick_host, non_blocking=True)
        self._seed_dev.copy_(self._seed_host, non_blocking=True)
        # only stream-ordered: ``_draw`` must not rewrite the pinned buffers before the DMA has read them
        self._uploaded = torch.cuda.Event()
        self._uploaded.record()

    def _refresh(self):
        self._draw()
        self._upload()

    def redraw(self):
        """Discard the randomness prepared for the next replay and draw it again (after re-seeding NumPy / torch)."""
        torch.cuda.current_stream().synchronize()
        self._refresh()

    # -- the loop without an idle GPU between epochs --------------------------------------------------------------------------------
    # ``step()`` reads the epoch's values right after its replay: the GPU idles from the end of replay k until the host has woken up,
    # uploaded the next draw and launched replay k + 1 (measured on the headline workload: 0.3-0.4 ms of a 7.2 ms epoch).  Nothing in
    # replay k + 1 depends on the VALUES of epoch k (the row sample and the dropout seed are host draws; gnn.py:333-340 only logs the
    # losses and accuracies), so ``step_async()`` launches replay k first and reads the values of epoch k - 1 afterwards: same
    # replays, same draws in the same order, every epoch's values still read by the host -- one call later.
    def _slots(self):
        if getattr(self, "_res_host", None) is None:
            self._res_host = [torch.zeros(40, dtype=torch.uint8).pin_memory() for _ in range(2)]
            self._res_done = [None, None]
            self._pending = None          # slot of the epoch whose values have not been handed out yet
            self._k = 0
            self.replay_events = None     # set to a list to collect (start, end) HIP events around every replay (bench.py)
        return self._res_host

    def _decode(self, host_bytes):
        losses = tuple(host_bytes[:12].view(torch.float32).tolist())
        return (losses, None) if self.accs is None else (losses, tuple(host_bytes[16:].view(torch.float64).tolist()))

    def _values(self, slot):
        self._res_done[slot].synchronize()
        return self._decode(self._res_host[slot])

    def step_async(self):
        """Launch one epoch and return the values of the PREVIOUS ``step_async`` epoch (None on the first call): the randomness of this
        replay was drawn during the previous one and is uploaded stream-ordered in front of it; its losses / accuracies travel to a
        pinned host slot behind it.  ``drain()`` hands out the values of the last epoch launched."""
        self._slots()
        slot = self._k & 1
        self._k += 1
        if self.replay_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.graph.replay()
        if self.replay_events is not None:
            e1.record()
            self.replay_events.append((e0, e1))
        self._res_host[slot].copy_(self._res_bytes, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._res_done[slot] = done
        self._draw()                      # host draw of the NEXT epoch, overlapped with this replay
        self._upload()                    # stream-ordered behind this replay: the static buffers change after it has read them
        prev, self._pending = self._pending, slot
        return None if prev is None else self._values(prev)

    def drain(self):
        """Values of the last epoch launched by ``step_async`` (None if they were handed out already)."""
        prev, self._pending = getattr(self, "_pending", None), None
        return None if prev is None else self._values(prev)

    def step(self):
        """Replay one epoch; returns ((loss, loss_cls, loss_aux), (train, valid, test accuracies) | None).  The host draw for
        the NEXT step (np.random.choice of 16 384 of 90 941 rows costs ~1 ms) runs while this replay executes."""
        if getattr(self, "_pending", None) is not None:
            raise RuntimeError("GraphedEpoch.step() after step_async(): call drain() first (an epoch's values are still in flight)")
        self.graph.replay()
        self._draw()                                            # overlapped with the replay; uploaded after the read below
        vals = self._decode(self._res_bytes.cpu())              # one device->host read per epoch
        self._upload()
        return vals


class GAT(nn.Module):
    """The PPI GAT teacher (/root/reference/ppi_pyg/gnn.py:86-117), run frozen inside the student step (:208-209):
    GATConv + linear skip per layer, ELU, dropout; the last layer averages its heads.  Inference only (nn.GATConv)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, heads=4):
        super().__init__()
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
            x = conv(x, adj_t) + ops.linear(x, lin.weight, lin.bias)
            x = F.elu(x)
            x = F.dropout(x, p=self.dropout, training=self.training)
            self.out_feat = x
        return self.convs[-1](x, adj_t) + ops.linear(x, self.lins[-1].weight, self.lins[-1].bias)


class ElementWiseLinear(nn.Module):
    """/root/reference/arxiv_dgl/models.py:11-45 (state_dict keys ``weight`` / ``bias``)."""

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
    """The arxiv GAT teacher (/root/reference/arxiv_dgl/models.py:239-313; ``gat.py`` builds it 3 layers x 250 x 3 heads,
    expt ``gat-3L250x3h``) for inference on the kernels -- same attribute names (``convs``, ``norms``, ``bias_last``), so the
    reference's ``checkpoints/<expt>/<seed>.pt['model_state_dict']`` loads.  ``self.feat`` = the last hidden features: the
    [N, 750] tensor the student reads as ``teacher_out_feat`` (arxiv_pyg/gnn.py:278)."""

    def __init__(self, in_feats, n_classes, n_hidden, n_layers, n_heads, activation, dropout=0.0, input_drop=0.0, attn_drop=0.0,
                 edge_drop=0.0, use_attn_dst=True, use_symmetric_norm=False):
        super().__init__()
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
                bn = self.norms[i]
                if h.is_cuda and not self.training and ops.bn_shape_ok(h) and getattr(self.activation, "__name__", "") == "relu":
                    h = ops.bn_act(h, bn, relu=True, p=0.0, training=False)      # BatchNorm (running statistics) + ReLU fused
                else:
                    h = self.dropout(self.activation(bn(h)))
                self.feat = h
        return self.bias_last(h.mean(1))


def add_labels(feat, labels, idx, n_classes):
    """gat.py:104-107: one-hot labels of ``idx`` appended to the features (zeros elsewhere)."""
    onehot = torch.zeros([feat.shape[0], n_classes], dtype=feat.dtype, device=feat.device)
    onehot[idx, labels[idx, 0]] = 1
    return torch.cat([feat, onehot], dim=-1)


@torch.no_grad()
def teacher_evaluate(model, graph, feat, labels, train_idx, val_idx, test_idx, n_classes, use_labels=True, n_label_iters=0):
    """The producer of the teacher artefacts (gat.py:151-183 ``evaluate``): eval-mode forward with the train labels as input
    features and ``n_label_iters`` label-reuse rounds (soft predictions written back for the unlabelled nodes, :162-166).
    Returns (pred [N, C] -> ``logits/<expt>/<seed>.pt``, model.feat [N, heads * hidden] -> ``features/<expt>/<seed>.pt``;
    ``data.save_teacher_artifacts`` writes them in the reference's layout, gat.py:243-251)."""
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


class TeacherNet(nn.Module):
    """The PPI teacher checkpointed by the reference (/root/reference/ppi_pyg/gnn.py:23-47): 4 x 256 GAT layers with
    linear skips, ELU, a 6-head averaging output layer; ``out_feat`` = second hidden.  Same attribute names (state_dict
    keys ``conv1.*``, ``lin1.*``, ...) so that the reference's ``checkpoint.pt`` loads."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
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
        x = F.elu(self.conv1(x, edge_index) + ops.linear(x, self.lin1.weight, self.lin1.bias))
        x = F.elu(self.conv2(x, edge_index) + ops.linear(x, self.lin2.weight, self.lin2.bias))
        self.out_feat = x
        return self.conv3(x, edge_index) + ops.linear(x, self.lin3.weight, self.lin3.bias)


class RGCN(nn.Module):
    """/root/reference/mag_pyg/gnn.py:71-168 on the kernels: ``forward`` on a (sampled) grouped subgraph through
    ``nn.RGCNConv``; ``inference`` full-batch, one rectangular mean-SpMM (``adj_t.matmul(x, reduce='mean')``, :151,162) and
    one GEMM per relation and layer.  The GraphSAINT sampler that feeds ``forward`` in the reference is out of scope."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, num_nodes_dict, x_types, num_edge_types):
        super().__init__()
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

    @torch.no_grad()
    def inference(self, x_dict, edge_index_dict, key2int):
        x_dict = dict(x_dict)
        for key, emb in self.emb_dict.items():
            x_dict[int(key)] = emb
        adj_t_dict = {}
        for key, (row, col) in edge_index_dict.items():
            n_dst, n_src = x_dict[key2int[key[-1]]].shape[0], x_dict[key2int[key[0]]].shape[0]
            adj_t_dict[key] = SparseTensor(row=col, col=row, sparse_sizes=(n_dst, n_src))   # unsorted constructor (mag:151)
        for i, conv in enumerate(self.convs):
            out_dict = {j: ops.linear(x, conv.root_lins[j].weight, conv.root_lins[j].bias) for j, x in x_dict.items()}
            for keys, adj_t in adj_t_dict.items():
                tmp = adj_t.matmul(x_dict[key2int[keys[0]]], reduce="mean")
                tgt = key2int[keys[-1]]
                out_dict[tgt] = out_dict[tgt] + ops.linear(tmp, conv.rel_lins[key2int[keys]].weight)
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
