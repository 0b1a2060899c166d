#!/usr/bin/env python3
"""bench.py -- training epochs/sec of the BASELINE.json workload on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One "step" = one epoch of the reference loop /root/reference/arxiv_pyg/gnn.py:333-340: one full-graph
``train()`` step (forward + G-CRD loss + backward + Adam + 3x .item()) PLUS one full-graph ``test()``
(eval forward + argmax + 3 accuracies), on the synthetic ogbn-arxiv-shaped graph of SURVEY.md 8(d)
(config 2: 3-layer GCN student, hidden 256, G-CRD with the run_gcn.sh:140-145 hyper-parameters
beta=0.1, nce_T=0.075, max_samples=16384, proj_dim=256).  Inputs are resident in HBM before timing.

Rank 0 prints ONE JSON line (contract in the task statement) carrying also
  "roofline":     the SpMM aggregate kernel (K=256 GCN layer) timed with HIP events on its launch stream; with the epoch
                  replayed as a hipGraph (default) the timed region has no per-kernel events, so the brackets are taken on
                  --probe-epochs EAGER epochs of the same problem right after it (over the timed region itself with --graph off);
                  achieved = algorithmic bytes / avg launch time
  "roofline_mfma": (extra) the G-CRD entry points -- the largest share of the step, matrix-pipe-bound -- timed the same way;
                  achieved = 6 S^2 P flops per step / their time
  "cpu_baseline": the CPU oracle (pure-PyTorch restatement of the reference path) timed on the host cores
                  on the SAME synthetic inputs, a bounded sample of epochs (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling

HP = dict(alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=16384, proj_dim=256, kernel="rbf")
MODEL = dict(hidden=256, layers=3, dropout=0.5, lr=0.01)
# per-loss hyper-parameters of record for the secondary configs (scripts/run_gcn.sh:52-145, run_sage.sh)
MODE_HP = {"lpw": dict(beta=100.0, kernel="rbf"), "gpw": dict(beta=100.0, kernel="cosine", max_samples=4096, proj_dim=128),
           "kd": dict(alpha=0.9, kd_T=4.0)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed epochs (default: a timed region of ~1.4 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.0, help="graph size multiplier (1.0 = ogbn-arxiv shape)")
    ap.add_argument("--max-samples", type=int, default=HP["max_samples"])
    ap.add_argument("--gnn", default="gcn", choices=["gcn", "sage"])
    ap.add_argument("--training", default="nce", choices=["nce", "kd", "gpw", "lpw", "supervised"])
    ap.add_argument("--kernel", default=None, choices=["cosine", "poly", "l2", "rbf"],
                    help="similarity kernel of the LSP / GSP losses (default: the value of record of the mode, MODE_HP)")
    ap.add_argument("--cpu-epochs", type=int, default=10, help="CPU-oracle epochs timed for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="untimed CPU-oracle warm-up epochs (BASELINE.md section 3: >= 3)")
    ap.add_argument("--reference-epochs", type=int, default=20,
                    help="epochs of the reference's unmodified train()/test() loop timed through dropin/ for the reference_loop object (0 = skip)")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size GPU-vs-oracle parity step")
    ap.add_argument("--parity-trajectory-steps", type=int, default=3,
                    help="replayed steps with dropout 0.5 compared against the oracle with the same masks injected (0 = skip)")
    ap.add_argument("--repeat-blocks", type=int, default=-1,
                    help="further blocks of --steps replays timed after the contract's block (spread of the measurement; 0 = none; "
                         "default: as many as fill ~2 s, at least 4, at most 50)")
    ap.add_argument("--settle-seconds", type=float, default=5.0,
                    help="untimed device-settle phase in front of the --warmup steps: replays of the timed program for this many seconds "
                         "(reported in `timing.settle_before_warmup` with its block times; 0 = none)")
    ap.add_argument("--graph", default=None, choices=["on", "off", "auto"],
                    help="on: the epoch (train step + eval) is captured once as a hipGraph and replayed (models.GraphedEpoch) -- a "
                         "capture failure ends the run with a non-zero exit code; off: eager launches; auto: replay if the capture "
                         "succeeds, eager launches otherwise (the `launch` field says which).  Default: on for the single-GPU path; auto for "
                         "the sharded path (all ranks agree on the outcome; a multi-rank capture has not met real multi-GPU hardware yet, "
                         "and a line with eager launches is worth more than no line)")
    ap.add_argument("--no-local-roofline", action="store_true", help="skip the second roofline object (aggregation on the reordered community graph)")
    ap.add_argument("--probe-epochs", type=int, default=5, help="eager epochs with per-kernel HIP-event brackets for the roofline objects")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--workload", default="arxiv", choices=["arxiv", "mag"],
                    help="arxiv: BASELINE.json configs[1] (the headline); mag: configs[4], the MAG-shaped SAGE-mean + KD run on "
                         "node-range shards (sharded code path; with --gpus 1 use --force-sharded)")
    ap.add_argument("--partition", default="auto", choices=["auto", "range"],
                    help="sharded runs: auto = cut the node ranges from the community order when that lowers the halo; range = node ids as given")
    ap.add_argument("--graph-kind", default="chunglu", choices=["chunglu", "local", "local-sorted"],
                    help="synthetic graph of the sharded runs: chunglu (headline: no locality) | local (community structure, ids shuffled)")
    ap.add_argument("--agg", default="auto", choices=["auto", "halo", "sliced"],
                    help="sharded runs: how an aggregation gets its remote operand rows -- halo (the referenced rows travel), sliced (the feature "
                         "columns are re-sharded around the aggregation: bytes independent of the halo), auto (per adjacency and width, whichever "
                         "moves fewer bytes; dist.ShardedAdj.sliced_pays)")
    ap.add_argument("--one-device", action="store_true",
                    help="all ranks of a multi-rank launch share cuda:0 and their collectives travel over gloo through host memory "
                         "(efficient-gnns_amd/hostcomm.py): a FUNCTIONAL run of the sharded step on the real kernels with a non-empty halo "
                         "on a one-GPU box (eager launches; its epochs/s is not a scaling measurement)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: the sharded path's HOST LOGIC over gloo (launcher, partition plan, collectives, the JSON line) with the tests' "
                         "stand-ins for the kernels (tests/bench_cpu_harness.py); without them every operator raises on a CPU tensor")
    ap.add_argument("--hidden", type=int, default=0, help="sharded runs: hidden width (default: the model of record, 256)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the node-range sharded (multi-GPU) code path even with one rank: RCCL init, halo plan, "
                         "SyncBN, row-block G-CRD -- a 1-GPU check of the path the N>1 runs take")
    return ap.parse_args()


def seed_all(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_problem(M, data, device, args, hp, dropout=None):
    """Model + projection heads + Adam exactly as gnn.py:251-315 builds them."""
    Net = M.GCN if args.gnn == "gcn" else M.SAGE
    model = Net(data.num_features, MODEL["hidden"], data.num_classes, MODEL["layers"],
                MODEL["dropout"] if dropout is None else dropout).to(device)
    sp = tp = None
    groups = [{"params": model.parameters(), "lr": MODEL["lr"]}]
    if args.training in ("nce", "gpw"):
        sp = M.make_projection(MODEL["hidden"], hp["proj_dim"]).to(device)
        tp = M.make_projection(data.teacher_out_feat.shape[1], hp["proj_dim"]).to(device)
        groups += [{"params": sp.parameters(), "lr": MODEL["lr"]}, {"params": tp.parameters(), "lr": MODEL["lr"]}]
    # the single-kernel (fused) implementation of the same torch.optim.Adam update (gnn.py:308-312)
    on_gpu = torch.device(device).type == "cuda"
    if on_gpu:
        # the reference's three groups carry the same hyper-parameters (gnn.py:308-312: lr for all of them), so one group is the
        # same update; the fused optimizer launches once per GROUP (~42 us each, latency-bound on 0.5 M parameters)
        groups = [{"params": [p for g in groups for p in g["params"]], "lr": MODEL["lr"]}]
    opt = torch.optim.Adam(groups, fused=on_gpu, capturable=on_gpu)
    return model, sp, tp, opt


def to_device(data, device):
    import types
    d = types.SimpleNamespace(**vars(data))
    d.x, d.y = data.x.to(device), data.y.to(device)
    d.adj_t = data.adj_t.to(device)
    d.split_idx = {k: v.to(device) for k, v in data.split_idx.items()}
    import efficient_gnns_amd.ops as _ops
    d.teacher_out_feat = _ops.pad_pitch(data.teacher_out_feat.to(device))   # same values, rows 16-byte aligned (750 -> pitch 752)
    d.teacher_logits = data.teacher_logits.to(device)
    return d


def epoch(M, model, d, opt, args, hp, sp, tp, edge_index):
    losses = M.train_step(model, d.x, d.adj_t, d.y, d.split_idx["train"], opt, args.training, hp,
                          d.teacher_out_feat, d.teacher_logits, sp, tp, edge_index)
    _, accs = M.evaluate(model, d.x, d.adj_t, d.y, d.split_idx)
    return losses, accs


class SpmmProbe:
    """Brackets every egnn_spmm_csr_f32 launch of the timed region with HIP events on the launch stream."""

    def __init__(self, ops):
        self.ops, self.records, self.active = ops, [], False
        self._orig = ops.spmm_raw

    def __enter__(self):
        def wrapped(adj, x, *args, **kw):
            if not self.active:
                return self._orig(adj, x, *args, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self._orig(adj, x, *args, **kw)
            b.record()
            self.records.append((x.shape[1], adj.nnz(), adj.spmm_algorithmic_bytes(x.shape[1]), a, b))
            return out
        self.ops.spmm_raw = wrapped
        return self

    def __exit__(self, *exc):
        self.ops.spmm_raw = self._orig

    def summary(self, K):
        ts = [(a.elapsed_time(b) * 1e-3, nbytes) for k, _, nbytes, a, b in self.records if k == K]
        if not ts:
            return None
        avg = sum(t for t, _ in ts) / len(ts)
        return dict(launches=len(ts), avg_s=avg, bytes=ts[0][1])


class NceProbe:
    """Brackets the G-CRD entry points (egnn_nce_fwd_f32 / egnn_nce_bwd_f32: the largest share of the step, MFMA-bound) with
    HIP events on the launch stream, for the secondary `roofline_mfma` object."""

    def __init__(self, lib):
        self.lib, self.records, self.active = lib, [], False
        self._orig = {n: getattr(lib, n) for n in ("egnn_nce_fwd_f32", "egnn_nce_bwd_f32")}

    def __enter__(self):
        def wrap(name, flops_per_s2p):
            orig = self._orig[name]

            def wrapped(*a):
                if not self.active:
                    return orig(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = orig(*a)
                e1.record()
                S, P = int(a[2]), int(a[3])
                self.records.append((flops_per_s2p * S * S * P, e0, e1))
                return rc
            return wrapped
        setattr(self.lib, "egnn_nce_fwd_f32", wrap("egnn_nce_fwd_f32", 2))   # Z = F T^T
        setattr(self.lib, "egnn_nce_bwd_f32", wrap("egnn_nce_bwd_f32", 4))   # dF = P T, dT = P^T F
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(self.lib, n, f)

    def summary(self):
        if not self.records:
            return None
        flops = sum(f for f, _, _ in self.records)
        secs = sum(a.elapsed_time(b) * 1e-3 for _, a, b in self.records)
        return dict(calls=len(self.records), flops=flops, secs=secs)


class GemmProbe:
    """Brackets every dense-GEMM entry point of the step that is NOT part of the G-CRD loss -- egnn_gemm_f32 / _ex / _add / _rows: the
    layer transforms x W, their input gradients, the transposed weight-gradient products and the projection heads
    (arxiv_pyg/gnn.py:47,296-306) -- with HIP events on the launch stream, for the `roofline_gemm` object.  A call's flops are
    2 M N K; calls with a class-count-wide dimension (min(M, N, K) <= 64: the HBM-bound skinny kernels of csrc/gemm_skinny.hip) are
    reported separately and are not part of the matrix-pipe fraction."""
    NAMES = ("egnn_gemm_f32", "egnn_gemm_ex_f32", "egnn_gemm_add_f32", "egnn_gemm_rows_f32", "egnn_gemm_tn_planes_f32", "egnn_gemm_rows_planes_f32")

    def __init__(self, lib):
        self.lib, self.records, self.active = lib, [], False
        self._orig = {n: getattr(lib, n) for n in self.NAMES}

    def __enter__(self):
        def wrap(name):
            orig = self._orig[name]

            def wrapped(*a):
                if not self.active:
                    return orig(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = orig(*a)
                e1.record()
                if rc != 0:                                # a refused shape / pipeline (EGNN_EALIGN: the caller takes another entry point)
                    return rc
                if name == "egnn_gemm_tn_planes_f32":      # C = A^T B[rows], B cut once into planes: (M, N, K, ...)
                    self.records.append((1, 0, int(a[0]), int(a[1]), int(a[2]), "+rows(planes)", e0, e1))
                elif name == "egnn_gemm_rows_planes_f32":  # C = A[rows] B^T, A cut once into planes
                    self.records.append((0, 1, int(a[0]), int(a[1]), int(a[2]), "+rows(planes)", e0, e1))
                else:
                    self.records.append((int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), "+rows" if name.endswith("rows_f32") else "", e0, e1))
                return rc
            return wrapped
        for n in self.NAMES:
            setattr(self.lib, n, wrap(n))
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(self.lib, n, f)

    def summary(self):
        if not self.records:
            return None
        wide = [r for r in self.records if min(r[2], r[3], r[4]) > 64]
        skinny = [r for r in self.records if min(r[2], r[3], r[4]) <= 64]
        tot = lambda rs: (sum(2.0 * r[2] * r[3] * r[4] for r in rs), sum(r[6].elapsed_time(r[7]) * 1e-3 for r in rs))  # noqa: E731
        shapes = {}
        for r in wide:
            key = f"{'T' if r[0] else 'N'}{'T' if r[1] else 'N'}{r[5]} {r[2]}x{r[3]}x{r[4]}"
            f, t, c = shapes.get(key, (0.0, 0.0, 0))
            shapes[key] = (f + 2.0 * r[2] * r[3] * r[4], t + r[6].elapsed_time(r[7]) * 1e-3, c + 1)
        return dict(wide=tot(wide), skinny=tot(skinny), n_wide=len(wide), n_skinny=len(skinny), shapes=shapes)


class GspProbe:
    """Brackets the GSP forward entry point (egnn_gsp_fwd_f32: the student and teacher Gram tiles + the squared difference, f32-input
    MFMA) with HIP events on the launch stream, for the `roofline_gsp` object of `--training gpw`."""

    def __init__(self, lib):
        self.lib, self.records, self.active = lib, [], False
        self._orig = lib.egnn_gsp_fwd_f32

    def __enter__(self):
        orig = self._orig

        def wrapped(*a):
            if not self.active:
                return orig(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig(*a)
            e1.record()
            Ps, Pt, S = int(a[2]), int(a[5]), int(a[6])
            self.records.append((2.0 * S * S * (Ps + Pt), S, Ps, Pt, e0, e1))      # 2 S^2 (P_s + P_t)
            return rc
        self.lib.egnn_gsp_fwd_f32 = wrapped
        return self

    def __exit__(self, *exc):
        self.lib.egnn_gsp_fwd_f32 = self._orig

    def summary(self):
        if not self.records:
            return None
        flops = sum(r[0] for r in self.records)
        secs = sum(r[4].elapsed_time(r[5]) * 1e-3 for r in self.records)
        return dict(calls=len(self.records), flops=flops, secs=secs, S=self.records[0][1], Ps=self.records[0][2], Pt=self.records[0][3])


class EdgeProbe:
    """Brackets the LSP forward entry points (egnn_edge_sim_f32 x 2 + egnn_lsp_loss_fwd_f32: per-edge similarities of gathered rows and the
    segment-softmax criterion, HBM / L2-bound gather work) with HIP events on the launch stream, for the `roofline_edges` object."""
    NAMES = ("egnn_edge_sim_f32", "egnn_lsp_loss_fwd_f32")

    def __init__(self, lib):
        self.lib, self.records, self.active = lib, [], False
        self._orig = {n: getattr(lib, n) for n in self.NAMES}

    def __enter__(self):
        def wrap(name):
            orig = self._orig[name]

            def wrapped(*a):
                if not self.active:
                    return orig(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = orig(*a)
                e1.record()
                self.records.append((name, int(a[2]) if name == "egnn_edge_sim_f32" else 0, e0, e1))
                return rc
            return wrapped
        for n in self.NAMES:
            setattr(self.lib, n, wrap(n))
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(self.lib, n, f)

    def summary(self):
        if not self.records:
            return None
        secs = sum(a.elapsed_time(b) * 1e-3 for _, _, a, b in self.records)
        steps = sum(1 for r in self.records if r[0] == "egnn_lsp_loss_fwd_f32")
        widths = sorted({r[1] for r in self.records if r[0] == "egnn_edge_sim_f32"})
        return dict(secs=secs, steps=max(steps, 1), widths=widths)


def cpu_baseline(args, data, hp):
    """The CPU oracle on the host cores: same inputs, same epoch definition, bounded number of epochs."""
    import oracle.models as OM
    import oracle.sparse as OS
    import types
    rowptr, col, _ = data.adj_t.csr()
    d = types.SimpleNamespace(**vars(data))
    d.adj_t = OS.SparseTensor(rowptr=rowptr, col=col, sparse_sizes=data.adj_t.sparse_sizes())
    seed_all(args.seed)
    model, sp, tp, opt = build_problem(OM, d, "cpu", args, hp)
    times = []
    for i in range(args.cpu_warmup + args.cpu_epochs):
        t0 = time.perf_counter()
        OM.train_step(model, d.x, d.adj_t, d.y, d.split_idx["train"], opt, args.training, hp,
                      d.teacher_out_feat, d.teacher_logits, sp, tp, None)
        OM.evaluate(model, d.x, d.adj_t, d.y, d.split_idx)
        times.append(time.perf_counter() - t0)
    timed = times[args.cpu_warmup:]  # warm-up epochs: gcn_norm caching, allocator, thread pool
    med = float(np.median(timed))
    return dict(value=1.0 / med, unit="epochs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(timed)} epochs after {args.cpu_warmup} warm-up, same synthetic graph/seeds, median {med:.3f} s/epoch "
                       f"(pure-PyTorch CPU oracle; the reference's PyG stack is not installable)")


# SURVEY.md 8(c) bars for ONE optimisation step at full size: losses rtol 1e-5 (G-CRD / GSP: 2e-5, the 1/tau resp. the
# difference of two Gram matrices amplify operand rounding), eval logits 1e-5 of max|ref|, parameter gradients rtol 1e-4
# (+ 2e-5 of the tensor's max|ref|: fp32 sums over up to 169 343 rows in a different order)
PARITY_BARS = dict(loss_rtol=1e-5, loss_rtol_pairwise=2e-5, logits=1e-5, grad_rtol=1e-4, grad_atol_over_max=2e-5)


def feeds_batchnorm(name, n_layers):
    """A bias that is added right in front of a BatchNorm (conv biases of the hidden layers, the Linear bias of a projection
    head): BatchNorm subtracts the column mean, so its gradient is mathematically zero and what either implementation holds
    is rounding noise -- not a measurement."""
    if name in ("student_proj.0.bias", "teacher_proj.0.bias"):
        return True
    return any(name in (f"model.convs.{i}.bias", f"model.convs.{i}.lin_l.bias") for i in range(n_layers - 1))


def grad_errors(got, ref32, ref64, n_layers=MODEL["layers"]):
    """Per-parameter gradient comparison after ONE step (the gradients are still in .grad: zero_grad() runs before backward).

    Reference = a FLOAT64 run of the CPU oracle.  At full size two correct fp32 implementations cannot agree to 1e-4 on the
    gradients upstream of a ReLU: of the 43 M pre-activations of a hidden layer a handful lie within fp32 rounding of zero, the
    two runs put them on different sides, and each such flip moves a weight gradient (a sum over 169 343 rows) by ~1 / sqrt(N) =
    2e-3 of its size (tools/checks/fullsize_ops_probe.py and tests/test_gpu_full_size.py::test_relu_mask_flips_... show the
    mechanism: with the SAME mask the layer's gradients agree to 1e-6).  The fp32 CPU oracle itself is 1e-4 .. 5e-3 away from
    its float64 run on those tensors.  The bar is therefore: per parameter tensor, |got - ref64| <= 1e-4 |ref64| + 2e-5
    max|ref64| OR max|got - ref64| <= 1.5 x max|oracle_fp32 - ref64| -- not farther from the truth than the
    reference's own fp32 CPU path.  Biases in front of a BatchNorm (true gradient zero, see feeds_batchnorm) are skipped.
    Returns (worst violation ratio, worst tensor, max error of `got` and of the fp32 oracle over max|ref64|)."""
    worst, worst_name, rel_got, rel_32 = 0.0, None, 0.0, 0.0
    for k, r in ref64.items():
        scale = float(r.abs().max())
        if feeds_batchnorm(k, n_layers) or scale == 0.0 or k not in got:
            continue
        e_got, e_32 = (got[k] - r).abs(), (ref32[k] - r).abs()
        bound = PARITY_BARS["grad_rtol"] * r.abs() + PARITY_BARS["grad_atol_over_max"] * scale
        ratio = min(float((e_got / bound).max()), float(e_got.max()) / max(1.5 * float(e_32.max()), 1e-300))
        rel_got, rel_32 = max(rel_got, float(e_got.max()) / scale), max(rel_32, float(e_32.max()) / scale)
        if ratio > worst:
            worst, worst_name = ratio, k
    return worst, worst_name, rel_got, rel_32


def _named_grads(model, sp, tp):
    named = [(f"model.{k}", v) for k, v in model.named_parameters()]
    for tag, m in (("student_proj", sp), ("teacher_proj", tp)):
        if m is not None:
            named += [(f"{tag}.{k}", v) for k, v in m.named_parameters()]
    return {k: v.grad.detach().double().cpu() for k, v in named if v.grad is not None}


def parity_check(args, data, d, device, hp, PM):
    """Full-size parity inside the driver-observed run: ONE optimisation step of the timed configuration (same graph,
    N = 169 343, S = max_samples, same np.random draw, same initial weights) on the GPU path and on the CPU oracle, with
    dropout = 0 (the dropout masks of the two implementations are only equal in distribution), plus the eval logits of
    the initial state.  Bars: PARITY_BARS (gnn.py:102-195, criterion.py:57-149); gradients against a float64 run of the
    oracle (see grad_errors)."""
    import copy
    import oracle.models as OM
    import oracle.sparse as OS
    import types
    rowptr, col, _ = data.adj_t.csr()
    dc = types.SimpleNamespace(**vars(data))
    dc.adj_t = OS.SparseTensor(rowptr=rowptr, col=col, sparse_sizes=data.adj_t.sparse_sizes())
    seed_all(args.seed + 17)
    om, osp, otp, oopt = build_problem(OM, dc, "cpu", args, hp, dropout=0.0)
    pm, psp, ptp, popt = build_problem(PM, d, device, args, hp, dropout=0.0)
    pm.load_state_dict(om.state_dict())
    for a, b in ((psp, osp), (ptp, otp)):
        if a is not None:
            a.load_state_dict(b.state_dict())
    # the float64 twin of the oracle (same initial weights; built before any forward so that no fp32 A^ is cached in it)
    om64, osp64, otp64 = (copy.deepcopy(m).double() if m is not None else None for m in (om, osp, otp))
    edge_o = edge_p = None
    if args.training == "lpw":
        import oracle.utils as OU
        from efficient_gnns_amd.utils import subgraph
        edge_o = OU.subgraph(dc.split_idx["train"], torch.stack(dc.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=dc.num_nodes)[0]
        edge_p = subgraph(d.split_idx["train"], torch.stack(d.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
    out_o, accs_o = OM.evaluate(om, dc.x, dc.adj_t, dc.y, dc.split_idx)
    out_p, accs_p = PM.evaluate(pm, d.x, d.adj_t, d.y, d.split_idx)
    logit_err = float((out_p.cpu() - out_o).abs().max() / out_o.abs().max().clamp_min(1e-30))
    np.random.seed(args.seed + 17)
    ref = OM.train_step(om, dc.x, dc.adj_t, dc.y, dc.split_idx["train"], oopt, args.training, hp, dc.teacher_out_feat,
                        dc.teacher_logits, osp, otp, edge_o)
    np.random.seed(args.seed + 17)
    got = PM.train_step(pm, d.x, d.adj_t, d.y, d.split_idx["train"], popt, args.training, hp, d.teacher_out_feat,
                        d.teacher_logits, psp, ptp, edge_p)
    groups64 = [{"params": m.parameters(), "lr": MODEL["lr"]} for m in (om64, osp64, otp64) if m is not None]
    fill32 = OS.SparseTensor.fill_value
    OS.SparseTensor.fill_value = lambda self, fill, dtype=torch.float64: fill32(self, fill, dtype)   # A^ values in double as well
    try:
        np.random.seed(args.seed + 17)
        ref64 = OM.train_step(om64, dc.x.double(), dc.adj_t, dc.y, dc.split_idx["train"], torch.optim.Adam(groups64), args.training, hp,
                              None if dc.teacher_out_feat is None else dc.teacher_out_feat.double(), dc.teacher_logits.double(), osp64, otp64, edge_o)
    finally:
        OS.SparseTensor.fill_value = fill32
    rtol = PARITY_BARS["loss_rtol_pairwise"] if args.training in ("nce", "gpw") else PARITY_BARS["loss_rtol"]
    # relative error of each of the three terms (a term that is exactly zero on the oracle -- `supervised` has no auxiliary
    # loss -- must be exactly zero here)
    rel = max((abs(a - b) / abs(b)) if b != 0 else (0.0 if a == 0 else float("inf")) for a, b in zip(got, ref))
    rel64 = max((abs(a - b) / abs(b)) if b != 0 else (0.0 if a == 0 else float("inf")) for a, b in zip(got, ref64))
    gworst, gname, grel, grel32 = grad_errors(_named_grads(pm, psp, ptp), _named_grads(om, osp, otp), _named_grads(om64, osp64, otp64))
    # a loss term passes at rtol against the fp32 oracle, or -- a term that is a small difference of large sums, e.g. the KL of two
    # nearly uniform edge distributions (LSP with the rbf kernel inside the train step: the student's similarities underflow, the
    # teacher's are all ~0.37; 1.9e-5 is what is left of O(1) terms over 680 k edges, condition number ~1e3: the fp32 oracle itself
    # is 7.5e-5 off its float64 run) -- when it is within 4 x the fp32 oracle's own distance from the float64 value
    loss_ok = all(abs(g - c) <= rtol * abs(c) or abs(g - t) <= max(rtol * abs(t), 4.0 * abs(c - t)) for g, c, t in zip(got, ref, ref64))
    ok = bool(loss_ok and logit_err <= PARITY_BARS["logits"] and gworst <= 1.0
              and all(abs(a - b) <= 1e-4 for a, b in zip(accs_p, accs_o)))
    traj = None
    if getattr(args, "parity_trajectory_steps", 0) > 0:
        traj = trajectory_dropout(args, data, d, device, hp, PM)
        ok = bool(ok and traj["ok"])
    return dict(ok=ok, trajectory_dropout=traj,
                what="first train step (dropout 0) + initial eval, GPU path vs CPU oracle, full size, same seeds/draw/weights",
                loss=dict(gpu=got[0], cpu=ref[0], cpu_f64=ref64[0]), loss_cls=dict(gpu=got[1], cpu=ref[1], cpu_f64=ref64[1]),
                loss_aux=dict(gpu=got[2], cpu=ref[2], cpu_f64=ref64[2]),
                max_rel_err=rel, max_rel_err_vs_f64=rel64, rtol=rtol, losses_ok=loss_ok,
                loss_bar="each term: |gpu - cpu| <= rtol |cpu|, or |gpu - f64| <= max(rtol |f64|, 4 |cpu - f64|)", eval_logits_max_abs_err_over_max_abs=logit_err, logits_tol=PARITY_BARS["logits"],
                grads=dict(reference="float64 run of the CPU oracle", gpu_max_abs_err_over_max_abs=grel, cpu_oracle_f32_max_abs_err_over_max_abs=grel32,
                           worst_violation_of_bar=round(gworst, 4), worst_tensor=gname,
                           bar=f"per parameter tensor: |gpu - f64| <= {PARITY_BARS['grad_rtol']} |f64| + {PARITY_BARS['grad_atol_over_max']} max|f64|, or "
                               "max|gpu - f64| <= 1.5 max|cpu_f32 - f64| (ReLU-mask flips at full size: bench.grad_errors)"),
                accs=dict(gpu=[round(a, 6) for a in accs_p], cpu=[round(a, 6) for a in accs_o]))


def trajectory_dropout(args, data, d, device, hp, PM):
    """The regime the benchmark times -- dropout 0.5, hipGraph replays, Adam moving -- against the oracle: the replays' dropout
    masks (counter hash of the recorded seeds, csrc/bn_common.h) are rebuilt on the host and injected into the oracle's F.dropout
    (oracle/training_parity.py; gnn.py:48-50), both sides start from the post-warm-up state and make the same np.random draws.
    Bar: every loss term of every step within 2e-4 (the golden trajectories' bar; GSP 1.5e-3, rbf-LSP 2e-3: differences of nearly
    equal O(1) sums).  The first replay is launched on an idle device -- the condition under which long torch reductions inside a
    replay returned stale values (profiles/r04_lsp_trace.txt)."""
    import efficient_gnns_amd.ops as ops
    import oracle.training_parity as TP
    from efficient_gnns_amd.utils import subgraph
    r = TP.trajectory(PM, ops, data, d, device, args.gnn, args.training, hp, steps=args.parity_trajectory_steps, graph=True,
                      subgraph_fn=subgraph, hidden=MODEL["hidden"], layers=MODEL["layers"], dropout=MODEL["dropout"], lr=MODEL["lr"],
                      seed=args.seed + 29, warmup=2, sync_before_first_replay=True)
    rtol = {"gpw": 1.5e-3, "lpw": 2e-3}.get(args.training, 2e-4)
    worst = 0.0
    for g, c in zip(r["got"], r["ref"]):
        for a, b in zip(g, c):
            worst = max(worst, abs(a - b) / max(abs(b), 1e-30) if b != 0 else (0.0 if a == 0 else float("inf")))
    return dict(ok=bool(worst <= rtol and all(np.isfinite(v) for g in r["got"] for v in g)), steps=len(r["got"]), dropout=MODEL["dropout"],
                launch="hipGraph replays (models.GraphedEpoch), first replay on an idle device", rtol=rtol, max_rel_err=worst,
                gpu=[[round(v, 6) for v in g] for g in r["got"]], cpu_oracle_same_masks=[[round(v, 6) for v in c] for c in r["ref"]])


def reference_loop(args, d, device, hp, epochs=20, warmup=5):
    """The reference's OWN loop on the kernels (north_star: "the existing train loops drop in unchanged"): arxiv_pyg/gnn.py's GCN /
    SAGE classes, ``train()`` and ``test()`` (gnn.py:23-99,102-218) -- the verbatim script staged under oracle/_ref/ by
    ``__graft_entry__.build()`` -- imported through efficient-gnns_amd/dropin and run with eager launches, as ``python gnn.py`` would.
    torch.nn.BatchNorm1d / F.relu / F.dropout / the three-group Adam stay PyTorch operators there (they are calls of the script's own
    model code, above the operator boundary); the package's loop (the headline ``value``) fuses them and replays a hipGraph.
    Outside the timed region, like ``cpu_baseline``."""
    import argparse as _ap
    import importlib.util
    import types
    ref_path = os.path.join(ROOT, "oracle", "_ref", "arxiv_pyg", "gnn.py")
    if not os.path.exists(ref_path):
        return dict(error="oracle/_ref/arxiv_pyg/gnn.py is not staged (run __graft_entry__.build() where /root/reference exists)")
    dropin = os.path.join(ROOT, "efficient-gnns_amd", "dropin")
    shimmed = ("criterion", "torch_geometric", "torch_geometric.nn", "torch_geometric.utils", "torch_geometric.transforms", "torch_sparse")
    stubs = ("ogb", "ogb.nodeproppred", "torch.utils.tensorboard", "logger")
    saved = {k: sys.modules.get(k) for k in shimmed + stubs}
    sys.dont_write_bytecode = True
    try:
        for name in stubs:   # dataset download / logging dependencies of the script, not on the hot path
            sys.modules[name] = types.ModuleType(name)
        sys.modules["ogb.nodeproppred"].PygNodePropPredDataset = None
        sys.modules["ogb.nodeproppred"].Evaluator = object
        sys.modules["torch.utils.tensorboard"].SummaryWriter = object
        sys.modules["logger"].Logger = object
        sys.path.insert(0, dropin)
        for k in shimmed:
            sys.modules.pop(k, None)
        spec = importlib.util.spec_from_file_location("ref_gnn_bench", ref_path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)

        class Evaluator:   # ogb.nodeproppred.Evaluator('ogbn-arxiv').eval: tensors -> NumPy, mean of equality
            def eval(self, dd):
                return {"acc": float((dd["y_true"].detach().cpu().numpy() == dd["y_pred"].detach().cpu().numpy()).mean())}
        data = types.SimpleNamespace(x=d.x, y=d.y, adj_t=d.adj_t)
        train_idx = d.split_idx["train"]
        edge_index = None
        if args.training == "lpw":
            from efficient_gnns_amd.utils import subgraph
            edge_index = subgraph(train_idx, torch.stack(d.adj_t.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
        ns = _ap.Namespace(training=args.training, **{k: hp[k] for k in ("alpha", "kd_T", "beta", "nce_T", "max_samples", "kernel")})
        ev = Evaluator()

        def build():
            """Model, heads and optimizer exactly as gnn.py:251-315 builds them (built per leg: with dropin/accel.py enabled the script's
            ``torch.optim.Adam(groups)`` call resolves to the fused implementation, as under launch.py)."""
            seed_all(args.seed)
            Net = ref.GCN if args.gnn == "gcn" else ref.SAGE
            model = Net(d.num_features, MODEL["hidden"], d.num_classes, MODEL["layers"], MODEL["dropout"]).to(device)
            sp = tp = None
            groups = [{"params": model.parameters(), "lr": MODEL["lr"]}]
            if args.training in ("nce", "gpw"):      # gnn.py:296-312
                sp = torch.nn.Sequential(torch.nn.Linear(MODEL["hidden"], hp["proj_dim"]), torch.nn.BatchNorm1d(hp["proj_dim"]), torch.nn.ReLU()).to(device)
                tp = torch.nn.Sequential(torch.nn.Linear(d.teacher_out_feat.shape[1], hp["proj_dim"]), torch.nn.BatchNorm1d(hp["proj_dim"]),
                                         torch.nn.ReLU()).to(device)
                groups += [{"params": sp.parameters(), "lr": MODEL["lr"]}, {"params": tp.parameters(), "lr": MODEL["lr"]}]
            return model, sp, tp, torch.optim.Adam(groups)

        def timed():
            model, sp, tp, opt = build()

            def one():
                losses = ref.train(model, data, train_idx, opt, ns, d.teacher_out_feat, d.teacher_logits, sp, tp, edge_index)
                return losses, ref.test(model, data, d.split_idx, ev)[1]
            for _ in range(warmup):
                one()
            best = None
            for _ in range(2):      # two blocks, the faster one: the first eager block of a leg still grows the allocator's pools (measured:
                torch.cuda.synchronize()                      # 12.7 vs 9.4 ms per epoch, tools/r06/refloop.py)
                t0 = time.perf_counter()
                for _ in range(epochs):
                    losses, accs = one()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                best = el if best is None or el < best else best
            return best, losses, accs
        # (1) as `launch.py --plain-torch-modules` runs it; (2) as `launch.py` runs it by default: torch.nn.BatchNorm1d / torch.nn.Linear of the
        # script's own model code re-pointed at the package's kernels (dropin/accel.py)
        el_plain, _, _ = timed()
        spec = importlib.util.spec_from_file_location("egnn_dropin_accel", os.path.join(dropin, "accel.py"))
        accel = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(accel)
        accel.LAZY = False
        accel.enable()
        try:
            el_eager, _, _ = timed()
        finally:
            accel.disable()
        accel.LAZY = True
        accel.enable()
        try:
            el, losses, accs = timed()
        finally:
            accel.disable()
        return dict(epochs_per_s=round(epochs / el, 3), ms_per_epoch=round(1e3 * el / epochs, 3), epochs=epochs, warmup=warmup,
                    plain_torch_modules=dict(epochs_per_s=round(epochs / el_plain, 3), ms_per_epoch=round(1e3 * el_plain / epochs, 3),
                                             what="launch.py --plain-torch-modules: torch.nn.BatchNorm1d / torch.nn.Linear on PyTorch's kernels"),
                    one_launch_per_torch_call=dict(epochs_per_s=round(epochs / el_eager, 3), ms_per_epoch=round(1e3 * el_eager / epochs, 3),
                                                   what="accel.LAZY = False (round 5's form): BatchNorm1d / Linear on the package's kernels, F.relu / F.dropout / "
                                                        "[train_idx] as separate torch launches"),
                    what="the reference's own arxiv_pyg/gnn.py train() + test() (verbatim script, staged under oracle/_ref) through "
                         "efficient-gnns_amd/dropin as dropin/launch.py runs it: eager launches; torch.nn.BatchNorm1d / torch.nn.Linear run on the "
                         "package's kernels, BatchNorm1d.forward returns a deferred activation (efficient_gnns_amd/lazy.py) that absorbs the script's "
                         "F.relu / F.dropout and is formed by its consumer in one launch (the next conv -- with its narrow h @ W --, a sampled "
                         "criterion as only the sampled rows, Linear(out_feat[train_idx]) as the gather-fused GEMM, the constant teacher_out_feat[train_idx] "
                         "gather deferred into the teacher head's planes GEMMs, inference convs folded with their BatchNorm); the script's "
                         "torch.optim.Adam(groups) resolves to the fused implementation (dropin/accel.py); faster of two blocks of `epochs`",
                    last_losses=[round(float(v), 5) for v in losses], last_accs=[round(float(a), 4) for a in accs])
    except Exception as e:  # noqa: BLE001  (a secondary leg must not take the headline line down)
        return dict(error=f"{type(e).__name__}: {str(e)[:300]}")
    finally:
        if dropin in sys.path:
            sys.path.remove(dropin)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("ref_gnn_bench", None)


def local_graph_roofline(args, device, ops):
    """Second roofline object: the same K = 256 aggregation on a graph WITH locality -- the synthetic community graph (same
    degree law as the headline graph, 75 % of a node's non-hub edges inside its community, node ids shuffled as real datasets
    come) after the product's reorder pass (sparse.community_order + SparseTensor.permute).  The headline graph (Chung-Lu)
    is the locality-free worst case; real citation graphs sit in between."""
    import efficient_gnns_amd as E
    import efficient_gnns_amd.data as D
    from efficient_gnns_amd.sparse import community_order
    d2 = D.arxiv_like(args.scale, seed=args.seed, with_teacher=False, graph="local")
    adj = d2.adj_t.to(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    perm = community_order(adj)
    adj_r = adj.permute(perm)
    torch.cuda.synchronize()
    reorder_s = time.perf_counter() - t0
    out = {}
    K = MODEL["hidden"]
    x = torch.randn(d2.num_nodes, K, device=device)
    for tag, a in (("shuffled_ids", adj), ("reordered", adj_r)):
        gn = E.gcn_norm(a)
        for _ in range(3):
            ops.spmm_raw(gn, x, "sum")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.spmm_raw(gn, x, "sum")
        e1.record()
        e1.synchronize()
        out[tag] = (e0.elapsed_time(e1) * 1e-3 / 20, gn.spmm_algorithmic_bytes(K))
    secs, nbytes = out["reordered"]
    gbs = nbytes / secs / 1e9
    traffic, traffic_note = measured_traffic("spmm_traffic_local.json")
    return dict(bound="hbm", kernel=f"spmm_blk_kernel + spmm_combine_kernel (egnn_spmm_csr_blk_f32 + combine, K={K}, reduce=sum)",
                graph="synthetic community graph (degree-corrected SBM, same degree law as the headline graph, mu = 0.25, node ids shuffled) "
                      "after the reorder pass (sparse.community_order + SparseTensor.permute)",
                achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                algorithmic_bytes_per_launch=nbytes, avg_launch_us=round(secs * 1e6, 2), launches_timed=20,
                avg_launch_us_before_reorder=round(out["shuffled_ids"][0] * 1e6, 2), reorder_seconds=round(reorder_s, 3),
                traffic=traffic, traffic_source=traffic_note)


def gather_ceiling(adj_gcn, K, device):
    """Driver-observed request-path ceiling of the aggregation: egnn_probe_gather_lines_f32 replays the
    gather stream of the K-wide aggregation on THIS graph -- the same nnz * K * 4 bytes of 128-byte lines, same slice <-> XCD
    binding -- with no reduction and no output, timed with HIP events on the launch stream; best of a few grid sizes."""
    from efficient_gnns_amd import _lib
    lib = _lib.load()
    _, col, bits = adj_gcn._index_arrays()
    if bits != 32 or K % 32:
        return None
    x = torch.randn(adj_gcn.sparse_sizes()[1], K, device=device)
    sink = torch.zeros(1, device=device)
    nnz = adj_gcn.nnz()
    best = None
    for bps, inflight in ((128, 8), (256, 8), (512, 8), (1024, 8), (256, 16), (512, 16), (1024, 4), (2048, 4)):
        def run():
            _lib.check(lib.egnn_probe_gather_lines_f32(_lib.ptr(x), x.stride(0), x.shape[0], K, _lib.ptr(col), nnz, bps, inflight,
                                                       _lib.ptr(sink), _lib.stream()), "egnn_probe_gather_lines_f32")
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        e1.synchronize()
        secs = e0.elapsed_time(e1) * 1e-3 / 10
        if best is None or secs < best[0]:
            best = (secs, bps, inflight)
    return dict(gathered_bytes=nnz * K * 4, secs=best[0], blocks_per_slice=best[1], loads_in_flight=best[2])


def lib_sha16() -> str:
    """Identity of the kernels the run uses: ``build.source_stamp()`` -- SHA-256[:16] over the compile flags and the library's sources
    (the binary embeds its build time, so a rebuild of the same sources changes its bytes but not this value).  The library is built
    from these sources by ``__graft_entry__.build()`` (stale objects are recompiled) before any bench run of the driver."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("egnn_build", os.path.join(ROOT, "efficient-gnns_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.source_stamp()


def measured_traffic(name):
    """L2-miss FABRIC bytes per aggregation call from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in their own rocprofv3 runs
    over the lab driver by tools/evidence.sh, calibrated on a copy of known size; profiles/<name>).  FETCH_SIZE counts the requests
    that leave L2 -- served by the 256 MiB Infinity Cache or by HBM alike (MI355X_MICROARCH.md, HBM section): for the arxiv-sized
    source matrix (173 MB, inside the cache) this is fabric traffic, not HBM traffic; for the MAG-sized one (993 MB) it is HBM
    (profiles/r04_gather_footprint.txt has the split).  The file records the
    stamp of the kernel sources it was measured on (lib_sha16): a file from other kernels is STALE and is not reported
    (traffic = null) -- the counters cannot be collected inside this process (rocprofv3 wraps the process it profiles)."""
    tj = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(tj):
        return None, f"profiles/{name} absent"
    try:
        with open(tj) as fh:
            j = json.load(fh)
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/{name} unreadable: {e}"
    have = lib_sha16()
    if j.get("lib_sha16") != have:
        return None, f"profiles/{name} was measured on another build (lib_sha16 {j.get('lib_sha16')}, this run {have}): stale, not reported"
    return float(j["hbm_bytes_per_call"]), (f"L2-miss fabric bytes (Infinity-Cache hits included; the 173 MB source matrix fits the 256 MiB cache): rocprofv3 --pmc "
                                            f"FETCH_SIZE / WRITE_SIZE of the same entry points on this very build (lib_sha16 {have}), separate passes, calibrated "
                                            f"on a 256 MiB copy (profiles/{name}; measured off-line by tools/evidence.sh, not in this run)")


def gpu_clock_state(device):
    """Best-effort read of the device's clocks / power while it is under load (sysfs of the amdgpu driver; no privileges needed): lets a
    reader tell box variance (clocks, power cap) from software differences between two runs.  None for what cannot be read."""
    import glob
    out = {}
    try:
        idx = torch.device(device).index or 0
        props = torch.cuda.get_device_properties(idx)
        out["name"], out["cus"] = props.name, props.multi_processor_count
        bus = getattr(props, "pci_bus_id", None)
        cards = sorted(glob.glob("/sys/class/drm/card*/device"))
        pick = None
        for c in cards:
            try:
                if bus is not None and f":{int(bus):02x}:" in os.path.realpath(c):
                    pick = c
            except (TypeError, ValueError):
                pass
        if pick is None:
            amd = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
            pick = amd[idx] if idx < len(amd) else (amd[0] if amd else None)
        if pick is None:
            return out or None

        def cur(name):
            try:
                for ln in open(os.path.join(pick, name)).read().splitlines():
                    if ln.strip().endswith("*"):
                        return ln.split(":", 1)[1].replace("*", "").strip()
            except OSError:
                return None
        out["sclk"], out["mclk"], out["fclk"] = cur("pp_dpm_sclk"), cur("pp_dpm_mclk"), cur("pp_dpm_fclk")
        for hw in glob.glob(os.path.join(pick, "hwmon", "hwmon*")):
            for key, fn in (("power_cap_w", "power1_cap"), ("power_avg_w", "power1_average"), ("power_input_w", "power1_input")):
                try:
                    out[key] = round(int(open(os.path.join(hw, fn)).read()) / 1e6, 1)
                except (OSError, ValueError):
                    pass
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {str(e)[:120]}"
    return out or None


def cap_cpu_threads(local_world: int = 1) -> int:
    """ATen sizes its OpenMP pool from the visible CPUs (256 on the GPU box) although the container's cgroup grants far
    fewer (cpu.max = 16 there): the idle workers spin, exhaust the CFS quota and the launching thread is throttled for
    tens of ms at a time.  Cap the pool at the quota (shared by the ranks of this node)."""
    limit = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            limit = min(limit, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    n = max(1, min(torch.get_num_threads(), limit // max(1, local_world)))
    torch.set_num_threads(n)
    return n


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher in front: re-run this very command line as N ranks of one node under
    ``torch.distributed.run`` (rendezvous on 127.0.0.1, a free port), one process per GPU; rank 0's JSON line goes to this
    process's stdout unchanged.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(os.environ.get("EGNN_BENCH_ENTRY", sys.argv[0])), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL across processes)
    print(f"bench.py: --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks (one process per GPU) and hand their exit code on -- a plain
        # `--gpus 8` must never run one GPU and print n_gpus 1
        raise SystemExit(spawn_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or leave WORLD_SIZE unset and "
                         f"let `python bench.py --gpus N` start them)")
    cap_cpu_threads(int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if args.one_device:
        local_rank = 0
    if args.device == "cpu":
        if not (world > 1 or args.force_sharded):
            raise SystemExit("bench.py: --device cpu exists for the host-logic harness of the sharded path only (tests/bench_cpu_harness.py)")
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    hp = dict(HP, max_samples=args.max_samples)
    if args.training in MODE_HP:
        hp.update(MODE_HP[args.training])
        if args.max_samples != HP["max_samples"]:   # an explicit --max-samples wins over the mode's value of record
            hp["max_samples"] = args.max_samples
    if args.kernel:
        hp["kernel"] = args.kernel
        if args.training == "gpw" and args.kernel in ("rbf", "l2") and args.max_samples == HP["max_samples"]:
            # run_gcn.sh:52-72: the reference caps S at 2048 for the kernels that need its [S,S,D] tensor (rbf: beta = 1e5)
            hp.update(max_samples=2048, beta=1e5 if args.kernel == "rbf" else 1.0)

    import efficient_gnns_amd  # noqa: F401  (fails loudly if libegnn_hip.so is missing)
    import efficient_gnns_amd.data as D
    import efficient_gnns_amd.models as PM
    import efficient_gnns_amd.ops as ops

    sharded_path = world > 1 or args.force_sharded or args.workload == "mag"
    if args.graph is None:
        args.graph = "auto" if sharded_path else "on"
    if world > 1 or args.force_sharded or args.workload == "mag":
        if "MASTER_PORT" not in os.environ:
            # one process started by hand (--force-sharded / --workload mag): any free port (a fixed one can collide with the lingering
            # socket of a run that has just ended; one run of a back-to-back series produced no line)
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        import efficient_gnns_amd.dist as dist_mod
        dist_mod._AGG_MODE = args.agg
        if args.one_device:
            from efficient_gnns_amd import hostcomm
            hostcomm.install()
            args.graph = "off"        # a host round trip cannot be captured into a hipGraph
        model_cfg = dict(MODEL, hidden=args.hidden) if args.hidden else MODEL
        dist_mod.bench_main(args, hp, model_cfg, rank, world, device, backend="gloo" if (args.one_device or args.device == "cpu") else "nccl")   # prints the line on rank 0; ends with barrier + destroy_process_group
        # leave without the interpreter / static-destructor teardown: communicator background threads have been seen racing it
        # (exit code -6 after all results were delivered) and the launcher reads every rank's exit code
        sys.stdout.flush()
        sys.stderr.flush()
        if os.environ.get("EGNN_BENCH_NORMAL_EXIT") == "1":   # profilers (rocprofv3) write their output in atexit handlers
            return
        os._exit(0)

    seed_all(args.seed)
    data = D.arxiv_like(args.scale, seed=args.seed)
    d = to_device(data, device)
    edge_index = None
    if args.training == "lpw":
        from efficient_gnns_amd.utils import subgraph
        ei = torch.stack(d.adj_t.coo()[:2])
        edge_index = subgraph(d.split_idx["train"], ei, relabel_nodes=True, num_nodes=d.num_nodes)[0]
    # (the CPU-bound legs -- parity against the oracle, the reference's own loop, the CPU baseline -- run AFTER the timed region:
    #  a device that has idled through tens of seconds of host work starts the timed replays in a low power state)
    seed_all(args.seed)
    model, sp, tp, opt = build_problem(PM, d, device, args, hp)

    graphed, graph_note = None, "off"
    if args.graph in ("on", "auto"):
        try:   # capture the epoch once (its constructor runs the warm-up steps)
            graphed = PM.GraphedEpoch(model, d.x, d.adj_t, d.y, d.split_idx["train"], opt, args.training, hp, d.teacher_out_feat,
                                      d.teacher_logits, sp, tp, edge_index, split_idx=d.split_idx, warmup=3)
            graph_note = ("hipGraph replay of train step + eval (models.GraphedEpoch: 3 eager steps before the capture, then the "
                          "--warmup untimed replays, then the timed replays)")
        except Exception as e:  # noqa: BLE001
            if args.graph == "on":   # the headline number is the replayed epoch: never silently time something else
                raise SystemExit(f"bench.py: hipGraph capture of the epoch failed ({type(e).__name__}: {str(e)[:300]}); "
                                 f"use --graph auto or --graph off to time eager launches")
            graph_note = f"capture failed, eager launches: {type(e).__name__}: {str(e)[:200]}"
            graphed = None
    if graphed is None:
        for _ in range(args.warmup):
            epoch(PM, model, d, opt, args, hp, sp, tp, edge_index)
    # keep generation-2 garbage collections (a ~40 ms walk over the ~170k objects the imports leave behind) out of the
    # training loop: freeze what survived set-up, as a long-running trainer would
    import gc
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    train_ms = eval_ms = 0.0
    block_ms: list = []
    from efficient_gnns_amd import _lib as _egnn_lib
    timing = None
    if graphed is not None:
        def block(n, sync_reads=False, mid=None):
            """n epochs: n replays and n host reads of (3 losses, 3 accuracies), bracketed by device syncs.  Default: step_async -- replay k
            is launched before the values of epoch k - 1 are read (models.GraphedEpoch); sync_reads: step(), the read-then-launch loop."""
            torch.cuda.synchronize()
            tb = time.perf_counter()
            if sync_reads:
                for _ in range(n):
                    vals = graphed.step()
            else:
                for _ in range(n):
                    graphed.step_async()
                if mid is not None:
                    mid()                      # (the last replay is still executing: a look at the device under load)
                vals = graphed.drain()         # the n-th read
            torch.cuda.synchronize()
            return time.perf_counter() - tb, vals
        # device settle (declared in the line): the GPU comes out of set-up (host-side graph generation, capture) in a low clock state and
        # its replay time keeps falling for ~2-3 s of load (r06: 6.70 -> 6.54 ms over fifteen 20-replay blocks) -- replays of the timed
        # program for --settle-seconds, block times recorded
        settle = dict(replays=0, seconds=0.0, block_ms_per_step=[])
        if args.settle_seconds > 0:
            ts = time.perf_counter()
            while time.perf_counter() - ts < args.settle_seconds:
                nb = max(args.steps, 10)
                el, _ = block(nb)
                settle["replays"] += nb
                settle["block_ms_per_step"].append(round(1e3 * el / nb, 3))
            settle["seconds"] = round(time.perf_counter() - ts, 3)
        block(args.warmup)                     # the W untimed warm-up steps of the contract: replays of the program that is timed
        graphed.replay_events = []
        elapsed, (losses, accs) = block(args.steps)        # THE timed region: exactly K epochs
        gpu_ms = sorted(a.elapsed_time(b) for a, b in graphed.replay_events)
        graphed.replay_events = None
        # the contract's block is the one above; further blocks of the same length show the spread of the measurement
        n_blocks = args.repeat_blocks if args.repeat_blocks >= 0 else max(4, min(50, int(np.ceil(2.0 / max(elapsed, 1e-3)))))
        for _ in range(n_blocks):
            block_ms.append(block(args.steps)[0] * 1e3 / args.steps)
        clocks = {}
        block(args.steps, mid=lambda: clocks.update(gpu_clock_state(device) or {}))    # (not one of the reported blocks)
        sync_ms = [block(args.steps, sync_reads=True)[0] * 1e3 / args.steps for _ in range(2)]
        timing = dict(
            gpu_ms_per_replay=round(float(np.median(gpu_ms)), 4), gpu_ms_per_replay_min_max=[round(gpu_ms[0], 4), round(gpu_ms[-1], 4)],
            host_gap_ms_per_step=round(1e3 * elapsed / args.steps - float(np.median(gpu_ms)), 4),
            what="gpu_ms_per_replay: HIP events around graph.replay() of every timed epoch (median); host_gap = ms_per_step - that",
            loop="step_async: replay k is launched before the values of epoch k-1 are read (K replays, K host reads of 3 losses + 3 "
                 "accuracies inside the timed region; same replays / draws / values as the read-then-launch loop: "
                 "test_graphed_epoch_step_async_hands_out_the_same_values_one_call_later)",
            ms_per_step_read_then_launch=[round(v, 3) for v in sync_ms],
            settle_before_warmup=dict(settle, why="untimed replays of the timed program for --settle-seconds in front of the --warmup steps: the device "
                                                  "leaves set-up in a low clock state and its replay time keeps falling for ~2-3 s of load"),
            device_clocks_under_load=clocks)
    # per-kernel brackets (HIP events on the launch stream) need eager launches: with a graph the timed region above has no
    # per-kernel events, so the roofline objects are measured on `probe_epochs` eager epochs of the same problem right after
    # it; without a graph they are measured over the timed region itself
    n_probe = args.probe_epochs if graphed is not None else args.steps
    with SpmmProbe(ops) as probe, NceProbe(_egnn_lib.load()) as nce_probe, EdgeProbe(_egnn_lib.load()) as edge_probe, \
            GspProbe(_egnn_lib.load()) as gsp_probe, GemmProbe(_egnn_lib.load()) as gemm_probe:
        probe.active = True
        gemm_probe.active = True
        nce_probe.active = True
        edge_probe.active = args.training == "lpw"
        gsp_probe.active = args.training == "gpw"
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_probe):
            ev[0].record()
            l2 = PM.train_step(model, d.x, d.adj_t, d.y, d.split_idx["train"], opt, args.training, hp,
                               d.teacher_out_feat, d.teacher_logits, sp, tp, edge_index)
            ev[1].record()
            _, a2 = PM.evaluate(model, d.x, d.adj_t, d.y, d.split_idx)
            ev[2].record()
            ev[2].synchronize()
            train_ms += ev[0].elapsed_time(ev[1])
            eval_ms += ev[1].elapsed_time(ev[2])
        torch.cuda.synchronize()
        eager_elapsed = time.perf_counter() - t1
        probe.active = False
        nce_probe.active = False
        edge_probe.active = False
        gsp_probe.active = False
        gemm_probe.active = False
    if graphed is None:
        elapsed, losses, accs = eager_elapsed, l2, a2
    K = MODEL["hidden"]
    roof = probe.summary(K)
    roofline = None
    if roof:
        gbs = roof["bytes"] / roof["avg_s"] / 1e9
        traffic, traffic_note = measured_traffic("spmm_traffic.json")
        sched = getattr(ops, "_SPMM_SCHEDULE", "classes")
        kern = {"blocks": "spmm_blk_kernel + spmm_combine_kernel (egnn_spmm_csr_blk_f32 + combine",
                "segments": "spmm_short_rows_kernel + spmm_combine_kernel (egnn_spmm_csr_seg_f32"}.get(
                    sched, "spmm_{short_rows,rows,long_rows}_kernel (egnn_spmm_csr_f32")
        roofline = dict(bound="hbm", kernel=f"{kern}, K={K}, reduce=sum)",
                        achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                        frac_of_measured_copy_peak=round(gbs / 6290.0, 4),
                        algorithmic_bytes_per_launch=roof["bytes"], avg_launch_us=round(roof["avg_s"] * 1e6, 2),
                        launches_timed=roof["launches"], traffic=traffic, traffic_source=traffic_note)
        try:   # the same call's gather stream alone (no reduction, no output): what the request path delivers on this graph
            import efficient_gnns_amd as _E
            gc = gather_ceiling(_E.gcn_norm(d.adj_t) if args.gnn == "gcn" else d.adj_t, K, device)   # the matrix the K-wide calls aggregate over
        except Exception as e:  # noqa: BLE001  (a diagnostic must not take the headline line down)
            gc = None
            roofline["gather_ceiling_error"] = f"{type(e).__name__}: {str(e)[:200]}"
        if gc:
            lines_gbs = gc["gathered_bytes"] / roof["avg_s"] / 1e9
            ceil_gbs = gc["gathered_bytes"] / gc["secs"] / 1e9
            roofline.update(gathered_line_bytes_per_launch=gc["gathered_bytes"], gather_GBs=round(lines_gbs, 1),
                            gather_ceiling_GBs=round(ceil_gbs, 1), gather_ceiling_us=round(gc["secs"] * 1e6, 2),
                            gather_ceiling_grid=dict(blocks_per_slice=gc["blocks_per_slice"], loads_in_flight=gc["loads_in_flight"]),
                            frac_of_gather_ceiling=round(lines_gbs / ceil_gbs, 4),
                            gather_ceiling_what="egnn_probe_gather_lines_f32 in this run: the call's nnz x K x 4 bytes of 128-byte lines, same "
                                                "slice <-> XCD binding, no reduction / values / output (HIP events, 10 launches, best of 8 launch shapes)")
    roofline_mfma = None
    nsum = nce_probe.summary()
    if nsum:
        tf = nsum["flops"] / nsum["secs"] / 1e12
        # the products run on the bf16 matrix pipe (csrc/gemm_split.h: three bf16 terms per fp32 operand, six exact partial
        # products per fp32 product), unless EGNN_GEMM_PIPE=f32 pins them to the f32-input MFMA
        split = not os.environ.get("EGNN_GEMM_PIPE", "").startswith("f")
        peak = 2500.0 / 6.0 if split else 157.3
        roofline_mfma = dict(bound="mfma", kernel="nce_fwd_kernel + nce_bwd_kernel x2 + split-K reduce (egnn_nce_fwd_f32, egnn_nce_bwd_f32): "
                                                  "the G-CRD loss, largest share of the step", achieved=round(tf, 1), peak=round(peak, 1),
                             unit="TFLOP/s (fp32 products)", frac=round(tf / peak, 4),
                             dtype="f32 operands and accumulators; products as 6 x v_mfma_f32_32x32x16_bf16 on a three-way bf16 split "
                                   "(peak = 2500 dense bf16 TFLOP/s / 6)" if split else "f32 (v_mfma_f32_32x32x2_f32)",
                             mfma_tflops_issued=round(tf * (6.0 if split else 1.0), 1),
                             flops_per_step=int(nsum["flops"] / max(1, n_probe)),
                             ms_per_step=round(1e3 * nsum["secs"] / max(1, n_probe), 3), calls_timed=nsum["calls"])
    roofline_gemm = None
    gmsum = gemm_probe.summary()
    if gmsum and gmsum["wide"][1] > 0:
        split = not os.environ.get("EGNN_GEMM_PIPE", "").startswith("f")
        peak = 2500.0 / 6.0 if split else 157.3
        fl, secs = gmsum["wide"]
        tf = fl / secs / 1e12
        per_shape = {k: dict(calls_per_step=round(c / max(1, n_probe), 2), us_per_call=round(1e6 * t / c, 1), tflops=round(f / t / 1e12, 1),
                             frac=round(f / t / 1e12 / peak, 3)) for k, (f, t, c) in sorted(gmsum["shapes"].items(), key=lambda kv: -kv[1][1])}
        roofline_gemm = dict(bound="mfma", kernel="every egnn_gemm_f32 / _ex / _add / _rows call of the step outside the G-CRD loss with all of "
                                                  "M, N, K > 64: layer transforms, input gradients, transposed weight gradients (split-K), projection "
                                                  "heads with fused row gathers (the teacher head's constant input cut once into planes: "
                                                  "egnn_gemm_{tn,rows}_planes_f32); pack / split-K-reduce launches included",
                             achieved=round(tf, 1), peak=round(peak, 1), unit="TFLOP/s (fp32 products)", frac=round(tf / peak, 4),
                             flops_per_step=int(fl / max(1, n_probe)), ms_per_step=round(1e3 * secs / max(1, n_probe), 3),
                             calls_per_step=round(gmsum["n_wide"] / max(1, n_probe), 2), by_shape=per_shape,
                             skinny_calls=dict(what="calls with a class-count-wide dimension (<= 64): HBM-bound kernels of gemm_skinny.hip, not in frac",
                                               calls_per_step=round(gmsum["n_skinny"] / max(1, n_probe), 2),
                                               ms_per_step=round(1e3 * gmsum["skinny"][1] / max(1, n_probe), 3)))
    if roofline is not None:
        # the two matrix-pipe fractions inside the object the driver's parser keeps; full objects: roofline_mfma / roofline_gemm
        roofline["mfma_frac"] = None if roofline_mfma is None else roofline_mfma["frac"]
        roofline["gemm_frac"] = None if roofline_gemm is None else roofline_gemm["frac"]
        roofline["mfma_gemm_frac_what"] = ("fractions of the 416.7 TFLOP/s six-product peak (2500 dense bf16 / 6); mfma_frac = the G-CRD entry points, "
                                           "gemm_frac = every other wide GEMM call of the step; HIP events around each call in the EAGER probe epochs "
                                           "that follow the timed region (stand-alone launches with event gaps between them, not kernels inside the "
                                           "replayed graph: rocprofv3 of the replayed epoch, profiles/r06_bench_kernel_stats.csv, is the in-epoch figure)")
    roofline_edges = None
    esum = edge_probe.summary()
    if esum and edge_index is not None:
        # SURVEY 8(d) row "LSP lpw fwd": compulsory bytes = E_tr (8 index bytes) + 4 N_tr (Ds + Dt) -- every train row of both feature
        # matrices once; the gathered figure E_tr * 4 * 2 * (Ds + Dt) is what the kernels actually request (mostly L2 / Infinity-Cache hits)
        E_tr, n_tr = int(edge_index.shape[1]), int(d.split_idx["train"].numel())
        Ds, Dt = MODEL["hidden"], int(d.teacher_out_feat.shape[1])
        alg = E_tr * 8 + 4 * n_tr * (Ds + Dt)
        gathered = E_tr * 4 * 2 * (Ds + Dt)
        per_step = esum["secs"] / esum["steps"]
        roofline_edges = dict(bound="hbm", kernel="edge_sim_kernel x 2 (student [N,256], teacher [N,750]) + lsp_loss_fwd_kernel "
                                                  "(egnn_edge_sim_f32, egnn_lsp_loss_fwd_f32): the LSP forward of one step",
                              achieved=round(alg / per_step / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg / per_step / 1e9 / HBM_PEAK_GBS, 4),
                              algorithmic_bytes_per_step=alg, us_per_step=round(per_step * 1e6, 1), steps_timed=esum["steps"],
                              gathered_bytes_per_step=gathered, gather_GBs=round(gathered / per_step / 1e9, 1),
                              note="E_tr = %d train-subgraph edges, N_tr = %d; rows are gathered 7.5 times each on average, so the request "
                                   "stream (gather_GBs) is what the memory system serves, from L2 / Infinity Cache for the most part" % (E_tr, n_tr),
                              traffic=None)
    roofline_gsp = None
    gsum = gsp_probe.summary()
    if gsum:
        tf = gsum["flops"] / gsum["secs"] / 1e12
        roofline_gsp = dict(bound="mfma", kernel="gsp_fwd_kernel (egnn_gsp_fwd_f32): student + teacher Gram tiles and the squared difference of "
                                                 "the similarity matrices, the GSP forward of one step", achieved=round(tf, 1), peak=157.3,
                            unit="TFLOP/s (fp32-input MFMA)", frac=round(tf / 157.3, 4), flops_per_call=gsum["flops"] / gsum["calls"],
                            us_per_call=round(gsum["secs"] / gsum["calls"] * 1e6, 1), calls_timed=gsum["calls"],
                            shape=dict(S=gsum["S"], P_student=gsum["Ps"], P_teacher=gsum["Pt"]),
                            note="the backward (two GEMMs on the stored weight matrices) runs on the split-bf16 pipeline through egnn_gemm_f32; the "
                                 "forward stays on the f32-input MFMA: with 8 k-steps per Gram product and two [S,S] weight matrices to write the kernel is "
                                 "epilogue-bound -- on the split pipeline it took 203 us against 190 us (round 6, same box; profiles/r06_gsp_split_null.txt)")
    roofline_local = None
    if not args.no_local_roofline:
        try:
            roofline_local = local_graph_roofline(args, device, ops)
        except Exception as e:  # noqa: BLE001  (a secondary object must not take the headline line down)
            roofline_local = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
    parity = None if args.no_parity else parity_check(args, data, d, device, hp, PM)
    # the other untimed GPU leg: the reference's own loop through dropin/ (own model, own optimizer; restores every hook it installs)
    ref_loop = reference_loop(args, d, device, hp, epochs=args.reference_epochs) if args.reference_epochs > 0 else None
    cpu = cpu_baseline(args, data, hp) if args.cpu_epochs > 0 else None

    out = dict(
        metric="training epochs/sec, ogbn-arxiv 3-layer GCN student + G-CRD, 1/2/4/8 MI355X",
        value=round(args.steps / elapsed, 3), unit="epochs/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
        ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True, scaling="strong", vs_baseline=None,
        dtype="f32", data="synthetic",
        arithmetic=("fp32 tensors, fp32 accumulation everywhere; GEMM / G-CRD products formed on the bf16 matrix pipe as the six exact "
                    "partial products of a three-way bf16 split of each fp32 operand (error vs float64 not above the f32-input MFMA's: "
                    "tests/test_gpu_parity.py::test_split_pipeline_error_is_not_above_the_f32_mfma_pipeline; DESIGN.md 3.2); "
                    "EGNN_GEMM_PIPE=f32 pins them to v_mfma_f32_32x32x2_f32")
                   if not os.environ.get("EGNN_GEMM_PIPE", "").startswith("f") else "fp32 tensors and accumulation; products on the f32-input MFMA",
        config=dict(workload=f"ogbn-arxiv-shaped synthetic graph (N={d.num_nodes}, nnz_sym={d.adj_t.nnz()}), "
                             f"3-layer {args.gnn.upper()}-256 student + {args.training}"
                             f"{' (G-CRD)' if args.training == 'nce' else ''} loss (max_samples={hp['max_samples']}, "
                             f"proj_dim={hp['proj_dim']}, nce_T={hp['nce_T']}, beta={hp['beta']}"
                             f"{', kernel=' + hp['kernel'] if args.training in ('gpw', 'lpw') else ''}), full-graph train step + eval per epoch",
                    gemm_backend=ops.gemm_backend(), partitioning="single GPU",
                    spmm_schedule=getattr(ops, "_SPMM_SCHEDULE", "classes"),
                    gcn_operand_order="aggregate on the narrower side of W (layer 1: (A x) W; same product as A (x W))",
                    adam="fused",
                    memoise_first_layer_aggregation=os.environ.get("EGNN_GCN_MEMOISE_AX", "0") == "1",
                    cache_constant_row_gathers=os.environ.get("EGNN_CACHE_CONST_ROWS", "0") == "1"),
        roofline=roofline, roofline_local=roofline_local, roofline_mfma=roofline_mfma, roofline_gemm=roofline_gemm, roofline_edges=roofline_edges, roofline_gsp=roofline_gsp, cpu_baseline=cpu, parity=parity,
        launch=graph_note,
        eager=dict(epochs_per_s=round(n_probe / eager_elapsed, 3), epochs=n_probe,
                   note="eager launches with per-kernel event brackets (where the roofline objects are measured)"),
        phases_ms=dict(train_step=round(train_ms / max(1, n_probe), 3), eval=round(eval_ms / max(1, n_probe), 3)),
        last_losses=[round(float(v), 5) for v in losses], last_accs=[round(float(a), 4) for a in accs],
        reference_loop=ref_loop,
        timing=timing,
        repeat_blocks_ms_per_step=[round(v, 3) for v in block_ms],   # further blocks of `steps` replays after the contract's block
        repeat_blocks_median_ms_per_step=(round(float(np.median(block_ms)), 3) if block_ms else None),
    )
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(out["value"] / cpu["value"], 1)
    if ref_loop and "epochs_per_s" in ref_loop:
        ref_loop["fraction_of_package_loop"] = round(ref_loop["epochs_per_s"] / out["value"], 4)
    print(json.dumps(out), flush=True)
    if parity is not None and not parity["ok"]:
        raise SystemExit("bench.py: full-size parity against the CPU oracle FAILED (see the 'parity' object of the JSON line)")


if __name__ == "__main__":
    main()
