"""world_size-2 gloo (CPU) coverage of the node-range sharded path (efficient-gnns_amd/dist.py).

The HIP kernels need a GPU, so for this host-logic test the product's ``ops`` entry points are monkeypatched
with oracle-backed CPU functions (tests may use the oracle as a stand-in; the product package never does).
What is verified is the distributed logic itself: partition plan, halo all_to_all with autograd, SyncBN,
the row-block G-CRD with its collectives, loss scaling and the flat gradient all-reduce -- against the
single-process oracle on the same graph, weights and NumPy sample.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _patch_ops_with_oracle():
    import efficient_gnns_amd.ops as ops
    from efficient_gnns_amd import _lib
    # the ONE seam (round 6): the package has a single code path -- every operator goes through `ops.*`, `_lib.require_gpu` refuses CPU
    # tensors; this test replaces both from the outside (no switch inside the product)
    _lib.require_gpu = lambda *tensors: None
    import oracle.sparse as OS
    import torch.nn.functional as F
    ops.bn_shape_ok = lambda x: False          # SyncBatchNorm1d then takes its torch-operator form (the one for widths the kernels refuse)
    ops.colsum = lambda g: g.sum(0)
    ops.add_bias = lambda x, bias: x if bias is None else x + bias
    ops.linear_add = lambda x, w, b, addend: F.linear(x, w, b) + addend
    ops.linear_rows = lambda x, idx, w, b=None, const_input=False: F.linear(x[idx], w, b)

    def split_accuracy(logits, y, split_idx, counts=False, out=None):
        y_pred, yv = logits.argmax(dim=-1), y.view(-1)
        keys = ("train", "valid", "test")
        hits = torch.stack([(yv[split_idx[k]] == y_pred[split_idx[k]]).sum() for k in keys]).double()
        sizes = torch.tensor([float(split_idx[k].numel()) for k in keys], dtype=torch.float64)
        res = torch.cat([hits, sizes]) if counts else hits / sizes
        return res if out is None else out.copy_(res)
    ops.split_accuracy = split_accuracy

    def to_oracle(adj):
        rowptr, col, val = adj.csr()
        return OS.SparseTensor(rowptr=rowptr, col=col, value=val, sparse_sizes=adj.sparse_sizes())

    ops.spmm = lambda adj, x, reduce="sum", bias=None, addend=None, **_: (OS.matmul(to_oracle(adj), x, reduce) + (0 if bias is None else bias)
                                                                            + (0 if addend is None else addend))
    ops.take_rows = lambda x, idx: x[idx]
    ops.matmul = lambda x, w, bias=None: x @ w if bias is None else x @ w + bias
    ops.linear = lambda x, w, b=None: F.linear(x, w, b)
    ops.cross_entropy = lambda logits, labels, rows=None: F.cross_entropy(logits if rows is None else logits[rows], labels if rows is None else labels[rows])

    def ce_and_kd(logits, labels, teacher, T, rows=None):
        if rows is not None:
            logits, labels, teacher = logits[rows], labels[rows], teacher[rows]
        return (F.cross_entropy(logits, labels),
                F.kl_div(F.log_softmax(logits / T, dim=1), F.softmax(teacher / T, dim=1), log_target=False))
    ops.ce_and_kd = ce_and_kd
    ops.gather_normalize = lambda x, idx=None, eps=1e-12: F.normalize(x if idx is None else x[idx], p=2, dim=-1)

    def nce_block_fwd(fhat, t_all, off, tau, inv_count):
        z = fhat.detach() @ t_all.detach().t() / tau
        lse = torch.logsumexp(z, dim=1)
        diag = z[torch.arange(z.shape[0]), torch.arange(z.shape[0]) + off]
        return z, lse, ((lse - diag).sum() * inv_count).reshape(1)

    def nce_block_bwd(fhat, t_all, off, scale, Z, lse, g, tau=None, unit_rows=True):
        p = torch.exp(Z - lse[:, None])
        p[torch.arange(Z.shape[0]), torch.arange(Z.shape[0]) + off] -= 1.0
        return scale * g * (p @ t_all), scale * g * (p.t() @ fhat)
    ops.nce_block_fwd, ops.nce_block_bwd = nce_block_fwd, nce_block_bwd

    import efficient_gnns_amd.ops_edge as ops_edge
    import efficient_gnns_amd.ops_pairwise as ops_pairwise
    import oracle.criterion as OC

    def lsp_loss(feat, teacher_feat, edge_index, kernel, criterion="kld"):
        src, dst = edge_index
        p_s = OC._segment_softmax(OC._edge_sim(feat, src, dst, kernel), dst)
        p_t = OC._segment_softmax(OC._edge_sim(teacher_feat, src, dst, kernel), dst)
        return F.kl_div(torch.log(p_s), p_t, log_target=False)
    ops_edge.lsp_loss = lsp_loss
    ops_pairwise.gsp_loss = lambda fs, ts, idx, kernel: F.mse_loss(OC._pairwise(fs, kernel), OC._pairwise(ts, kernel))


def _make_data(seed=3, train_ids_below=None):
    import types
    import efficient_gnns_amd.data as D
    import oracle.sparse as OS
    d = D.arxiv_like(scale=0.004, seed=seed)  # ~680 nodes
    if train_ids_below is not None:   # id-ordered split: every train node in the low id range (the last shards own none)
        tr = d.split_idx["train"]
        d.split_idx["train"] = tr[tr < train_ids_below].clone()
    rowptr, col, _ = d.adj_t.csr()
    oadj = OS.SparseTensor(rowptr=rowptr, col=col, sparse_sizes=d.adj_t.sparse_sizes())
    g = OS.gcn_norm_sparse(oadj)
    import efficient_gnns_amd as E
    d.gcn_struct = E.SparseTensor(rowptr=g.csr()[0], col=g.csr()[1], value=g.csr()[2], sparse_sizes=g.sparse_sizes())
    d.oracle_adj = oadj
    return d


HP = dict(alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=96, kernel="rbf")


def _reference_run(gnn, mode, steps=3, hp=None):
    HP = dict(globals()["HP"], **(hp or {}))
    HP.pop("static_sigmas", None)
    HP.pop("host_staged", None)
    HP.pop("agg_mode", None)
    import oracle.models as OM
    d = _make_data(train_ids_below=HP.pop("train_ids_below", None))
    torch.manual_seed(0)
    np.random.seed(0)
    model = (OM.GCN if gnn == "gcn" else OM.SAGE)(d.num_features, 32, d.num_classes, 3, 0.0)
    sp = tp = None
    groups = [{"params": model.parameters(), "lr": 0.01}]
    if mode in ("nce", "gpw"):
        sp, tp = OM.make_projection(32, 16), OM.make_projection(750, 16)
        groups += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
    opt = torch.optim.Adam(groups)
    edge_index = None
    if mode == "lpw":   # gnn.py:246-250: the train-node subgraph, relabelled to train positions
        import oracle.utils as OU
        rowptr, col, _ = d.adj_t.csr()
        row = torch.repeat_interleave(torch.arange(d.num_nodes), rowptr[1:] - rowptr[:-1])
        edge_index = OU.subgraph(d.split_idx["train"], torch.stack([row, col]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
    logits, accs = OM.evaluate(model, d.x, d.oracle_adj, d.y, d.split_idx)   # eval at the initial state
    losses = [OM.train_step(model, d.x, d.oracle_adj, d.y, d.split_idx["train"], opt, mode, HP, d.teacher_out_feat,
                            d.teacher_logits, sp, tp, edge_index) for _ in range(steps)]
    return losses, logits, accs


def _free_port() -> int:
    """A port the kernel just handed out (bind to 0): no collisions between the tests of one session."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _quiet_exit():
    """End a rank: all ranks first meet at a barrier (none tears its connections down while a peer still talks), then the
    process leaves WITHOUT running the interpreter / C++ static destructors: gloo's background threads otherwise race the
    teardown now and then ('terminate called without an active exception', exit code -6, after every result was
    delivered).  The SimpleQueue payload is written synchronously, nothing is left to flush."""
    try:
        dist.barrier()
        dist.destroy_process_group()
    finally:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def _worker(rank, world, port, gnn, mode, q, hp=None):
    HP = dict(globals()["HP"], **(hp or {}))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _patch_ops_with_oracle()
        import efficient_gnns_amd.dist as DD
        import efficient_gnns_amd.models as PM
        if HP.pop("host_staged", False):   # hostcomm's staging code on host tensors (its device <-> host copies are no-ops here)
            from efficient_gnns_amd import hostcomm
            hostcomm._STAGE_HOST_TENSORS = True
            hostcomm.install()
        d = _make_data(train_ids_below=HP.pop("train_ids_below", None))
        # "halo" (what these tests pin unless they say otherwise): the referenced rows travel; "sliced": every kernel-aligned aggregation
        # re-shards the feature columns instead ("auto" -- the product default -- already picks sliced at 4 ranks on this 677-node graph)
        DD._AGG_MODE = HP.pop("agg_mode", "halo")
        prob = DD.ShardedProblem(d, world, rank, "cpu", None, need_gcn=True)
        if "static_sigmas" in HP:    # the sampled criteria in draw-independent shapes (what ShardedGraphedEpoch captures on > 1 rank)
            prob.static_sample = DD.StaticSample(prob, HP["max_samples"], sigmas=HP.pop("static_sigmas"))
        torch.manual_seed(0)
        np.random.seed(0)
        model = (PM.GCN if gnn == "gcn" else PM.SAGE)(d.num_features, 32, d.num_classes, 3, 0.0)
        sp = tp = None
        groups = [{"params": model.parameters(), "lr": 0.01}]
        if mode in ("nce", "gpw"):
            sp, tp = PM.make_projection(32, 16), PM.make_projection(750, 16)
            groups += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
        DD.swap_batchnorm(model)
        if sp is not None:
            DD.swap_batchnorm(sp)
            DD.swap_batchnorm(tp)
        opt = torch.optim.Adam(groups)
        out, accs = DD.sharded_evaluate(model, prob)
        losses = [DD.sharded_train_step(model, prob, opt, mode, HP, sp, tp) for _ in range(3)]
        with DD.CommTrace() as trace:          # one more step with the collectives recorded: which exchange forms ran
            DD.sharded_train_step(model, prob, opt, mode, HP, sp, tp)
        gathered, comm = [None] * world, [None] * world
        dist.all_gather_object(gathered, out.numpy())
        dist.all_gather_object(comm, (trace.summary(), trace.records))
        if rank == 0:
            q.put((losses, np.concatenate(gathered, 0), accs, prob.adj.plan.n_halo,
                   dict(per_rank=[c[0] for c in comm], consistent=DD.consistent_collectives([c[1] for c in comm]))))
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    _quiet_exit()


@pytest.mark.parametrize("gnn,mode,world,max_samples", [
    ("gcn", "kd", 2, 96), ("gcn", "nce", 2, 96), ("sage", "nce", 2, 96), ("sage", "supervised", 2, 96),
    ("gcn", "nce", 3, 96),     # 3 ranks: uneven node ranges and sample counts
    ("gcn", "nce", 4, 5),      # 5 samples over 4 ranks: some ranks own no sampled row (empty row block, collectives still run)
    # every train node in the first 300 ids: the last rank(s) own NO train row -- their loss terms must stay attached to
    # the graph so that all ranks run the same backward collectives (no 'does not require grad', no hang)
    ("gcn", "kd", 3, -300), ("gcn", "nce", 3, -300), ("sage", "supervised", 2, -300),
    # GSP (all-pairs loss on the gathered sample, evaluated on every rank) and LSP (train-subgraph edges, own halo plan)
    ("gcn", "gpw", 2, 96), ("gcn", "gpw", 3, 40), ("sage", "lpw", 2, 96), ("gcn", "lpw", 3, 96), ("gcn", "lpw", 3, -300)])
def test_sharded_training_matches_single_process_oracle(gnn, mode, world, max_samples):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    hp = dict(max_samples=max_samples)
    if max_samples < 0:
        hp = dict(max_samples=64, train_ids_below=-max_samples)
    if mode in ("gpw", "lpw"):   # the weights of record (scripts/run_gcn.sh: beta = 100): the auxiliary gradient dominates the step
        hp.update(kernel="cosine", beta=100.0)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gnn, mode, q, hp)) for r in range(world)]
    for p in procs:
        p.start()
    losses, logits, accs, n_halo, comm = q.get()  # read before join: the payload is larger than the pipe buffer
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert comm["consistent"] is None, comm["consistent"]
    assert all("sliced_exchanges" not in c for c in comm["per_rank"])      # (these cases pin the halo form)

    ref_losses, ref_logits, ref_accs = _reference_run(gnn, mode, hp=hp)
    np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=2e-4, atol=1e-6)
    # eval is compared at the initial state: after Adam steps the pre-BatchNorm biases are rounding-noise driven
    # (tests/golden/make_golden.py) and eval-mode logits stop being reproducible across implementations
    np.testing.assert_allclose(logits, ref_logits.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(accs, ref_accs, atol=1e-9)
    assert n_halo > 0


@pytest.mark.parametrize("gnn,mode,world", [("gcn", "nce", 2), ("gcn", "kd", 4), ("sage", "nce", 4), ("sage", "lpw", 2), ("gcn", "nce", 3)])
def test_column_sliced_aggregation_matches_single_process_oracle(gnn, mode, world):
    """the sharded step with the aggregation in COLUMN-SLICED form (dist._SlicedAggregate: all_to_all of the feature
    columns to [N, K / world] slices, aggregation of all rows on the all-gathered full adjacency, all_to_all back -- bytes independent of
    the halo) against the single-process oracle: same losses over three steps, same initial eval.  World 3: the 32-wide layers are not
    divisible into kernel-aligned slices (32 % 12), every aggregation stays in halo form -- the mode is a per-call decision."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    hp = dict(max_samples=96, agg_mode="sliced")
    if mode == "lpw":
        hp.update(kernel="cosine", beta=100.0)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gnn, mode, q, hp)) for r in range(world)]
    for p in procs:
        p.start()
    losses, logits, accs, n_halo, comm = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert comm["consistent"] is None, comm["consistent"]
    if world == 3:
        assert all("sliced_exchanges" not in c for c in comm["per_rank"])
    else:
        # GCN: the 32-wide hidden aggregation forward + backward (the static input and the class-wide output layer stay in halo form);
        # every sliced aggregation is two exchanges; all ranks put the same number of bytes on the wire (equal node ranges or not: the
        # formula is N K 4 (G - 1) / G^2 per direction up to the last range's remainder)
        assert all(c["sliced_exchanges"] >= 4 and c["sliced_exchanges"] % 2 == 0 for c in comm["per_rank"]), comm["per_rank"]
        assert all(c["sliced_all_to_all_bytes_sent"] > 0 for c in comm["per_rank"])
    ref_losses, ref_logits, ref_accs = _reference_run(gnn, mode, hp=hp)
    np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(logits, ref_logits.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(accs, ref_accs, atol=1e-9)


def test_shard_plan_integer_logic():
    import efficient_gnns_amd.dist as DD
    d = _make_data(seed=7)
    rowptr, col, _ = d.adj_t.csr()
    n, world = d.num_nodes, 3
    plans = [DD.ShardPlan.from_global(rowptr, col, None, n, world, r) for r in range(world)]
    dense = torch.zeros(n, n)
    dense[d.adj_t.storage.row(), col] = 1
    for r, p in enumerate(plans):
        # local rows reproduce the global rows after un-mapping the extended column ids
        ext_to_global = torch.cat([torch.arange(p.lo, p.hi), p.halo_ids])
        rows = torch.repeat_interleave(torch.arange(p.n_local), p.rowptr_local[1:] - p.rowptr_local[:-1])
        loc = torch.zeros(p.n_local, n)
        loc[rows, ext_to_global[p.col_ext]] = 1
        assert torch.equal(loc, dense[p.lo:p.hi])
        assert sum(p.recv_counts) == p.n_halo and p.recv_counts[r] == 0
        # what r receives from q is exactly what q believes it must send to r, in the same (ascending) order
        off = 0
        for q_, cnt in enumerate(p.recv_counts):
            ids = p.halo_ids[off:off + cnt]
            off += cnt
            pq = plans[q_]
            so = sum(pq.send_counts[:r])
            assert torch.equal(pq.send_idx[so:so + pq.send_counts[r]] + pq.lo, ids)


def _trace_worker(rank, world, port, gnn, mode, q, hp=None):
    HP = dict(globals()["HP"], **(hp or {}))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _patch_ops_with_oracle()
        import efficient_gnns_amd.dist as DD
        import efficient_gnns_amd.models as PM
        d = _make_data(train_ids_below=HP.pop("train_ids_below", None))
        prob = DD.ShardedProblem(d, world, rank, "cpu", None, need_gcn=True)
        if "static_sigmas" in HP:
            prob.static_sample = DD.StaticSample(prob, HP["max_samples"], sigmas=HP.pop("static_sigmas"))
        torch.manual_seed(0)
        np.random.seed(0)
        model = (PM.GCN if gnn == "gcn" else PM.SAGE)(d.num_features, 32, d.num_classes, 3, 0.5)
        sp = tp = None
        groups = [{"params": model.parameters(), "lr": 0.01}]
        if mode in ("nce", "gpw"):
            sp, tp = DD.swap_batchnorm(PM.make_projection(32, 16)), DD.swap_batchnorm(PM.make_projection(750, 16))
            groups += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
        DD.swap_batchnorm(model)
        opt = torch.optim.Adam(groups)
        DD.sharded_train_step(model, prob, opt, mode, HP, sp, tp)          # warm-up: one-off exchanges (static halo, teacher rows)
        steps = []
        for _ in range(3):
            with DD.CommTrace() as tr:
                DD.sharded_train_step(model, prob, opt, mode, HP, sp, tp)
                DD.sharded_evaluate(model, prob)
            steps.append(tr.records)
        everyone = [None] * world
        dist.all_gather_object(everyone, steps)
        if rank == 0:
            q.put(everyone)
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    _quiet_exit()


@pytest.mark.parametrize("gnn,mode,world,max_samples", [("gcn", "nce", 2, 96), ("gcn", "nce", 3, 5), ("gcn", "kd", 3, -300), ("sage", "lpw", 3, -300),
                                                        ("gcn", "gpw", 2, 40)])
def test_every_rank_issues_the_same_collective_sequence(gnn, mode, world, max_samples):
    """The sharded step is a fixed collective PROGRAM: every rank issues the same operations in the same order with matching
    sizes, step after step -- also a rank that owns no train row or no sampled row (its loss terms are attached zeros so that
    its backward still runs every halo exchange and SyncBN reduction).  Any mismatch here is a hang over RCCL."""
    import efficient_gnns_amd.dist as DD
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    hp = dict(max_samples=max_samples)
    if max_samples < 0:
        hp = dict(max_samples=64, train_ids_below=-max_samples)
    if mode in ("gpw", "lpw"):
        hp.update(kernel="cosine", beta=100.0)
    port = _free_port()
    procs = [ctx.Process(target=_trace_worker, args=(r, world, port, gnn, mode, q, hp)) for r in range(world)]
    for p in procs:
        p.start()
    everyone = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert len(everyone) == world
    for step in range(3):
        per_rank = [everyone[r][step] for r in range(world)]
        assert len(per_rank[0]) > 10, "the trace is empty: the step issued no collectives?"
        bad = DD.consistent_collectives(per_rank)
        assert bad is None, f"step {step}: {bad}"
    # the program does not change from step to step (same operations and payload sizes; the per-peer split of the sampled rows may)
    for r in range(world):
        sig = [[(o[0], o[3]) if o[0] != "all_to_all_single" else (o[0], o[3]) for o in everyone[r][st]] for st in range(3)]
        assert [s_[0] for s_ in sig[0]] == [s_[0] for s_ in sig[1]] == [s_[0] for s_ in sig[2]], f"rank {r}: the operation sequence changes between steps"
    # and the checker itself catches a mismatch
    broken = [list(everyone[r][0]) for r in range(world)]
    broken[-1] = broken[-1][:-1]
    assert DD.consistent_collectives(broken) is not None
    a2a = next(i for i, o in enumerate(everyone[0][0]) if o[0] == "all_to_all_single" and o[4] is not None)
    broken = [list(everyone[r][0]) for r in range(world)]
    o = broken[0][a2a]
    broken[0][a2a] = (o[0], o[1], o[2], o[3], tuple(v + 1 for v in o[4]), o[5])
    assert DD.consistent_collectives(broken) is not None


def test_sharded_adj_from_a_prebuilt_plan_without_a_process_group():
    """``ShardPlan.from_global`` + ``ShardedAdj(_plan=...)`` (single-process tools): the raw adjacency is built without any
    collective; the normalised one needs the peers' degrees, so asking for it without a process group is a clear error."""
    import efficient_gnns_amd.dist as DD
    d = _make_data(seed=7)
    rowptr, col, _ = d.adj_t.csr()
    n, world, rank = d.num_nodes, 2, 1
    plan = DD.ShardPlan.from_global(rowptr, col, None, n, world, rank)
    lo, hi, _ = DD.node_range(n, world, rank)
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    sadj = DD.ShardedAdj(rowptr[lo:hi + 1] - e0, col[e0:e1], n, world, rank, "cpu", with_gcn=False, _plan=plan)
    assert sadj.plan is plan and sadj.raw.sparse_sizes() == (plan.n_local, plan.n_local + plan.n_halo)
    with pytest.raises(RuntimeError, match="with_gcn=False"):
        sadj.gcn_normalized()
    with pytest.raises(RuntimeError, match="process group"):
        DD.ShardedAdj(rowptr[lo:hi + 1] - e0, col[e0:e1], n, world, rank, "cpu", with_gcn=True, _plan=plan)


def _bench_worker(rank, world, port, path, workload="arxiv", overlap="1"):
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), EGNN_DIST_OVERLAP=overlap)
    _patch_ops_with_oracle()
    import efficient_gnns_amd.dist as DD
    workload, _, training = workload.partition("-")          # "arxiv-gpw": the arxiv workload with another loss
    graph_kind = "chunglu"
    if training == "local":                                   # "arxiv-local": the community graph (ids shuffled): ranges cut from the community order
        training, graph_kind = "", "local"
    args = types.SimpleNamespace(seed=0, scale=(0.02 if graph_kind == "local" else 0.004) if workload == "arxiv" else 0.0006, gnn="gcn",
                                 training=training or "nce", warmup=1, steps=2, workload=workload, graph_kind=graph_kind)
    hp = dict(alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=128, proj_dim=16, kernel="rbf")
    cfg = dict(hidden=32, layers=3 if workload == "arxiv" else 2, dropout=0.5, lr=0.01)
    lines = []
    try:
        DD.bench_main(args, hp, cfg, rank, world, "cpu", backend="gloo", emit=lines.append)   # ends with barrier + destroy_process_group
        if rank == 0:
            with open(path, "w") as f:
                f.write(lines[0])
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)   # see _quiet_exit


@pytest.mark.parametrize("workload,world,overlap", [("arxiv", 2, "1"), ("arxiv", 2, "0"), ("mag", 2, "1"), ("mag", 4, "1"),
                                                    ("arxiv-gpw", 2, "1"), ("arxiv-lpw", 2, "1"), ("arxiv-local", 4, "1")])
def test_bench_entry_point_runs_sharded_and_prints_contract_json(tmp_path, workload, world, overlap):
    """bench.py's multi-rank entry (dist.bench_main) over gloo: the headline workload with and without the halo / compute
    overlap, and BASELINE.json configs[4] (MAG-shaped graph, SAGE-mean + logit KD) on 2 and 4 node-range shards."""
    import json
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "bench.json")
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, path, workload, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    out = json.loads(open(path).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out
    assert out["n_gpus"] == world and out["scaling"] == "strong" and out["value"] > 0 and "workload" in out["config"]
    assert ("mag" in out["config"]["workload"]) == (workload == "mag")   # ("arxiv-gpw" etc. are arxiv workloads)
    assert all(np.isfinite(out["last_losses"]))
    order = out["config"]["node_order"]
    assert out["comm_per_epoch"]["halo_rows"] == order
    assert "halo_rows_per_rank_as_given" in order and len(order["halo_rows_per_rank_as_given"]) == world
    if workload == "arxiv-local":    # locality found: the ranges were cut from the community order and the halo shrank
        assert order["order"].startswith("community order") and order["halo_rows_change"] < -0.1, order    # (-19 % on this 3.4 k-node graph)


def _plan_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import efficient_gnns_amd.dist as DD
        d = _make_data(seed=7)
        rowptr, col, _ = d.adj_t.csr()
        n = d.num_nodes
        lo, hi, _ = DD.node_range(n, world, rank)
        e0, e1 = int(rowptr[lo]), int(rowptr[hi])
        loc = DD.ShardPlan.from_local(rowptr[lo:hi + 1] - e0, col[e0:e1], None, n, world, rank)     # own rows only + index-list exchange
        glo = DD.ShardPlan.from_global(rowptr, col, None, n, world, rank)                           # read off the global structure
        same = (torch.equal(loc.send_idx, glo.send_idx) and loc.send_counts == glo.send_counts and loc.recv_counts == glo.recv_counts
                and torch.equal(loc.halo_ids, glo.halo_ids) and torch.equal(loc.col_ext, glo.col_ext))
        # local GCN normalisation of the shard == the rows of the globally normalised matrix (values bit-for-bit: same formula)
        sadj = DD.ShardedAdj(rowptr[lo:hi + 1] - e0, col[e0:e1], n, world, rank, "cpu", None, with_gcn=True)
        g = d.gcn_struct
        grp, gcol, gval = g.csr()
        g0, g1 = int(grp[lo]), int(grp[hi])
        pl = sadj.gcn_normalized().plan
        ext_to_global = torch.cat([torch.arange(lo, hi), pl.halo_ids])
        same_gcn = (torch.equal(pl.rowptr_local, grp[lo:hi + 1] - g0) and torch.equal(ext_to_global[pl.col_ext], gcol[g0:g1])
                    and torch.allclose(pl.value_local, gval[g0:g1], rtol=1e-6, atol=0))
        # the scatter matrix adds every received row into its owner row exactly once
        sc = sadj.scatter
        ok_scatter = sc.nnz() == loc.send_idx.numel() and torch.equal(torch.sort(sc.csr()[1]).values, torch.arange(sc.nnz()))
        # round 6: the FULL matrix the column-sliced form aggregates on, all-gathered from the shards' pieces (uneven ranges at 3 ranks):
        # A^ entry for entry, and the mean form of the raw adjacency (1 / rowcount per entry), both equal to the global structure
        full = sadj.gcn_normalized().full_adj(False, False)
        frp, fcol, fval = full.csr()
        same_full = (torch.equal(frp, grp) and torch.equal(fcol, gcol) and torch.allclose(fval, gval, rtol=1e-6, atol=0))
        mean_full = sadj.full_adj(True, True)
        mrp, mcol, mval = mean_full.csr()
        cnt = (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(torch.float32)
        same_mean = (torch.equal(mrp, rowptr) and torch.equal(mcol, col)
                     and torch.allclose(mval, torch.repeat_interleave(1.0 / cnt, rowptr[1:] - rowptr[:-1]), rtol=1e-6, atol=0))
        got = [None] * world
        dist.all_gather_object(got, (same, same_gcn, ok_scatter, same_full, same_mean))
        if rank == 0:
            q.put(got)
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    _quiet_exit()


@pytest.mark.parametrize("world", [2, 3])
def test_collective_plan_from_local_rows_equals_the_global_plan(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(all(t) for t in got), got


def _spawn(target, world, gnn, mode, hp):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, gnn, mode, q, hp)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return out


@pytest.mark.parametrize("gnn,mode,world,max_samples,sigmas", [
    ("gcn", "nce", 2, 96, 6.0), ("gcn", "nce", 3, 96, 6.0), ("sage", "nce", 2, 96, 6.0), ("gcn", "gpw", 2, 96, 6.0), ("gcn", "gpw", 3, 40, 6.0),
    ("gcn", "nce", 4, 5, 6.0),        # 5 samples over 4 ranks: row blocks of 2, 2, 1, 0 rows; ranks that own no sampled row
    ("gcn", "nce", 3, -300, 6.0),     # the last rank owns no train row: an empty block in the all-gather, still every collective
    ("gcn", "nce", 3, 96, -50.0),     # capacity 1: EVERY draw overflows -> the dynamic-shape fallback with the same draw
    ("gcn", "gpw", 2, 96, -50.0)])
def test_static_shape_sampled_criteria_match_the_oracle(gnn, mode, world, max_samples, sigmas):
    """G-CRD / GSP on shards in the draw-independent layout (dist.StaticSample: fixed-capacity row blocks + per-step permutation,
    G-CRD as even row blocks of the gathered S x S problem) reproduce the single-process oracle like the dynamic layout does --
    same np.random draws, 3 optimisation steps.  With a negative sigma the capacity is 1 row and every step takes the overflow
    fallback."""
    hp = dict(max_samples=max_samples, static_sigmas=sigmas)
    if max_samples < 0:
        hp.update(max_samples=64, train_ids_below=-max_samples)
    if mode == "gpw":
        hp.update(kernel="cosine", beta=100.0)
    losses, logits, accs, n_halo, _comm = _spawn(_worker, world, gnn, mode, hp)
    ref_losses, ref_logits, ref_accs = _reference_run(gnn, mode, hp=hp)
    np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=2e-4, atol=1e-6)


def _hostcomm_shapes_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efficient_gnns_amd import hostcomm
        hostcomm._STAGE_HOST_TENSORS = True
        hostcomm.install()
        # [world, C] output for a [C] input (ops._sync_stats): fine over RCCL, gloo itself insists on the flat concatenation
        out = torch.empty(world, 5)
        dist.all_gather_into_tensor(out, torch.arange(5.) + 10 * rank)
        ok = all(torch.equal(out[r], torch.arange(5.) + 10 * r) for r in range(world))
        o2 = torch.empty(world * 3, 4)
        dist.all_gather_into_tensor(o2, torch.full((3, 4), float(rank)))
        ok = ok and all(bool(o2[3 * r:3 * r + 3].eq(r).all()) for r in range(world))
        full = torch.arange(world * 12.).view(world * 3, 4)
        rs = torch.empty(3, 4)
        dist.reduce_scatter_tensor(rs, full.clone())                  # (gloo has none: all_reduce + this rank's block)
        ok = ok and torch.equal(rs, world * full[rank * 3:(rank + 1) * 3])
        send = torch.arange(6.).view(3, 2) + 100 * rank                  # rows 0..r0 to rank 0, the rest to rank 1 (world 2)
        counts_out = counts_in = [1, 2] if rank == 0 else [2, 1]      # rank 0 sends 1 + 2 rows and receives 1 + 2; rank 1: 2 + 1
        recv = torch.empty(sum(counts_in), 2)
        work = dist.all_to_all_single(recv, send, counts_in, counts_out, async_op=True)
        work.wait()
        want = torch.cat([send[:1], send[:2] + 100]) if rank == 0 else torch.cat([send[1:] - 100, send[2:]])
        ok = ok and torch.equal(recv, want)
        red = torch.ones(3) * (rank + 1)
        dist.all_reduce(red)
        ok = ok and bool(red.eq(sum(range(1, world + 1))).all()) and hostcomm.stats()["calls"] == 5
        got = [None] * world
        dist.all_gather_object(got, bool(ok))
        if rank == 0:
            q.put(got)
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    _quiet_exit()


def test_host_staged_collectives_shapes_and_semantics():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_hostcomm_shapes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got == [True, True]


@pytest.mark.parametrize("gnn,mode,world", [("gcn", "nce", 3), ("sage", "lpw", 2)])
def test_host_staged_collectives_carry_the_same_program(gnn, mode, world):
    """efficient-gnns_amd/hostcomm.py (the transport of the several-ranks-on-one-GPU runs, tests/test_gpu_multirank.py): with its
    wrappers installed and host tensors routed through the staging code -- all_to_all / all_gather / all_reduce on staged copies,
    reduce_scatter as all_reduce + this rank's block, async calls completed on return -- the sharded steps reproduce the oracle."""
    hp = dict(max_samples=96, host_staged=True)
    if mode == "nce":
        hp.update(static_sigmas=6.0)        # _GatherPadded's backward takes the reduce_scatter branch under hostcomm
    if mode == "lpw":
        hp.update(kernel="cosine", beta=100.0)
    losses, logits, accs, n_halo, _comm = _spawn(_worker, world, gnn, mode, hp)
    ref_losses, ref_logits, ref_accs = _reference_run(gnn, mode, hp=hp)
    np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(logits, ref_logits.numpy(), rtol=1e-4, atol=1e-5)
    assert n_halo > 0


@pytest.mark.parametrize("gnn,mode,world,max_samples", [("gcn", "nce", 3, 40), ("gcn", "gpw", 2, 40), ("gcn", "nce", 3, -300)])
def test_static_shape_steps_issue_one_consistent_collective_program(gnn, mode, world, max_samples):
    """Same property as test_every_rank_issues_the_same_collective_sequence for the static layout, plus what makes it capturable:
    the sequence AND the sizes are identical from step to step (the dynamic layout's all-gather / all-reduce sizes move with the draw)."""
    import efficient_gnns_amd.dist as DD
    hp = dict(max_samples=max_samples, static_sigmas=6.0)
    if max_samples < 0:
        hp.update(max_samples=64, train_ids_below=-max_samples)
    if mode == "gpw":
        hp.update(kernel="cosine", beta=100.0)
    everyone = _spawn(_trace_worker, world, gnn, mode, hp)
    for step in range(3):
        bad = DD.consistent_collectives([everyone[r][step] for r in range(world)])
        assert bad is None, f"step {step}: {bad}"
    for r in range(world):
        sig = [[(rec[0], rec[1], rec[2]) for rec in everyone[r][step]] for step in range(3)]
        assert sig[0] == sig[1] == sig[2], f"rank {r}: the collective program changes between steps"


def test_static_sample_layout():
    """StaticSample.fill: the permutation addresses each sampled row at (owner, index within the owner's block) in draw order, the
    padded id list holds the owned sampled rows first and DISTINCT unsampled rows after them, an oversized draw is refused."""
    import types
    import efficient_gnns_amd.dist as DD
    g = torch.Generator().manual_seed(0)
    n, world, ntr, S = 1000, 3, 420, 100
    tr = torch.randperm(n, generator=g)[:ntr]
    per = (n + world - 1) // world
    owner = torch.div(tr, per, rounding_mode="floor")
    samples = []
    for rank in range(world):
        mine = owner == rank
        localpos = torch.zeros_like(tr)
        localpos[mine] = torch.arange(int(mine.sum()))
        prob = types.SimpleNamespace(n_train_global=ntr, world=world, rank=rank, train_owner=owner, train_localpos=localpos, device="cpu")
        samples.append(DD.StaticSample(prob, S))
    cap = samples[0].cap
    assert all(s.cap == cap for s in samples) and cap < S
    np.random.seed(4)
    pick = samples[0].draw()
    assert all(s.fill(pick) for s in samples)
    perm = samples[0].perm_host.numpy()
    assert all(np.array_equal(s.perm_host.numpy(), perm) for s in samples)
    assert len(set(perm.tolist())) == S and perm.min() >= 0 and perm.max() < world * cap
    own = owner.numpy()[pick]
    for rank, s in enumerate(samples):
        k = perm[own == rank] - rank * cap
        assert np.array_equal(k, np.arange(k.size)), "draw order inside the owner's block"
        idx = s.idx_host.numpy()
        cnt = int((own == rank).sum())
        assert s.counts[rank] == cnt and idx.size == s.m
        assert np.array_equal(idx[:cnt], s._localpos[pick[own == rank]])
        assert len(set(idx.tolist())) == idx.size and idx.max() < s.n_mine
    tiny = DD.StaticSample(types.SimpleNamespace(n_train_global=ntr, world=world, rank=0, train_owner=owner,
                                                 train_localpos=torch.from_numpy(samples[0]._localpos), device="cpu"), S, sigmas=-50.0)
    assert tiny.cap == 1 and not tiny.fill(pick)


def test_community_order_lowers_the_halo_and_keeps_the_problem():
    """dist.locality_order / reorder_nodes on the synthetic community graph (ids shuffled, as datasets come): cutting the 8 node
    ranges from the community order lowers the halo rows by >= 40 %; the relabelled problem is the same problem (oracle loss and
    gradients of one GCN + KD step agree); the Chung-Lu graph (no locality) keeps its given order."""
    import efficient_gnns_amd.data as D
    import efficient_gnns_amd.dist as DD
    import oracle.models as OM
    import oracle.sparse as OS
    d = D.arxiv_like(scale=0.1, seed=1, graph="local")
    perm, before, after = DD.locality_order(d, 8)
    assert perm is not None and sum(after) <= 0.6 * sum(before), (before, after)
    assert sorted(perm.tolist()) == list(range(d.num_nodes))
    d2 = DD.reorder_nodes(d, perm)
    assert all(torch.equal(d.y[d.split_idx[k]], d2.y[d2.split_idx[k]]) for k in d.split_idx), "split lists keep their order"
    rowptr, col, _ = d2.adj_t.csr()
    assert DD.halo_rows_per_rank(rowptr, col, d.num_nodes, 8) == after

    def step(data):
        rp, c, _ = data.adj_t.csr()
        adj = OS.SparseTensor(rowptr=rp, col=c, sparse_sizes=data.adj_t.sparse_sizes())
        torch.manual_seed(0)
        m = OM.GCN(data.num_features, 32, data.num_classes, 3, 0.0)
        opt = torch.optim.Adam(m.parameters(), lr=0.01)
        hp = dict(alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=64, kernel="cosine")
        losses = OM.train_step(m, data.x, adj, data.y, data.split_idx["train"], opt, "kd", hp, data.teacher_out_feat, data.teacher_logits)
        return losses, [p.grad.clone() for p in m.parameters()]
    (l1, g1), (l2, g2) = step(d), step(d2)
    np.testing.assert_allclose(l1, l2, rtol=1e-5)
    scale = max(float(a.abs().max()) for a in g1)     # biases in front of a BatchNorm carry rounding noise only: one absolute bar
    for a, b in zip(g1, g2):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-3, atol=1e-5 * scale)
    flat = D.arxiv_like(scale=0.05, seed=1)
    perm_f, bf, af = DD.locality_order(flat, 8)
    assert perm_f is None or sum(af) < sum(bf)


def _agg_forms_worker(rank, world, port, q):
    """Operator level: ``ShardedAdj.aggregate`` in halo and in column-sliced form (forward and backward, sum with A^ values / valueless sum /
    mean, with a bias) against the dense product, on a random graph with empty rows, a hub row and uneven node ranges."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _patch_ops_with_oracle()
        import efficient_gnns_amd.dist as DD
        g = torch.Generator().manual_seed(11)
        n, K = 203, 16 * world                                   # 203 rows: the last range is shorter
        dense = (torch.rand(n, n, generator=g) < 0.03).float()
        dense[5] = 0.0                                           # an empty row
        dense[:, 7] = 0.0                                        # a node nobody references
        dense[11, :150] = 1.0                                    # a hub row
        dense.fill_diagonal_(0.0)
        rows, cols = dense.nonzero(as_tuple=True)
        rowptr = torch.zeros(n + 1, dtype=torch.int64)
        torch.cumsum(torch.bincount(rows, minlength=n), 0, out=rowptr[1:])
        lo, hi, _ = DD.node_range(n, world, rank)
        e0, e1 = int(rowptr[lo]), int(rowptr[hi])
        sadj = DD.ShardedAdj(rowptr[lo:hi + 1] - e0, cols[e0:e1], n, world, rank, "cpu", None, with_gcn=True)
        # dense references: raw A (sum / mean) and A^ = D^-1/2 (A + I) D^-1/2
        ahat = dense + torch.eye(n)
        dinv = ahat.sum(1).pow(-0.5)
        ahat = dinv[:, None] * ahat * dinv[None, :]
        cnt = dense.sum(1).clamp(min=1)
        X = torch.randn(n, K, generator=g)
        G = torch.randn(n, K, generator=g)
        bias = torch.randn(K, generator=g)
        worst = 0.0
        for form in ("halo", "sliced"):
            for kind, A, adj, kw in (("gcn", ahat, sadj.gcn_normalized(), dict(reduce="sum")),
                                     ("sum", dense, sadj, dict(reduce="sum", valueless=True)),
                                     ("mean", dense / cnt[:, None], sadj, dict(reduce="mean", valueless=True))):
                adj.agg_mode = form
                x = X[lo:hi].clone().requires_grad_(True)
                b = bias.clone().requires_grad_(True)
                y = adj.aggregate(x, bias=b, **kw)
                y.backward(G[lo:hi])
                xr = X.clone().requires_grad_(True)
                br = bias.clone().requires_grad_(True)
                (A @ xr + br)[lo:hi].backward(G[lo:hi])
                gx_all = xr.grad.clone()                         # this rank's rows contribute to every source row: sum over ranks
                dist.all_reduce(gx_all)
                gb_ref = br.grad
                for got, ref in ((y.detach(), (A @ X + bias)[lo:hi]), (x.grad, gx_all[lo:hi]), (b.grad, gb_ref)):
                    worst = max(worst, float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)))
                assert adj.sliced_pays(K) == (form == "sliced"), (form, kind)
        out = [None] * world
        dist.all_gather_object(out, worst)
        if rank == 0:
            q.put(out)
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    _quiet_exit()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_aggregate_forms_equal_the_dense_product_forward_and_backward(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_agg_forms_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    worst = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert max(worst) < 2e-5, worst
