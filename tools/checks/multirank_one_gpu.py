"""The node-range sharded step (efficient-gnns_amd/dist.py) with world_size > 1 ON THE REAL HIP KERNELS, on a one-GPU box.

RCCL refuses two ranks on one device; ``hostcomm`` carries the step's collectives over gloo through pinned host memory
instead, so W processes can share ``cuda:0``: every rank runs the product kernels on its shard (kernel stand-ins OFF) with
a NON-EMPTY halo.  For every case the ranks run ``sharded_evaluate`` + 3 x ``sharded_train_step`` (eager launches) and rank 0
compares with the single-GPU product path (``models.evaluate`` / ``models.train_step``) on the same graph, weights and NumPy
draws; the ranks' collective traces (``CommTrace``) are checked for one consistent program.

    python tools/checks/multirank_one_gpu.py --world 2 --out gpurun_out/multirank_w2.json [--cases name,name,...]

Exit code 0 and the line MULTIRANK-OK when every case is inside the bars (the same ones as
tests/test_gpu_parity.py::test_sharded_path_with_one_rank_over_rccl_matches_single_gpu_path).  Used by
tests/test_gpu_multirank.py.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def all_cases():
    """name -> dict.  arxiv-shaped cases: {G-CRD in the static layout, G-CRD with every draw overflowing the static capacity (the
    dynamic-shape fallback), GSP in the static layout, SAGE + LSP} x {halo exchange overlapped / blocking} x {node ids as given
    (Chung-Lu: no locality) / community graph with the ranges cut from the community order}; MAG-shaped: SAGE-mean + logit KD."""
    cases = {}
    for order in ("natural", "community"):
        for overlap in (1, 0):
            tag = f"{order}-ov{overlap}"
            cases[f"gcn-nce-static-{tag}"] = dict(gnn="gcn", mode="nce", sigmas=6.0, order=order, overlap=overlap)
            cases[f"gcn-nce-overflow-{tag}"] = dict(gnn="gcn", mode="nce", sigmas=-50.0, order=order, overlap=overlap)
            cases[f"gcn-gpw-static-{tag}"] = dict(gnn="gcn", mode="gpw", sigmas=6.0, order=order, overlap=overlap)
            cases[f"sage-lpw-{tag}"] = dict(gnn="sage", mode="lpw", sigmas=None, order=order, overlap=overlap)
    for overlap in (1, 0):
        cases[f"mag-sage-kd-ov{overlap}"] = dict(gnn="sage", mode="kd", sigmas=None, order="natural", overlap=overlap, workload="mag")
    # every train node in the first 40 % of the ids: the last rank(s) own NO train row (no loss rows, no sampled rows, an empty block in
    # the sample all-gather) and still have to run every halo exchange / SyncBN reduction of the backward -- on the real kernels
    for name, gnn, mode, sig in (("gcn-nce-static", "gcn", "nce", 6.0), ("gcn-kd", "gcn", "kd", None), ("sage-lpw", "sage", "lpw", None),
                                 ("gcn-gpw-static", "gcn", "gpw", 6.0)):
        cases[f"{name}-notrain-tail-ov1"] = dict(gnn=gnn, mode=mode, sigmas=sig, order="natural", overlap=1, train_below=0.4)
    # 5 sampled rows over the ranks in the dynamic layout: ranks that own train rows but NO sampled row (an empty pick under SyncBN,
    # an empty row block of the G-CRD problem) next to ranks that do
    cases["gcn-nce-overflow-5samples-ov1"] = dict(gnn="gcn", mode="nce", sigmas=-50.0, order="natural", overlap=1, max_samples=5)
    cases["gcn-gpw-overflow-5samples-ov1"] = dict(gnn="gcn", mode="gpw", sigmas=-50.0, order="natural", overlap=1, max_samples=5)
    # round 6: the aggregation in COLUMN-SLICED form (dist._SlicedAggregate: the feature columns re-sharded around the aggregation,
    # bytes independent of the halo) forced on every kernel-aligned call -- the 64-wide hidden layers here
    cases["gcn-nce-static-sliced-natural-ov1"] = dict(gnn="gcn", mode="nce", sigmas=6.0, order="natural", overlap=1, agg="sliced")
    cases["sage-lpw-sliced-natural-ov1"] = dict(gnn="sage", mode="lpw", sigmas=None, order="natural", overlap=1, agg="sliced")
    cases["mag-sage-kd-sliced-ov1"] = dict(gnn="sage", mode="kd", sigmas=None, order="natural", overlap=1, workload="mag", agg="sliced")
    cases["gcn-kd-sliced-notrain-tail-ov1"] = dict(gnn="gcn", mode="kd", sigmas=None, order="natural", overlap=1, train_below=0.4, agg="sliced")
    return cases


def full_size_cases():
    """Not part of the pytest run (evidence sessions: --cases full-...): the BASELINE.json configs[1] problem at FULL size (N = 169 343,
    GCN-256 + G-CRD, S = 16 384, P = 256) on node-range shards against the single-GPU path, dropout 0."""
    return {"full-gcn-nce-static-natural-ov1": dict(gnn="gcn", mode="nce", sigmas=6.0, order="natural", overlap=1, scale=1.0, hidden=256, proj=256,
                                                    max_samples=16384, seed=0),
            "full-sage-lpw-natural-ov1": dict(gnn="sage", mode="lpw", sigmas=None, order="natural", overlap=1, scale=1.0, hidden=256, proj=256,
                                              max_samples=16384, seed=0),
            # round 6: the headline problem with the hidden-layer aggregations column-sliced, and BASELINE.json configs[4] at FULL size
            # (N = 1 939 743, 42.2 M entries, SAGE-256 mean + logit KD) in both exchange forms
            "full-gcn-nce-static-sliced-natural-ov1": dict(gnn="gcn", mode="nce", sigmas=6.0, order="natural", overlap=1, scale=1.0, hidden=256,
                                                           proj=256, max_samples=16384, seed=0, agg="sliced"),
            "full-mag-sage-kd-ov1": dict(gnn="sage", mode="kd", sigmas=None, order="natural", overlap=1, workload="mag", mag_scale=1.0, hidden=256,
                                         seed=0),
            "full-mag-sage-kd-sliced-ov1": dict(gnn="sage", mode="kd", sigmas=None, order="natural", overlap=1, workload="mag", mag_scale=1.0,
                                                hidden=256, seed=0, agg="sliced")}


def _problem(case, world, dev):
    import efficient_gnns_amd.data as D
    import efficient_gnns_amd.dist as DD
    note = {}
    if case.get("workload") == "mag":
        d = DD.mag_problem(case.get("mag_scale", 0.05), case.get("seed", 5))             # default: N = 96 987, 2.1 M stored entries
    else:
        d = D.arxiv_like(scale=case.get("scale", 0.02), seed=case.get("seed", 5), graph="local" if case["order"] == "community" else "chunglu")
        if case.get("train_below"):
            tr = d.split_idx["train"]
            d.split_idx["train"] = tr[tr < int(case["train_below"] * d.num_nodes)].clone()
        if case["order"] == "community":
            perm, before, after = DD.locality_order(d, world, dev)
            note = dict(halo_rows_as_given=before, halo_rows_community_order=after, reordered=perm is not None)
            if perm is not None:
                d = DD.reorder_nodes(d, perm)
    return d, note


def _build(case, d, dev):
    import efficient_gnns_amd.models as PM
    hidden, proj = case.get("hidden", 64), case.get("proj", 32)
    torch.manual_seed(0)
    np.random.seed(0)
    model = (PM.GCN if case["gnn"] == "gcn" else PM.SAGE)(d.num_features, hidden, d.num_classes, 3, 0.0).to(dev)
    sp = tp = None
    groups = [{"params": model.parameters(), "lr": 0.01}]
    if case["mode"] in ("nce", "gpw"):
        sp, tp = PM.make_projection(hidden, proj).to(dev), PM.make_projection(d.teacher_out_feat.shape[1], proj).to(dev)
        groups += [{"params": sp.parameters(), "lr": 0.01}, {"params": tp.parameters(), "lr": 0.01}]
    return model, sp, tp, groups


def _hp(case):
    hp = dict(alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=case.get("max_samples", 256), kernel="rbf")
    if case["mode"] in ("gpw", "lpw"):
        hp.update(kernel="cosine", beta=100.0)
    return hp


def _single_gpu(case, d, dev, steps):
    """The single-GPU product path on the same problem (rank 0 only)."""
    import efficient_gnns_amd.models as PM
    hp = _hp(case)
    model, sp, tp, groups = _build(case, d, dev)
    opt = torch.optim.Adam(groups)
    adj = d.adj_t.to(dev)
    x, y = d.x.to(dev), d.y.to(dev)
    split = {k: v.to(dev) for k, v in d.split_idx.items()}
    edge_index = None
    if case["mode"] == "lpw":      # gnn.py:246-250
        from efficient_gnns_amd.utils import subgraph
        edge_index = subgraph(split["train"], torch.stack(adj.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
    tf = d.teacher_out_feat.to(dev) if getattr(d, "teacher_out_feat", None) is not None else None
    if tf is not None:
        from efficient_gnns_amd.ops import pad_pitch
        tf = pad_pitch(tf)              # (what bench.py / the sharded problem hand the kernels: rows behind a 16-byte aligned pitch)
    tl = d.teacher_logits.to(dev) if getattr(d, "teacher_logits", None) is not None else None
    logits, accs = PM.evaluate(model, x, adj, y, split)
    losses = [PM.train_step(model, x, adj, y, split["train"], opt, case["mode"], hp, tf, tl, sp, tp, edge_index) for _ in range(steps)]
    final, _ = PM.evaluate(model, x, adj, y, split)
    return dict(logits=logits.cpu().numpy(), accs=list(accs), losses=[list(l) for l in losses], final=final.cpu().numpy())


def _sharded(case, d, world, rank, dev, steps):
    import efficient_gnns_amd.dist as DD
    hp = _hp(case)
    DD._OVERLAP = bool(case["overlap"])
    DD._AGG_MODE = case.get("agg", "halo")       # (the tests pin the form; "auto" is what bench.py runs)
    model, sp, tp, groups = _build(case, d, dev)
    for m in (model, sp, tp):
        if m is not None:
            DD.swap_batchnorm(m)
    opt = torch.optim.Adam(groups)
    prob = DD.ShardedProblem(d, world, rank, dev, None, need_gcn=(case["gnn"] == "gcn"))
    if case["sigmas"] is not None:
        prob.static_sample = DD.StaticSample(prob, hp["max_samples"], sigmas=case["sigmas"])
    with DD.CommTrace() as trace:
        out, accs = DD.sharded_evaluate(model, prob)
        losses = [DD.sharded_train_step(model, prob, opt, case["mode"], hp, sp, tp) for _ in range(steps)]
        final, _ = DD.sharded_evaluate(model, prob)
    torch.cuda.synchronize()
    info = dict(n_local=prob.adj.plan.n_local, n_halo=prob.adj.plan.n_halo, comm=trace.summary(), n_train_local=int(prob.train_local.numel()))
    if case.get("train_below"):
        assert (info["n_train_local"] == 0) == (rank == world - 1 or prob.lo >= int(case["train_below"] * d.num_nodes)), info
    if case["sigmas"] is not None:
        info["static_cap"] = prob.static_sample.cap
    gathered = [None] * world
    dist.all_gather_object(gathered, (out.cpu().numpy(), final.cpu().numpy(), trace.records, info))
    return dict(accs=list(accs), losses=[list(l) for l in losses], gathered=gathered)


def _worker(rank, world, port, names, out_path, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)                      # every rank on the ONE device
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    try:
        import efficient_gnns_amd.dist as DD
        from efficient_gnns_amd import _lib, hostcomm
        assert not hasattr(_lib, "HOST_STANDINS")      # (round 6: the product has no stand-in switch any more)
        hostcomm.install()
        cases = dict(all_cases(), **full_size_cases())
        report = {}
        for name in names:
            case = cases[name]
            t0 = time.perf_counter()
            d, note = _problem(case, world, dev)
            ref = _single_gpu(case, d, dev, steps) if rank == 0 else None
            dist.barrier()
            got = _sharded(case, d, world, rank, dev, steps)
            if rank != 0:
                continue
            logits = np.concatenate([g[0] for g in got["gathered"]], 0)
            final = np.concatenate([g[1] for g in got["gathered"]], 0)
            records = [g[2] for g in got["gathered"]]
            infos = [g[3] for g in got["gathered"]]
            scale = float(np.abs(ref["logits"]).max())
            logit_err = float(np.max(np.abs(logits - ref["logits"]) / (1e-4 * np.abs(ref["logits"]) + 1e-5 * scale)))   # <= 1: inside rtol 1e-4 + 1e-5 max
            fscale = float(np.abs(ref["final"]).max())
            final_err = float(np.max(np.abs(final - ref["final"])) / fscale)
            la, lb = np.array(got["losses"]), np.array(ref["losses"])
            loss_err = float(np.max(np.abs(la - lb) / (2e-4 * np.abs(lb) + 1e-6)))                                    # <= 1: inside rtol 2e-4 + 1e-6
            min_split = min(int(v.numel()) for v in d.split_idx.values())
            acc_err = float(np.max(np.abs(np.array(got["accs"]) - np.array(ref["accs"]))))
            bad = DD.consistent_collectives(records)
            entry = dict(case=case, world=world, losses=got["losses"], ref_losses=ref["losses"], accs=got["accs"], ref_accs=ref["accs"],
                         logit_err_in_bars=logit_err, loss_err_in_bars=loss_err, acc_abs_err=acc_err, acc_one_node=1.0 / min_split,
                         final_logits_rel_err=final_err, collectives_consistent=bad is None, collectives_mismatch=bad,
                         per_rank=infos, note=note, seconds=round(time.perf_counter() - t0, 2), host_staged=hostcomm.stats())
            entry["ok"] = bool(logit_err <= 1.0 and loss_err <= 1.0 and acc_err <= 1.5 / min_split and bad is None
                               and all(i["n_halo"] > 0 for i in infos)
                               and (all(i["comm"]["halo_all_to_all_bytes_sent"] > 0 for i in infos) if case.get("agg") != "sliced"
                                    else all(i["comm"].get("sliced_exchanges", 0) > 0 for i in infos)))
            report[name] = entry
            with open(out_path, "w") as f:      # after every case: a crash in a later one keeps what has been compared
                json.dump(report, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
            print(f"[multirank w={world}] {name}: ok={entry['ok']} loss_err={loss_err:.3f} logit_err={logit_err:.3f} acc_err={acc_err:.2e} "
                  f"halo={[i['n_halo'] for i in infos]} a2a_sent={[i['comm']['halo_all_to_all_bytes_sent'] for i in infos]} {entry['seconds']} s",
                  flush=True)
        if rank == 0:
            ok = all(e["ok"] for e in report.values()) and len(report) == len(names)
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
    try:
        dist.barrier()
        dist.destroy_process_group()
    finally:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0 if ok else 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--out", required=True)
    ap.add_argument("--cases", default="")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    names = [c for c in args.cases.split(",") if c] or list(all_cases())
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, args.world, port, names, args.out, args.steps)) for r in range(args.world)]
    for p in procs:
        p.start()
    codes = []
    for p in procs:
        p.join(1500)
        codes.append(p.exitcode)
    if any(c != 0 for c in codes):
        print(f"MULTIRANK-FAILED exit codes {codes}")
        sys.exit(1)
    print("MULTIRANK-OK")


if __name__ == "__main__":
    main()
