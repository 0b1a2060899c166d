#!/usr/bin/env python3
"""Per-kernel timings on one MI355X (HIP events on the launch stream), written as JSON lines.

    python tools/kernel_bench.py [--out gpurun_out/kernel_bench.jsonl] [--quick]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import efficient_gnns_amd as E  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
import efficient_gnns_amd.ops as ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def emit(f, **kw):
    line = json.dumps(kw)
    print(line, flush=True)
    f.write(line + "\n")
    f.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kernel_bench.jsonl"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma list of sections: spmm,gemm,nce,misc")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    want = lambda sec: not only or sec in only  # noqa: E731
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    f = open(args.out, "w")
    emit(f, what="device", name=torch.cuda.get_device_name(0), lib=E._lib.build_info())

    data = D.arxiv_like(1.0, seed=0, with_teacher=False)
    adj = data.adj_t.to(DEV)
    gn = E.gcn_norm(adj)
    n = data.num_nodes
    # ---- SpMM ------------------------------------------------------------------------------------
    for K in ((256, 128, 40) if want('spmm') else ()):
        x = torch.randn(n, K, device=DEV)
        for label, a, kw in (("gcn_sum_val", gn, {}), ("sage_mean", adj, {"reduce": "mean"}),
                             ("gcn_sum_val_noplan", gn, {"use_plan": False})):
            t = timeit(lambda: ops.spmm_raw(a, x, **kw))
            nbytes = a.spmm_algorithmic_bytes(K)
            emit(f, what="spmm", variant=label, K=K, nnz=a.nnz(), us=round(t * 1e6, 2), alg_MB=round(nbytes / 1e6, 1),
                 GBs=round(nbytes / t / 1e9, 1), frac_8TBs=round(nbytes / t / 8e12, 4),
                 gather_GBs=round(a.nnz() * K * 4 / t / 1e9, 1))
        a64 = E.SparseTensor(rowptr=gn.csr()[0], col=gn.csr()[1], value=gn.csr()[2], sparse_sizes=gn.sparse_sizes())
        a64._struct["idx"] = (a64._rowptr, a64._col, 64)
        t = timeit(lambda: ops.spmm_raw(a64, x))
        emit(f, what="spmm", variant="gcn_sum_val_int64", K=K, us=round(t * 1e6, 2))
    # torch baseline for the same op (rocSPARSE via torch.sparse) for context
    try:
        if not want('spmm') or args.quick:
            raise RuntimeError('skipped')
        csr = torch.sparse_csr_tensor(gn.csr()[0], gn.csr()[1], gn.csr()[2], size=(n, n))
        x = torch.randn(n, 256, device=DEV)
        t = timeit(lambda: torch.sparse.mm(csr, x), iters=5, warmup=1)
        emit(f, what="spmm", variant="torch.sparse.mm(rocSPARSE)", K=256, us=round(t * 1e6, 2))
    except Exception as ex:  # noqa: BLE001
        emit(f, what="spmm", variant="torch.sparse.mm", error=str(ex)[:200])
    # HBM copy ceiling on this box (float4 copy of 512 MB)
    src = torch.empty(128 * 1024 * 1024, device=DEV)
    dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src), iters=10)
    emit(f, what="hbm_copy", GBs=round(2 * src.numel() * 4 / t / 1e9, 1))
    del src, dst

    # ---- dense GEMM: hand MFMA kernel vs rocBLAS ----------------------------------------------------
    n_tr = 90941
    shapes = [("xW1 NN", n, 256, 128, False, False), ("xW2 NN", n, 256, 256, False, False), ("xW3 NN", n, 40, 256, False, False),
              ("proj_s NT", n_tr, 256, 256, False, True), ("proj_t NT", n_tr, 256, 750, False, True),
              ("dX NT", n, 256, 256, False, True), ("dW TN", 256, 256, n, True, False), ("dWt TN", 256, 750, n_tr, True, False),
              # the class-count-wide backward forms of the output layer (csrc/gemm_skinny.hip)
              ("dX3 NT", n, 256, 40, False, True), ("dW3 TN", 256, 40, n, True, False)]
    for label, M, N, K, ta, tb in (shapes if want('gemm') else ()):
        a = torch.randn((K, M) if ta else (M, K), device=DEV)
        b = torch.randn((N, K) if tb else (K, N), device=DEV)
        flops = 2.0 * M * N * K
        t1 = timeit(lambda: ops.gemm_raw(a, b, ta, tb), iters=10)
        t2 = timeit(lambda: (a.t() if ta else a) @ (b.t() if tb else b), iters=10)
        emit(f, what="gemm", shape=label, M=M, N=N, K=K, hip_us=round(t1 * 1e6, 1), hip_TF=round(flops / t1 / 1e12, 1),
             blas_us=round(t2 * 1e6, 1), blas_TF=round(flops / t2 / 1e12, 1))

    # ---- G-CRD ----------------------------------------------------------------------------------------
    for S in (() if not want('nce') else ((8192,) if args.quick else (8192, 16384))):
        P = 256
        torch.cuda.empty_cache()   # the [S,S] score matrix: start from an empty pool (after the earlier sections the caching allocator was seen
        #                            re-shaping its pool once inside the timed loop: one ~90 ms host stall read as 8-19 ms per iteration)
        fh = torch.nn.functional.normalize(torch.randn(S, P, device=DEV), dim=-1)
        th = torch.nn.functional.normalize(torch.randn(S, P, device=DEV), dim=-1)
        t_f = timeit(lambda: ops.nce_unit(fh, th, 0.075), iters=5, warmup=2)

        def fb():
            a = fh.clone().requires_grad_(True)
            b = th.clone().requires_grad_(True)
            ops.nce_unit(a, b, 0.075).backward()
        t_fb = timeit(fb, iters=5, warmup=4)

        def torch_fb():
            a = fh.clone().requires_grad_(True)
            b = th.clone().requires_grad_(True)
            z = a @ b.t() / 0.075
            torch.nn.functional.cross_entropy(z, torch.arange(S, device=DEV)).backward()
        t_t = timeit(torch_fb, iters=5, warmup=2)
        fl = 2.0 * S * S * P
        emit(f, what="nce", S=S, P=P, fwd_us=round(t_f * 1e6, 1), fwd_TF=round(fl / t_f / 1e12, 1), fwd_bwd_us=round(t_fb * 1e6, 1),
             fwd_bwd_TF=round(3 * fl / t_fb / 1e12, 1), torch_fwd_bwd_us=round(t_t * 1e6, 1))

    # ---- elementwise neighbours of the convs (next row f-1), torch ops today -----------------------------
    if not want('misc'):
        f.close()
        return
    h = torch.randn(n, 256, device=DEV, requires_grad=True)
    bn = torch.nn.BatchNorm1d(256).to(DEV)
    t = timeit(lambda: torch.nn.functional.dropout(torch.relu(bn(h)), 0.5, True), iters=10)
    emit(f, what="bn_relu_dropout_fwd(torch)", us=round(t * 1e6, 1), alg_MB=round(2 * n * 256 * 4 / 1e6, 1))
    idx = torch.randperm(n, device=DEV)[:n_tr]
    tf = torch.randn(n, 750, device=DEV)
    t = timeit(lambda: tf[idx], iters=10)
    emit(f, what="gather_teacher_rows(torch)", us=round(t * 1e6, 1))
    lg, tl = torch.randn(n_tr, 40, device=DEV, requires_grad=True), torch.randn(n_tr, 40, device=DEV)
    lb = torch.randint(0, 40, (n_tr,), device=DEV)
    t = timeit(lambda: E.kd_criterion(lg, lb, tl)[0].backward(), iters=10)
    emit(f, what="kd_criterion fwd+bwd", us=round(t * 1e6, 1))
    f.close()


if __name__ == "__main__":
    main()
