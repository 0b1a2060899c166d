#!/usr/bin/env python3
"""Copies what tools/r06/evidence_r06.sh left under gpurun_out/r06/evidence/ into profiles/ (tracked names).  Missing / empty files are
reported, not fatal.      python tools/r06/collect.py"""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E, P = os.path.join(ROOT, "gpurun_out", "r06", "evidence"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    s = os.path.join(E, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
        return True
    print("MISSING / EMPTY:", src)
    return False


for f in ("bench_line.json", "bench_line_driver_cmd.json", "bench_line_f32pipe.json", "bench_kernel_stats.csv", "bench_domain_stats.csv",
          "config_benches.txt", "pytest_gpu.log", "smoke.log", "replay_gaps.txt", "multirank_w2.json", "multirank_w4.json",
          "multirank_fullsize_w2.json", "multirank_fullsize_sliced_w4.json", "one_device_2ranks_halo_bench.json",
          "one_device_2ranks_sliced_bench.json", "one_device_4ranks_halo_bench.json", "one_device_4ranks_sliced_bench.json"
          ):
    cp(f, "r06_" + f)
cp(os.path.join("epoch_gcn_nce", "epoch.txt"), "r06_epoch_kernels.txt")
for f in ("arxiv", "arxiv_eager", "arxiv_gpw", "arxiv_lpw", "mag"):
    cp(f"sharded_1rank_{f}.json", f"r06_sharded_1rank_{f}.json")
cp("spmm_traffic.json", "spmm_traffic.json")
cp("spmm_traffic_local.json", "spmm_traffic_local.json")
if os.path.isdir(os.path.join(E, "pmc")):
    os.makedirs(os.path.join(P, "r06_pmc"), exist_ok=True)
    for f in os.listdir(os.path.join(E, "pmc")):
        if f.endswith(".csv") and os.path.getsize(os.path.join(E, "pmc", f)) > 0:
            shutil.copy(os.path.join(E, "pmc", f), os.path.join(P, "r06_pmc", f))
