#!/usr/bin/env python3
"""Copies what `tools/evidence.sh <tag>` left under gpurun_out/<tag>/evidence/ into profiles/ (the tracked names) and prints the
figures DESIGN.md section 8 quotes.   python tools/collect_evidence.py r03"""
import json
import os
import re
import shutil
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
E = os.path.join(ROOT, "gpurun_out", tag, "evidence")
P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    s = os.path.join(E, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
        return True
    print("MISSING / EMPTY:", src)
    return False


for f in ("bench_line.json", "bench_line_f32pipe.json", "bench_line_r02_paths.json", "bench_kernel_stats.csv", "bench_domain_stats.csv",
          "epoch_kernels.txt", "config_benches.txt", "kernel_bench.jsonl", "pytest_gpu.log", "smoke.log"):
    cp(f, f"{tag}_{f}")
for f in ("arxiv", "arxiv_eager", "arxiv_gpw", "arxiv_lpw", "mag"):
    cp(f"sharded_1rank_{f}.json", f"{tag}_sharded_1rank_{f}.json")
cp("spmm_traffic.json", "spmm_traffic.json")
cp("spmm_traffic_local.json", "spmm_traffic_local.json")
os.makedirs(os.path.join(P, f"{tag}_pmc"), exist_ok=True)
for f in os.listdir(os.path.join(E, "pmc")) if os.path.isdir(os.path.join(E, "pmc")) else []:
    if f.endswith(".csv"):
        shutil.copy(os.path.join(E, "pmc", f), os.path.join(P, f"{tag}_pmc", f))

for f in ("bench_line.json", "bench_line_f32pipe.json", "bench_line_r02_paths.json"):
    d = json.loads(open(os.path.join(E, f)).read().strip().splitlines()[-1])
    r, m = d["roofline"], d["roofline_mfma"]
    print(f, d["value"], d["ms_per_step"], d.get("phases_ms"))
    print("   spmm", r["avg_launch_us"], r["achieved"], r["frac"], "gather", r.get("gather_GBs"), r.get("gather_ceiling_GBs"), r.get("gather_ceiling_us"),
          r.get("frac_of_gather_ceiling"), "traffic", r.get("traffic"))
    print("   mfma", m["ms_per_step"], m["achieved"], m["frac"])
    if d.get("roofline_local"):
        print("   local", d["roofline_local"]["avg_launch_us"], d["roofline_local"]["frac"], d["roofline_local"].get("traffic"))
    if d.get("cpu_baseline"):
        print("   cpu", d["cpu_baseline"]["value"], d["value"] / d["cpu_baseline"]["value"])
    if d.get("parity"):
        print("   parity", d["parity"]["ok"], d["parity"]["max_rel_err"], d["parity"].get("max_rel_err_vs_f64"))
print(open(os.path.join(E, "epoch_kernels.txt")).readline().strip())
cat, n = Counter(), Counter()
for line in open(os.path.join(E, "epoch_kernels.txt")):
    m = re.match(r"\s+([\d.]+)\s+(.*)", line)
    if not m:
        continue
    us, k = float(m.group(1)), m.group(2)
    if "nce" in k or ("pack_planes_kernel<128, 32>" in k and us > 15) or "pack_planes_kernel<256, 16>" in k:
        c = "gcrd"
    elif "spmm" in k or "rows_add" in k:
        c = "aggregation"
    elif "tail_" in k:
        c = "tail"
    elif "gemm" in k or "skinny" in k or "splitk" in k or "pack_planes" in k:
        c = "gemm"
    elif "bn_" in k or "colsum" in k:
        c = "bn"
    elif "at::native" in k or "rocclr" in k:
        c = "torch"
    else:
        c = "loss/rows"
    cat[c] += us
    n[c] += 1
print({c: (round(v, 1), n[c]) for c, v in cat.most_common()})
for line in open(os.path.join(E, "config_benches.txt")):
    if line.startswith("--"):
        print(line.strip(), end="  ")
    elif line.startswith("{"):
        m = re.search(r'"value": ([\d.]+).*?"ms_per_step": ([\d.]+)', line)
        ok = re.search(r'"parity_ok": (\w+)', line)
        print(m.group(1), m.group(2), ok.group(1) if ok else "")
for f in ("arxiv", "arxiv_eager", "arxiv_gpw", "arxiv_lpw", "mag"):
    pth = os.path.join(E, f"sharded_1rank_{f}.json")
    if os.path.exists(pth) and os.path.getsize(pth):
        d = json.loads(open(pth).read().strip().splitlines()[-1])
        print("sharded", f, d["value"], d["ms_per_step"])
for line in open(os.path.join(E, "kernel_bench.jsonl")):
    d = json.loads(line)
    if d["what"] == "gemm":
        print(d["shape"], d["hip_us"], d["hip_TF"], d["blas_us"])
    elif d["what"] == "nce":
        print("nce", d["S"], d["fwd_us"], d["fwd_bwd_us"], d["fwd_bwd_TF"], d["torch_fwd_bwd_us"])
    elif d["what"] == "spmm" and d.get("variant") in ("gcn_sum_val", "sage_mean", "torch.sparse.mm(rocSPARSE)"):
        print(d["variant"], d["K"], d["us"])
print(open(os.path.join(E, "pytest_gpu.log")).read().strip().splitlines()[-1])
for f in ("spmm_traffic.json", "spmm_traffic_local.json"):
    d = json.load(open(os.path.join(E, f)))
    print(f, d["lib_sha16"], d["hbm_bytes_per_call"], d["traffic_over_algorithmic"])
