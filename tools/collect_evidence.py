#!/usr/bin/env python3
"""Copies what an evidence session (`tools/evidence.sh <tag>`, `tools/r05/evidence_r05.sh`) left under gpurun_out/<tag>/evidence/ into
profiles/ (the tracked names) and prints the figures DESIGN.md section 8 quotes.  Missing files are reported, not fatal.
    python tools/collect_evidence.py r05"""
import json
import os
import re
import shutil
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
E = os.path.join(ROOT, "gpurun_out", tag, "evidence")
P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    s = os.path.join(E, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
        return True
    print("MISSING / EMPTY:", src)
    return False


def last_json(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


for f in ("bench_line.json", "bench_line_driver_cmd.json", "bench_line_f32pipe.json", "bench_kernel_stats.csv", "bench_domain_stats.csv",
          "config_benches.txt", "kernel_bench.jsonl", "pytest_gpu.log", "smoke.log"):
    cp(f, f"{tag}_{f}")
if not cp("epoch_kernels.txt", f"{tag}_epoch_kernels.txt"):
    cp(os.path.join("epoch_gcn_nce", "epoch.txt"), f"{tag}_epoch_kernels.txt")
cp(os.path.join("epoch_compare", "compare.txt"), f"{tag}_sharded_vs_single_epoch.txt")
for f in ("arxiv", "arxiv_eager", "arxiv_gpw", "arxiv_lpw", "mag"):
    cp(f"sharded_1rank_{f}.json", f"{tag}_sharded_1rank_{f}.json")
for w in (2, 4):
    cp(f"multirank_w{w}.json", f"{tag}_multirank_w{w}.json")
    cp(f"one_device_{w}ranks_bench.json", f"{tag}_one_device_{w}ranks_bench.json")
cp("multirank_fullsize_w2.json", f"{tag}_multirank_fullsize_w2.json")
cp("spmm_traffic.json", "spmm_traffic.json")
cp("spmm_traffic_local.json", "spmm_traffic_local.json")
if os.path.isdir(os.path.join(E, "pmc")):
    os.makedirs(os.path.join(P, f"{tag}_pmc"), exist_ok=True)
    for f in os.listdir(os.path.join(E, "pmc")):
        if f.endswith(".csv"):
            shutil.copy(os.path.join(E, "pmc", f), os.path.join(P, f"{tag}_pmc", f))

for f in ("bench_line.json", "bench_line_driver_cmd.json", "bench_line_f32pipe.json"):
    pth = os.path.join(E, f)
    if not (os.path.exists(pth) and os.path.getsize(pth)):
        continue
    d = last_json(pth)
    r, m, g = d["roofline"], d["roofline_mfma"], d.get("roofline_gemm") or {}
    print(f, d["value"], d["ms_per_step"], d.get("repeat_blocks_ms_per_step"))
    print("   spmm", r["avg_launch_us"], r["achieved"], r["frac"], "gather", r.get("gather_GBs"), r.get("gather_ceiling_GBs"), r.get("gather_ceiling_us"),
          r.get("frac_of_gather_ceiling"), "traffic", r.get("traffic"))
    print("   mfma", m["ms_per_step"], m["achieved"], m["frac"], "  gemm", g.get("ms_per_step"), g.get("achieved"), g.get("frac"))
    for k, v in (g.get("by_shape") or {}).items():
        print("        ", k, v)
    if d.get("roofline_local"):
        print("   local", d["roofline_local"]["avg_launch_us"], d["roofline_local"]["frac"], d["roofline_local"].get("traffic"))
    if d.get("cpu_baseline"):
        print("   cpu", d["cpu_baseline"]["value"], d["value"] / d["cpu_baseline"]["value"])
    if d.get("parity"):
        t = d["parity"].get("trajectory_dropout") or {}
        print("   parity", d["parity"]["ok"], d["parity"]["max_rel_err"], d["parity"].get("max_rel_err_vs_f64"), "trajectory", t.get("ok"), t.get("max_rel_err"))
    if d.get("reference_loop"):
        rl = d["reference_loop"]
        print("   reference loop", rl.get("epochs_per_s"), rl.get("fraction_of_package_loop"), (rl.get("plain_torch_modules") or {}).get("epochs_per_s"))
ek = os.path.join(P, f"{tag}_epoch_kernels.txt")
if os.path.exists(ek):
    print(open(ek).readline().strip())
    cat, n = Counter(), Counter()
    for line in open(ek):
        if line.startswith("# ---- by kernel"):
            break
        m = re.match(r"\s+([\d.]+)\s+(.*)", line)
        if not m:
            continue
        us, k = float(m.group(1)), m.group(2)
        if "nce" in k or ("pack_planes_kernel<128, 32>" in k and us > 15) or ("pack_planes_kernel<256, 16>" in k and us > 15):
            c = "gcrd"
        elif "spmm" in k or "rows_add" in k:
            c = "aggregation"
        elif "tail_" in k:
            c = "tail"
        elif "gemm" in k or "skinny" in k or "splitk" in k or "pack_planes" in k:
            c = "gemm"
        elif "bn_" in k or "colsum" in k:
            c = "bn"
        elif "at::native" in k or "rocclr" in k or "elementwise" in k or "multi_tensor" in k:
            c = "torch"
        else:
            c = "loss/rows"
        cat[c] += us
        n[c] += 1
    print({c: (round(v, 1), n[c]) for c, v in cat.most_common()})
cb = os.path.join(E, "config_benches.txt")
if os.path.exists(cb):
    for line in open(cb):
        if line.startswith("--"):
            print(line.strip(), end="  ")
        elif line.startswith("{"):
            m = re.search(r'"value": ([\d.]+).*?"ms_per_step": ([\d.]+)', line)
            ok = re.search(r'"parity_ok": (\w+)', line)
            print(m.group(1), m.group(2), ok.group(1) if ok else "")
for f in ("arxiv", "arxiv_eager", "arxiv_gpw", "arxiv_lpw", "mag"):
    pth = os.path.join(E, f"sharded_1rank_{f}.json")
    if os.path.exists(pth) and os.path.getsize(pth):
        d = last_json(pth)
        print("sharded", f, d["value"], d["ms_per_step"])
for w in (2, 4):
    pth = os.path.join(E, f"multirank_w{w}.json")
    if os.path.exists(pth):
        r = json.load(open(pth))
        print(f"multirank world {w}: {sum(1 for e in r.values() if e['ok'])} / {len(r)} cases ok; worst loss / logit deviation (fraction of the bar): "
              f"{max(e['loss_err_in_bars'] for e in r.values()):.3f} / {max(e['logit_err_in_bars'] for e in r.values()):.3f}")
pth = os.path.join(E, "multirank_fullsize_w2.json")
if os.path.exists(pth):
    for k, e in json.load(open(pth)).items():
        print("full size", k, e["ok"], round(e["loss_err_in_bars"], 3), round(e["logit_err_in_bars"], 3), [i["n_halo"] for i in e["per_rank"]])
pl = os.path.join(E, "pytest_gpu.log")
if os.path.exists(pl):
    print(open(pl).read().strip().splitlines()[-1])
for f in ("spmm_traffic.json", "spmm_traffic_local.json"):
    pth = os.path.join(E, f)
    if os.path.exists(pth) and os.path.getsize(pth):
        d = json.load(open(pth))
        print(f, d["lib_sha16"], d["hbm_bytes_per_call"], d["traffic_over_algorithmic"])
