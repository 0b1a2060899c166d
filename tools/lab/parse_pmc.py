#!/usr/bin/env python3
"""Tabulate the CSVs of tools/lab/pmc.sh: per (graph, variant, kernel) the per-launch average of every counter.
    python tools/lab/parse_pmc.py gpurun_out/r02/pmc > profiles/r02_spmm_pmc.json"""
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
res = {}
for f in sorted(glob.glob(os.path.join(d, "*.csv"))):
    base = os.path.basename(f)[:-4]
    graph, rest = base.split("_", 1)
    var, group = rest.rsplit("_", 1)
    if group in ("time", "inst"):   # sq_time / sq_inst
        var, g2 = var.rsplit("_", 1)
        group = g2 + "_" + group
    per = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r"(spmm_\w+|bn_\w+|copyBuffer\w*)", r["Kernel_Name"])
        kn = m.group(1) if m else r["Kernel_Name"][:40]
        k = per.setdefault(kn, {})
        k.setdefault("_disp", set()).add(r["Dispatch_Id"])
        k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        k.setdefault("_ns", {})[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for kn, k in per.items():
        n = len(k.pop("_disp"))
        ns = k.pop("_ns")
        out = res.setdefault(graph, {}).setdefault(var, {}).setdefault(kn, {})
        out.setdefault("launches", n)
        out.setdefault("avg_us_profiled", {})[group] = round(sum(ns.values()) / n * 1e-3, 1)
        for cn, v in k.items():
            out[cn] = v / n
json.dump(res, sys.stdout, indent=1)
