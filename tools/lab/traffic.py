#!/usr/bin/env python3
"""HBM-side traffic per aggregation call from the FETCH_SIZE / WRITE_SIZE passes of tools/lab/pmc.sh (separate rocprofv3 --pmc
passes, counters only), calibrated on the two 256 MiB device-to-device copies the lab driver issues in the same pass
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-byte requests at 64 B on gfx950; WRITE_SIZE is calibrated likewise).

    python tools/lab/traffic.py gpurun_out/r03/pmc chunglu default 367506008 > profiles/spmm_traffic.json

The JSON records ``lib_sha16`` (build.source_stamp(): SHA-256 over the compile flags and the sources of the libegnn_hip.so the passes ran
on -- the binary embeds its build time, its own hash would not survive a rebuild of the same sources): bench.py reports the traffic
only when it runs on those kernels."""
import csv
import hashlib
import json
import os
import sys

d, graph, var, alg = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
COPY = 256 << 20
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402  (repo root: the one definition of the stamp)
out = {"graph": graph, "variant": var, "algorithmic_bytes_per_call": alg, "lib_sha16": bench.lib_sha16()}
total = 0.0
for key, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = [r for r in csv.DictReader(open(os.path.join(d, f"{graph}_{var}_{key}.csv"))) if r["Counter_Name"] == counter]
    per_disp = {}
    for r in rows:
        e = per_disp.setdefault(r["Dispatch_Id"], [r["Kernel_Name"], 0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        e[1] += float(r["Counter_Value"])
    copies = sorted((v for k, v, _ in per_disp.values() if "copyBuffer" in k), reverse=True)[:2]   # the two 256 MiB calibration copies
    unit = COPY / (sum(copies) / len(copies))
    kern = {}
    for k, v, ns in per_disp.values():
        if "spmm" in k:
            name = "spmm_blk_kernel" if "spmm_blk_kernel" in k else ("spmm_combine_kernel" if "combine" in k else k[:40])
            a = kern.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += v
            a[2] += ns * 1e-3
    per_call = sum(a[1] / a[0] for a in kern.values()) * unit
    out[key] = dict(counter=counter, calibration_copy_counter=copies, calibration_copy_bytes=COPY, bytes_per_unit=unit,
                    counter_per_kernel={k: a[1] / a[0] for k, a in kern.items()}, launches={k: a[0] for k, a in kern.items()},
                    avg_us_per_kernel_profiled={k: a[2] / a[0] for k, a in kern.items()}, bytes_per_call=per_call)
    total += per_call
out["hbm_bytes_per_call"] = total
out["traffic_over_algorithmic"] = total / alg
json.dump(out, sys.stdout, indent=1)
