#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM traffic of the SpMM call.

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB-like units
of the fabric request tallies and FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950, so both counters are
CALIBRATED here on a copy of known size taken in the same pass (bytes_per_unit = known bytes / counter value)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out = sys.argv[2]
COPY_BYTES = 64 * 1024 * 1024 * 4


def load(prefix, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", f"*{prefix}*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"]), int(r["Grid_Size"]),
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
    return rows


res = {}
for prefix, counter, key in (("fetch", "FETCH_SIZE", "read"), ("write", "WRITE_SIZE", "write")):
    rows = load(prefix, counter)
    # calibration = the three 256 MiB dst.copy_(src) launches (the largest __amd_rocclr_copyBuffer dispatches)
    big = max((g for k, _, g, _ in rows if "copyBuffer" in k), default=0)
    copies = [v for k, v, g, _ in rows if "copyBuffer" in k and g == big]
    per, dur = {}, {}
    for k, v, _, us in rows:
        if "spmm" in k:
            name = k.split("::")[1].split("<")[0] if "::" in k else k
            per.setdefault(name, []).append(v)
            dur.setdefault(name, []).append(us)
    calib = COPY_BYTES / (sum(copies) / len(copies)) if copies else None
    res[key] = dict(counter=counter, calibration_copy_counter=copies, calibration_copy_bytes=COPY_BYTES, bytes_per_unit=calib,
                    counter_per_kernel={k: sum(v) / len(v) for k, v in per.items()},
                    avg_us_per_kernel={k: sum(v) / len(v) for k, v in dur.items()},
                    launches={k: len(v) for k, v in per.items()})
    if calib:
        res[key]["bytes_per_call"] = sum(sum(v) / len(v) for v in per.values()) * calib
if "bytes_per_call" in res.get("read", {}) and "bytes_per_call" in res.get("write", {}):
    res["hbm_bytes_per_call"] = res["read"]["bytes_per_call"] + res["write"]["bytes_per_call"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
