#!/usr/bin/env python3
"""Runs one rocprofv3 --pmc pass per counter group over tools/spmm_pmc_workload.py and tabulates, per graph and kernel,
the per-launch average of every counter (counters-only passes: --kernel-trace, no other trace domain).

    python tools/pmc_diag.py <out.json> [group ...]        # groups default to all of GROUPS
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {
    "sq_time": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
    "sq_inst": ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"],
    "sq_lvl": ["SQ_INST_LEVEL_VMEM", "SQ_LEVEL_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM_RD", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_BUSY_CU_CYCLES"],
    "tcp1": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"],
    "tcp2": ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
    "tcp3": ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"],
    "tcc1": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    "tcc2": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_TAG_STALL_sum", "TCC_BUSY_sum"],
    "tcc3": ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    "ta": ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TA_BUSY_avr"],
    "td": ["TD_TD_BUSY_sum", "TD_TC_STALL_sum", "TD_LOAD_WAVEFRONT_sum", "TD_SPI_STALL_sum"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}


def run_group(name, counters, outdir):
    d = os.path.join(outdir, name)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", name, "--",
           sys.executable, os.path.join(ROOT, "tools", "spmm_pmc_workload.py")]
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return p.returncode, p.stdout[-1500:], rows


def tabulate(rows):
    """dispatches in order; a marker dispatch (Grid_Size 7777-ish fill) separates graphs; calibration copies come first."""
    by_disp = {}
    for r in rows:
        key = int(r["Dispatch_Id"])
        e = by_disp.setdefault(key, dict(kernel=r["Kernel_Name"], grid=int(r["Grid_Size"]),
                                         us=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, c={}))
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out, gi, seen_marker_for = {}, -1, False
    copies = []
    disp = [by_disp[k] for k in sorted(by_disp)]
    big = max((e["grid"] for e in disp if "copyBuffer" in e["kernel"]), default=0)
    pending = []
    for e in disp:
        k = e["kernel"]
        if "copyBuffer" in k and e["grid"] == big:
            copies.append(e)
        if "spmm" in k:
            pending.append(e)
    # segment the spmm dispatches into graphs: each graph = 1 warm-up call + REPS calls; a call = short_rows (+ combine)
    return disp, copies


def main():
    out = sys.argv[1]
    groups = sys.argv[2:] or list(GROUPS)
    outdir = "/tmp/pmc_diag"
    subprocess.run(["rm", "-rf", outdir])
    reps = int(os.environ.get("EGNN_PMC_REPS", "3"))
    graphs = os.environ.get("EGNN_PMC_GRAPHS", "chunglu,window4096").split(",")
    res = {}
    for g in groups:
        rc, tail, rows = run_group(g, GROUPS[g], outdir)
        print(f"[pmc] group {g}: rc={rc} rows={len(rows)}", flush=True)
        if rc != 0 or not rows:
            print(tail, flush=True)
            res[g] = dict(error=tail[-400:])
            continue
        disp, copies = tabulate(rows)
        # order of spmm calls: per graph (1 warm-up + reps) calls
        calls, cur = [], None
        for e in disp:
            name = e["kernel"]
            if "spmm_short_rows_kernel" in name or "spmm_lds_rows_kernel" in name or "spmm_rows_kernel" in name:
                cur = [e]
                calls.append(cur)
            elif "spmm_combine_kernel" in name and cur is not None:
                cur.append(e)
        per_graph = {}
        for gi, gname in enumerate(graphs):
            mine = calls[gi * (reps + 1) + 1:(gi + 1) * (reps + 1)]   # skip the warm-up call
            agg = {}
            for call in mine:
                for e in call:
                    kn = e["kernel"].split("(")[0].split("::")[-1].split("<")[0]
                    a = agg.setdefault(kn, dict(n=0, us=0.0, c={}))
                    a["n"] += 1
                    a["us"] += e["us"]
                    for cn, v in e["c"].items():
                        a["c"][cn] = a["c"].get(cn, 0.0) + v
            per_graph[gname] = {kn: dict(launches=a["n"], avg_us=a["us"] / max(a["n"], 1),
                                         counters={cn: v / max(a["n"], 1) for cn, v in a["c"].items()}) for kn, a in agg.items()}
        cal = {}
        for e in copies:
            for cn, v in e["c"].items():
                cal.setdefault(cn, []).append(v)
        res[g] = dict(per_graph=per_graph, calibration_copy_256MiB={cn: sum(v) / len(v) for cn, v in cal.items()},
                      calibration_copy_us=sum(e["us"] for e in copies) / max(len(copies), 1))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:6000])


if __name__ == "__main__":
    main()
