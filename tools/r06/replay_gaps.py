"""Busy time vs span inside the replayed epoch, from a rocprofv3 --kernel-trace CSV: which share of a replay is the GPU idle
between two kernel nodes of the (chain-shaped) graph, and in front of which kernels.
usage: replay_gaps.py <kernel_trace.csv> [out.txt]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
n = len(names)
best = None
for p in range(60, 260):          # kernels per replay
    run, i, best_run = 0, 0, (0, 0)
    while i + p < n:
        if names[i] == names[i + p]:
            run += 1
        else:
            if run > best_run[0]:
                best_run = (run, i - run)
            run = 0
        i += 1
    if run > best_run[0]:
        best_run = (run, i - run)
    if best is None or best_run[0] > best[1][0] + p:      # longer periodic stretch (prefer the smallest period that explains it)
        best = (p, best_run)
p, (run, start) = best
reps = run // p
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(f"kernels per replay {p}; periodic stretch of {run} kernels = {reps} replays starting at record {start}", file=out)
seg = rows[start + p: start + p * (reps - 1)]      # drop the first and last replay of the stretch
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
k = len(seg) // p
print(f"replays analysed {k}: span {span / k / 1e3:.1f} us per replay, kernel busy {busy / k / 1e3:.1f} us, idle between kernels {(span - busy) / k / 1e3:.1f} us "
      f"({100.0 * (span - busy) / span:.2f} %)", file=out)
gap = collections.defaultdict(list)
dur = collections.defaultdict(list)
for j in range(1, len(seg)):
    pos = (j % p)
    gap[pos].append(seg[j][0] - seg[j - 1][1])
    dur[pos].append(seg[j][1] - seg[j][0])
print("pos  avg_dur_us  avg_gap_before_us  kernel", file=out)
for pos in range(p):
    if gap[pos]:
        nm = seg[pos][2] if pos < len(seg) else "?"
        print(f"{pos:3d}  {sum(dur[pos]) / len(dur[pos]) / 1e3:9.1f}  {sum(gap[pos]) / len(gap[pos]) / 1e3:8.2f}  {nm[:110]}", file=out)
