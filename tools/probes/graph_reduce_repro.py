"""Micro reproducer: torch multi-block reductions (Reduce.cuh: semaphores zeroed by a captured cudaMemsetAsync = a memset
NODE) inside a hipGraph whose private pool hands the reduction an output / semaphore block that an earlier kernel of the same
graph dirtied.  Prints, per variant, how many replays returned a wrong value."""
import os, sys
import torch
dev = torch.device("cuda:0")
E = 679910
SYNC = os.environ.get("SYNC", "1") == "1"


def run(name, body, want, replays=6):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    if SYNC:
        torch.cuda.synchronize()
    bad = 0
    vals = []
    for _ in range(replays):
        g.replay()
        v = out.tolist()
        vals.append(v)
        bad += int(any(abs(a - b) > 1e-3 * max(1.0, abs(b)) for a, b in zip(v, want)))
    print(f"{name:40s} sync_before_first_replay={SYNC} bad_replays={bad}/{replays} first={vals[0]} want={want}", flush=True)


x = torch.rand(E, device=dev) + 1.0
m, mn, mx = float(x.mean()), float(x.min()), float(x.max())


def dirty_then_reduce():
    ws = torch.empty(3072, device=dev)
    ws.fill_(368.0)
    tmp = ws.sum()              # consume
    del ws
    a = x.mean()                # output / semaphores may land in the freed block
    return torch.stack([a, tmp / 3072])


def reduce_only():
    return torch.stack([x.mean(), x.min(), x.max()])


def dirty_many():
    outs = []
    for i in range(4):
        ws = torch.empty(3072, device=dev)
        ws.fill_(100.0 + i)
        del ws
        outs.append(x.mean())
        outs.append(x.min())
    return torch.stack(outs)


run("reduce_only", reduce_only, [m, mn, mx])
run("dirty_then_reduce", dirty_then_reduce, [m, 368.0])
run("dirty_many", dirty_many, [m, mn] * 4)
