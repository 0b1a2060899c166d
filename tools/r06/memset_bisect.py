"""Smallest failing graph of the round-4 finding?  (ATen's multi-block reductions zero their semaphores with a memset NODE once
captured; inside the replayed ~120-node epoch graph such a reduction left its output unwritten in replays that found the device idle; a
3-kernel micro graph did not reproduce it.)  This sweep builds synthetic CHAINS: N small kernel nodes, then long reductions (each: memset node
+ kernels), then M small kernel nodes -- and looks for a replay whose reduction output was not written (the output is poisoned before every
replay, the input changes every replay, the device is idle when the replay is launched).
    python tools/r06/memset_bisect.py [--explicit]"""
import itertools, sys, time
import torch

dev = torch.device("cuda:0")
EXPLICIT = "--explicit" in sys.argv
E = 679910
torch.manual_seed(0)


def census(g):
    try:
        sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
        from efficient_gnns_amd._audit import graph_node_kinds
        return graph_node_kinds(g)
    except Exception as e:  # noqa: BLE001
        return f"census failed: {e}"


def trial(N, M, n_red, big_between, replays=8, idle_ms=5.0):
    x = torch.rand(E, device=dev) + 1.0
    a = torch.zeros(256, device=dev)
    b = torch.zeros(256, device=dev)
    w = torch.rand(4096, 256, device=dev) if big_between else None

    def body():
        for _ in range(N):
            a.add_(1.0)
        outs = []
        for i in range(n_red):
            if big_between:                      # a kernel with real work between the reductions (allocates + frees pool blocks)
                t = (w @ w.t()).relu_()
                a.add_(t[0, :256])
                del t
            ws = torch.empty(3072, device=dev)   # a block that is dirtied, freed and likely re-used by the reduction's scratch
            ws.fill_(368.0 + i)
            a.add_(ws[:256])
            del ws
            outs.append((x * (1.0 + 0.5 * i)).mean())
        for _ in range(M):
            b.add_(1.0)
        return torch.stack(outs)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if EXPLICIT:      # keep the graph, read it back, instantiate explicitly (the path models.GraphedEpoch has taken since round 5)
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            out = body()
        kinds = census(g)
        g.instantiate()
    else:             # torch's default: instantiation inside capture_end (the path of rounds 2-4, on which the finding was made)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = body()
        kinds = "-"
    torch.cuda.synchronize()
    bad = 0
    first_bad = None
    for r in range(replays):
        x.mul_(1.0 + 0.01 * (r + 1))
        want = [float(x.double().mean()) * (1.0 + 0.5 * i) for i in range(n_red)]
        out.fill_(-777.0)
        torch.cuda.synchronize()
        time.sleep(idle_ms * 1e-3)               # the replay finds an idle device
        g.replay()
        got = out.tolist()
        wrong = [i for i, (p, q) in enumerate(zip(got, want)) if abs(p - q) > 1e-4 * abs(q)]
        if wrong:
            bad += 1
            first_bad = first_bad or (r, wrong, [got[i] for i in wrong[:3]], [want[i] for i in wrong[:3]])
    return bad, first_bad, kinds


print("N kernels before | reductions | M kernels after | matmul between | bad replays / 8 | node census | first bad (replay, outputs, got, want)")
found = None
for N, n_red, M, big in itertools.product((0, 20, 100, 400), (1, 4, 16), (0, 100), (False, True)):
    bad, first_bad, kinds = trial(N, M, n_red, big)
    print(f"{N:5d} {n_red:3d} {M:5d} {str(big):5s}  bad={bad}  {kinds}  {first_bad}", flush=True)
    if bad and found is None:
        found = (N, n_red, M, big)
print("SMALLEST-FAILING", found)
