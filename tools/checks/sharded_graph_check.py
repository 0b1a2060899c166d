#!/usr/bin/env python3
"""dist.ShardedGraphedEpoch against the eager sharded steps on one rank over RCCL (same host draws, dropout 0): three replays
reproduce three eager steps.  Exit code 0 + the line SHARDED-GRAPH-OK on success (tests/test_gpu_parity.py runs it in its own
interpreter).

    python tools/checks/sharded_graph_check.py --gnn gcn --mode nce
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import efficient_gnns_amd.data as D  # noqa: E402
import efficient_gnns_amd.dist as DD  # noqa: E402
import efficient_gnns_amd.models as PM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gnn", default="gcn")
ap.add_argument("--mode", default="nce")
ap.add_argument("--static", action="store_true", help="the fixed-capacity sample layout of the multi-rank runs (dist.StaticSample) on this one rank")
ap.add_argument("--port", type=int, default=0, help="rendezvous port (0 = any free port: back-to-back runs must not meet a lingering socket)")
a = ap.parse_args()
gnn, mode, DEV = a.gnn, a.mode, "cuda"
if a.port == 0:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        a.port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{a.port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d = D.arxiv_like(scale=0.02, seed=5)
hp = dict(alpha=0.9, kd_T=4.0, beta=0.1 if mode == "nce" else 100.0, nce_T=0.075, max_samples=512, kernel="cosine")
prob = DD.ShardedProblem(d, 1, 0, torch.device(DEV, 0), None, need_gcn=(gnn == "gcn"))


def build():
    torch.manual_seed(0)
    m = DD.swap_batchnorm((PM.GCN if gnn == "gcn" else PM.SAGE)(d.num_features, 64, d.num_classes, 3, 0.0).to(DEV))
    sp = tp = None
    params = list(m.parameters())
    if mode in ("nce", "gpw"):
        sp = DD.swap_batchnorm(PM.make_projection(64, 32).to(DEV))
        tp = DD.swap_batchnorm(PM.make_projection(750, 32).to(DEV))
        params += list(sp.parameters()) + list(tp.parameters())
    return m, sp, tp, torch.optim.Adam(params, lr=0.01, fused=True, capturable=True)


m1, sp1, tp1, o1 = build()
m2, sp2, tp2, o2 = build()
np.random.seed(3)
eager = []
for _ in range(3):
    l = DD.sharded_train_step(m1, prob, o1, mode, hp, sp1, tp1)
    _, acc = DD.sharded_evaluate(m1, prob)
    eager.append((l, acc))
# the graph's constructor runs `warmup` untimed steps on the model: restore the initial state afterwards
state = [copy.deepcopy(x.state_dict()) if x is not None else None for x in (m2, sp2, tp2)]
ge = DD.ShardedGraphedEpoch(m2, prob, o2, mode, hp, sp2, tp2, warmup=2, static_sample=True if a.static else None)
for x, st in zip((m2, sp2, tp2), state):
    if x is not None:
        x.load_state_dict(st)
for grp in o2.param_groups:          # Adam state back to step 0
    for p_ in grp["params"]:
        stt = o2.state[p_]
        if stt:
            stt["exp_avg"].zero_()
            stt["exp_avg_sq"].zero_()
            stt["step"].zero_()
np.random.seed(3)
ge._refresh()
got = [ge.step() for _ in range(3)]
for (le, ae), (lg, ag) in zip(eager, got):
    np.testing.assert_allclose(np.array(lg), np.array(le), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(np.array(ag), np.array(ae), atol=2e-3)
print("eager", eager[-1], "replayed", got[-1])
# round 6: the launch-ahead loop (step_async) on an identically prepared twin: same replays, same draws -> the SAME values as `got`, one call later
m3, sp3, tp3, o3 = build()
state3 = [copy.deepcopy(x.state_dict()) if x is not None else None for x in (m3, sp3, tp3)]
ge2 = DD.ShardedGraphedEpoch(m3, prob, o3, mode, hp, sp3, tp3, warmup=2, static_sample=True if a.static else None)
for x, st in zip((m3, sp3, tp3), state3):
    if x is not None:
        x.load_state_dict(st)
for grp in o3.param_groups:
    for p_ in grp["params"]:
        stt = o3.state[p_]
        if stt:
            stt["exp_avg"].zero_()
            stt["exp_avg_sq"].zero_()
            stt["step"].zero_()
np.random.seed(3)
ge2._refresh()
assert ge2.step_async() is None
later = [ge2.step_async(), ge2.step_async()]
try:
    ge2.step()
    raise SystemExit("step() must refuse while an epoch's values are in flight")
except RuntimeError:
    pass
later.append(ge2.drain())
assert ge2.drain() is None
for (lg, ag), (la, aa) in zip(got, later):
    np.testing.assert_allclose(np.array(la), np.array(lg), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(np.array(aa), np.array(ag), atol=1e-9)
print("async", later[-1])
print("SHARDED-GRAPH-OK", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.stdout.flush()
os._exit(0)     # leave without the interpreter teardown (communicator threads, see bench.py)
