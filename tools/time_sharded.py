"""Per-phase wall time (with device syncs) of the sharded path with one rank over RCCL: finds host/allocator stalls."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
import numpy as np, torch, torch.distributed as dist
import bench
import efficient_gnns_amd.dist as DD, efficient_gnns_amd.data as D, efficient_gnns_amd.models as PM
from efficient_gnns_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if os.environ.get("NO_DIST_INIT_BUT_SAME_CODE") == "1":
    dist.init_process_group("gloo")
    # world of one: every collective is the identity / a copy
    dist.all_reduce = lambda t, *a, **k: None
    dist.all_gather_into_tensor = lambda o, i, *a, **k: o.view(-1).copy_(i.reshape(-1))
    dist.all_to_all_single = lambda o, i, *a, **k: None
else:
    dist.init_process_group("nccl", device_id=dev)
hp = dict(bench.HP); cfg = bench.MODEL
data = D.arxiv_like(1.0, seed=0)
prob = DD.ShardedProblem(data, 1, 0, dev, None, need_gcn=True)
model = PM.GCN(data.num_features, cfg["hidden"], data.num_classes, cfg["layers"], cfg["dropout"]).to(dev)
_swap = (lambda m: m) if os.environ.get("NO_SWAP_BN") == "1" else DD.swap_batchnorm
_swap(model)
sp = _swap(PM.make_projection(cfg["hidden"], hp["proj_dim"]).to(dev))
tp = _swap(PM.make_projection(750, hp["proj_dim"]).to(dev))
ADJ = data.adj_t.to(dev) if os.environ.get("USE_PLAIN_ADJ") == "1" else prob.adj
opt = torch.optim.Adam([{"params": m.parameters(), "lr": 0.01} for m in (model, sp, tp)])
import gc
GC = []
def _cb(phase, info, _t=[0.0]):
    if phase == "start": _t[0] = time.perf_counter()
    else: GC.append((info["generation"], time.perf_counter() - _t[0], info["collected"]))
gc.callbacks.append(_cb)
if os.environ.get("FREEZE") == "1":
    gc.collect(); gc.freeze()
import threading, traceback
SAMPLES = []
_main_id = threading.get_ident()
_stop = False
def _sampler():
    while not _stop:
        fr = sys._current_frames().get(_main_id)
        if fr is not None:
            st = traceback.extract_stack(fr)[-4:]
            SAMPLES.append((time.perf_counter(), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st))))
        time.sleep(0.004)
threading.Thread(target=_sampler, daemon=True).start()
SLOW = []
T = {}
COLL = {}
def _wrap(name):
    orig = getattr(dist, name)
    def f(*a, **k):
        if os.environ.get("COLL_SYNC") == "1":
            torch.cuda.synchronize(); t = time.perf_counter(); r = orig(*a, **k); torch.cuda.synchronize()
            COLL.setdefault((name, tuple(a[0].shape)), []).append(time.perf_counter() - t)
            return r
        return orig(*a, **k)
    setattr(dist, name, f)
for _n in ("all_reduce", "all_gather_into_tensor", "all_to_all_single"):
    _wrap(_n)
def tick(name, t0):
    torch.cuda.synchronize(); d = time.perf_counter() - t0; T.setdefault(name, []).append(d)
    if d > 0.02: SLOW.append((name, t0, t0 + d))
    return time.perf_counter()
for ep in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    model.train(); sp.train(); tp.train()
    out_full = model(prob.x, ADJ); t = tick("fwd model", t)
    _tk = ops.take_rows if os.environ.get('USE_TAKE_ROWS') == '1' else (lambda x, i: x[i])
    out = _tk(out_full, prob.train_local); labels = prob.y.squeeze(1)[prob.train_local]
    loss_cls = ops.cross_entropy(out, labels); t = tick("take+ce", t)
    f = sp(_tk(model.out_feat, prob.train_local)); tt = tp(_tk(prob.teacher_out_feat, prob.train_local)); t = tick("proj heads", t)
    S = hp["max_samples"]; ntr = prob.n_train_global
    pick = np.random.choice(ntr, S, replace=False); pick_t = torch.from_numpy(pick)
    owner = prob.train_owner[pick_t]; counts = torch.bincount(owner, minlength=1).tolist()
    idx = prob.train_localpos[pick_t[owner == 0]].to(dev); t = tick("sampling(host)", t)
    fhat = ops.gather_normalize(f, idx); that = ops.gather_normalize(tt, idx)
    loss_aux = ops.nce_unit(fhat, that, hp["nce_T"]) if os.environ.get('USE_NCE_UNIT') == '1' else DD._DistNCE.apply(fhat, that, hp["nce_T"], counts, 0, None); t = tick("nce fwd", t)
    loss = loss_cls + hp["beta"] * loss_aux
    opt.zero_grad(); loss.backward(); t = tick("backward", t)
    params = [p for g in opt.param_groups for p in g["params"]]
    DD.allreduce_grads(params, None); t = tick("grad allreduce", t)
    opt.step(); t = tick("adam", t)
    if os.environ.get("USE_PLAIN_ADJ") == "1":
        with torch.no_grad():
            model.eval(); o = model(prob.x, ADJ); o.argmax(-1)
    else:
        DD.sharded_evaluate(model, prob)
    t = tick("eval", t)
for k, v in T.items():
    v = v[3:]
    print(f"{k:18s} mean {1e3 * sum(v) / len(v):8.2f} ms   max {1e3 * max(v):8.2f}   min {1e3 * min(v):8.2f}")
import collections
by = collections.defaultdict(list)
for g, d, c in GC: by[g].append(d)
for g in sorted(by): print(f"gc gen{g}: {len(by[g])} collections, total {1e3 * sum(by[g]):.1f} ms, max {1e3 * max(by[g]):.1f} ms")
print("total mean", 1e3 * sum(sum(v[3:]) / len(v[3:]) for v in T.values()))


_stop = True
import collections
for name, a, b in SLOW[-6:]:
    c = collections.Counter(st for (t, st) in SAMPLES if a <= t <= b)
    print(f"SLOW {name} {1e3 * (b - a):.1f} ms:")
    for st, n in c.most_common(3): print(f"   {n:3d} x {st}")

for k, v in sorted(COLL.items(), key=lambda kv: -max(kv[1])):
    print(f"COLL {k}: n={len(v)} median {1e6 * sorted(v)[len(v) // 2]:.0f} us max {1e6 * max(v):.0f} us")
