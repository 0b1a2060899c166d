"""How long does the HOST need to enqueue one training step (no device sync until the losses are read)?  If that is close
to the step time the run is launch-bound -- what an 8-rank run becomes when the per-rank GPU work shrinks 8x.  (tools only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import efficient_gnns_amd.data as D, efficient_gnns_amd.models as PM, efficient_gnns_amd.criterion as C
from efficient_gnns_amd import ops
bench.cap_cpu_threads()
dev = torch.device("cuda", 0)
hp = dict(bench.HP); cfg = bench.MODEL
data = D.arxiv_like(1.0, seed=0)
d = bench.to_device(data, dev)
args = type("A", (), dict(gnn="gcn", training="nce", seed=0))()
model, sp, tp, opt = bench.build_problem(PM, d, dev, args, hp)
tr = d.split_idx["train"]
enq, tot = [], []
for ep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.train(); sp.train(); tp.train()
    out = ops.take_rows(model(d.x, d.adj_t), tr)
    labels = d.y.squeeze(1)[tr]
    loss, lc, la = PM.distill_loss("nce", model, out, labels, tr, d.teacher_out_feat, d.teacher_logits, hp, sp, tp, None, d.adj_t, False)
    opt.zero_grad(); loss.backward(); opt.step()
    t1 = time.perf_counter()
    vals = (loss.item(), lc.item(), la.item())
    t2 = time.perf_counter()
    PM.evaluate(model, d.x, d.adj_t, d.y, d.split_idx)
    t3 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(f"train step: host enqueue {1e3 * np.median(enq[3:]):.2f} ms, until losses read {1e3 * np.median(tot[3:]):.2f} ms")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for ep in range(5):
    model.train()
    out = ops.take_rows(model(d.x, d.adj_t), tr)
    labels = d.y.squeeze(1)[tr]
    loss, lc, la = PM.distill_loss("nce", model, out, labels, tr, d.teacher_out_feat, d.teacher_logits, hp, sp, tp, None, d.adj_t, False)
    opt.zero_grad(); loss.backward(); opt.step()
    loss.item()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
