"""GPU path vs CPU oracle over MANY optimisation steps of one configuration (default: SAGE + LSP(rbf), beta = 100), same
weights / seeds / dropout 0: prints the three losses of both per step.  A drift that the 3-step goldens cannot see shows here."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efficient_gnns_amd.data as D, efficient_gnns_amd.models as PM
import oracle.models as OM, oracle.sparse as OS, oracle.utils as OU
ap = argparse.ArgumentParser()
ap.add_argument("--gnn", default="sage"); ap.add_argument("--mode", default="lpw"); ap.add_argument("--steps", type=int, default=25)
ap.add_argument("--scale", type=float, default=0.05); ap.add_argument("--kernel", default="rbf"); ap.add_argument("--beta", type=float, default=100.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
d = D.arxiv_like(scale=a.scale, seed=1)
hp = dict(alpha=0.9, kd_T=4.0, beta=a.beta, nce_T=0.075, max_samples=1024, kernel=a.kernel)
torch.manual_seed(0)
Net = (OM.GCN, PM.GCN) if a.gnn == "gcn" else (OM.SAGE, PM.SAGE)
om = Net[0](d.num_features, 64, d.num_classes, 3, 0.0)
pm = Net[1](d.num_features, 64, d.num_classes, 3, 0.0).to(dev)
pm.load_state_dict(om.state_dict())
osp = otp = psp = ptp = None
og, pg = [{"params": om.parameters(), "lr": 0.01}], [{"params": pm.parameters(), "lr": 0.01}]
if a.mode in ("nce", "gpw"):
    osp, otp = OM.make_projection(64, 32), OM.make_projection(750, 32)
    psp, ptp = PM.make_projection(64, 32).to(dev), PM.make_projection(750, 32).to(dev)
    psp.load_state_dict(osp.state_dict()); ptp.load_state_dict(otp.state_dict())
    og += [{"params": osp.parameters(), "lr": 0.01}, {"params": otp.parameters(), "lr": 0.01}]
    pg += [{"params": psp.parameters(), "lr": 0.01}, {"params": ptp.parameters(), "lr": 0.01}]
oo, po = torch.optim.Adam(og), torch.optim.Adam(pg)
rowptr, col, _ = d.adj_t.csr()
oadj = OS.SparseTensor(rowptr=rowptr, col=col, sparse_sizes=d.adj_t.sparse_sizes())
tr = d.split_idx["train"]
eo = ep = None
if a.mode == "lpw":
    row = torch.repeat_interleave(torch.arange(d.num_nodes), rowptr[1:] - rowptr[:-1])
    eo = OU.subgraph(tr, torch.stack([row, col]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
    ep = eo.to(dev)
X, A, Y, T, TL = d.x.to(dev), d.adj_t.to(dev), d.y.to(dev), d.teacher_out_feat.to(dev), d.teacher_logits.to(dev)
worst = 0.0
for s in range(a.steps):
    np.random.seed(100 + s); ro = OM.train_step(om, d.x, oadj, d.y, tr, oo, a.mode, hp, d.teacher_out_feat, d.teacher_logits, osp, otp, eo)
    np.random.seed(100 + s); rp = PM.train_step(pm, X, A, Y, tr.to(dev), po, a.mode, hp, T, TL, psp, ptp, ep)
    rel = max(abs(x - y) / max(abs(y), 1e-6 * max(abs(ro[0]), 1.0)) for x, y in zip(rp, ro))
    worst = max(worst, rel)
    print(f"step {s:2d}  gpu {tuple(round(v, 5) for v in rp)}  cpu {tuple(round(v, 5) for v in ro)}  rel {rel:.2e}")
print("WORST", worst)
