"""Full-size SAGE + LSP(rbf, beta = 100) on the GPU path for N steps: prints the losses per step (is a blow-up of the
auxiliary term there, and does it depend on the pipeline switches given in the environment?)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efficient_gnns_amd.data as D, efficient_gnns_amd.models as PM, efficient_gnns_amd.ops as ops
from efficient_gnns_amd.utils import subgraph
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
d = D.arxiv_like(scale=1.0, seed=0)
hp = dict(alpha=0.9, kd_T=4.0, beta=100.0, nce_T=0.075, max_samples=16384, kernel="rbf")
torch.manual_seed(0); np.random.seed(0)
m = PM.SAGE(d.num_features, 256, d.num_classes, 3, 0.5).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=0.01)
A = d.adj_t.to(dev)
tr = d.split_idx["train"].to(dev)
ei = subgraph(tr, torch.stack(A.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
X, Y, T, TL = d.x.to(dev), d.y.to(dev), ops.pad_pitch(d.teacher_out_feat.to(dev)), d.teacher_logits.to(dev)
if os.environ.get("NO_TAP"):
    ops.grad_tap = lambda x: x
if os.environ.get("KERNEL"):
    hp["kernel"] = os.environ["KERNEL"]
if os.environ.get("GRAPH"):
    opt = torch.optim.Adam(m.parameters(), lr=0.01, fused=True, capturable=True)
    split = {k: v.to(dev) for k, v in d.split_idx.items()}
    ge = PM.GraphedEpoch(m, X, A, Y, tr, opt, "lpw", hp, T, TL, None, None, edge_index=ei, split_idx=split, warmup=3)
    for s in range(steps):
        l, a = ge.step()
        print(f"replay {s:2d} loss {l[0]:.5f} cls {l[1]:.5f} aux {l[2]:.3e} accs {tuple(round(v, 4) for v in a)}")
    sys.exit(0)
for s in range(steps):
    r = PM.train_step(m, X, A, Y, tr, opt, "lpw", hp, T, TL, None, None, ei)
    f = m.out_feat.detach()
    print(f"step {s:2d} loss {r[0]:.5f} cls {r[1]:.5f} aux {r[2]:.3e}  |out_feat| mean row norm {float(f.norm(dim=1).mean()):.3f} finite {bool(torch.isfinite(f).all())}")
