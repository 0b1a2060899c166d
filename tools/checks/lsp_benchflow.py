"""Bisect: bench.py's own set-up of the SAGE + LSP problem, then GraphedEpoch replays, with switches that swap single set-up
steps for the ones tools/checks/lsp_trace.py uses (where the replayed loss_aux is right)."""
import os, sys, types
os.environ.setdefault("EGNN_GRAPH_AUDIT", "0")   # reproducer of the r04 finding: captures steps with long torch reductions on purpose
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
import efficient_gnns_amd.data as D, efficient_gnns_amd.models as PM, efficient_gnns_amd.ops as ops
from efficient_gnns_amd.utils import subgraph
sw = set(os.environ.get("SW", "").split(","))
steps = int(os.environ.get("STEPS", 6))
args = types.SimpleNamespace(gnn="sage", training="lpw", seed=0, scale=1.0)
hp = dict(bench.HP); hp.update(bench.MODE_HP["lpw"])
device = torch.device("cuda", 0)
if "threads" not in sw:
    bench.cap_cpu_threads(1)
torch.cuda.set_device(0)
bench.seed_all(0)
data = D.arxiv_like(1.0, seed=0)
d = bench.to_device(data, device)
ei = torch.stack(d.adj_t.coo()[:2])
tr = d.split_idx["train"]
if "owntrain" in sw:
    tr = tr.clone()
edge_index = subgraph(tr, ei, relabel_nodes=True, num_nodes=d.num_nodes)[0]
bench.seed_all(0)
if "ownmodel" in sw:
    model = PM.SAGE(d.num_features, 256, d.num_classes, 3, 0.5).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, fused=True, capturable=True)
    sp = tp = None
else:
    model, sp, tp, opt = bench.build_problem(PM, d, device, args, hp)
split = d.split_idx if "ownsplit" not in sw else {k: v.clone() for k, v in d.split_idx.items()}
if "nosplit" in sw:
    split = None
import efficient_gnns_amd.ops_edge as OE
from efficient_gnns_amd import _lib
if "ownsum" in sw:
    def own(feat, teacher_feat, edge_index, kern, criterion="kld"):
        n = feat.shape[0]
        plan = OE.edge_plan(edge_index, n)
        p_s = OE._SegSoftmax.apply(OE._EdgeSim.apply(feat, plan, kern), plan.ptr_b)
        p_t = OE._SegSoftmax.apply(OE._EdgeSim.apply(teacher_feat, plan, kern), plan.ptr_b)
        el = torch.nn.functional.kl_div(torch.log(p_s), p_t, log_target=False, reduction="none")
        # sum without torch's multi-block reduction: [1, E] @ [E, 1] on the package GEMM is overkill; chunked 2-D sum keeps every
        # torch reduction single-block (<= 1024 outputs of short rows)
        pad = (-el.numel()) % 1024
        el2 = torch.cat([el, el.new_zeros(pad)]).view(1024, -1)
        return el2.sum(1).sum() / el.numel()
    OE.lsp_loss = own
if "dot" in sw:
    _G = torch.cuda.CUDAGraph
    class DG(_G):
        def __new__(cls, *a, **k):
            g = _G.__new__(cls, *a, **k)
            return g
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.enable_debug_mode()
    torch.cuda.CUDAGraph = DG
ge = PM.GraphedEpoch(model, d.x, d.adj_t, d.y, tr, opt, "lpw", hp, d.teacher_out_feat, d.teacher_logits, sp, tp, edge_index,
                     split_idx=split, warmup=3)
if "gc" in sw:
    import gc
    gc.collect()
    gc.freeze()
torch.cuda.synchronize()
if "dot" in sw:
    out = os.path.join(R, "gpurun_out/r04/call8/graph.dot")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    try:
        ge.graph.debug_dump(out)
        txt = open(out).read()
        import re, collections
        print("# dot bytes", len(txt), "edges", txt.count("->"), flush=True)
        kinds = collections.Counter(re.findall(r"(MEMSET|MEMCPY|KERNEL|EMPTY|Memset|Memcpy|memset|memcpy)", txt))
        print("# node kinds", dict(kinds), flush=True)
    except Exception as e:
        print("# debug_dump failed", type(e).__name__, str(e)[:200], flush=True)
print("# SW=", sorted(sw), flush=True)
for s in range(steps):
    l, a = ge.step()
    print(f"step {s} loss {l[0]:.5f} cls {l[1]:.5f} aux {l[2]:.4e}", flush=True)
