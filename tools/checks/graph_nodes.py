"""What is inside the captured epoch graph?  Node types and edge structure of models.GraphedEpoch's hipGraph, read through the HIP
runtime (hipGraphGetNodes / hipGraphNodeGetType / hipGraphGetEdges), for
  VARIANT=product  the shipped LSP step (no torch reduction inside), and
  VARIANT=torchkl  the pre-round-4 criterion tail (segment softmaxes + torch.log + F.kl_div(..., 'mean'): ATen's multi-block reduction,
                   whose semaphores are zeroed by cudaMemsetAsync -> memset nodes).
Then replays each on an idle device and prints loss_aux.   MODEL=sage|gcn  TRAINING=lpw|nce
"""
import collections
import ctypes
import os
import sys

os.environ.setdefault("EGNN_GRAPH_AUDIT", "0")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efficient_gnns_amd.data as D
import efficient_gnns_amd.models as PM
import efficient_gnns_amd._audit as _A
# this tool captures the REFUSED kind of graph on purpose (the reproducer of the round-4 finding): the structural guard of round 5
# (models.GraphedEpoch reads every captured graph back and refuses memset nodes) is reduced to its census here
_A.check_captured_graph = lambda graph, what, kernels_only: _A.graph_node_kinds(graph)
import efficient_gnns_amd.ops as ops
import efficient_gnns_amd.ops_edge as OE
from efficient_gnns_amd.utils import subgraph

variant = os.environ.get("VARIANT", "product")
training = os.environ.get("TRAINING", "lpw")
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}

_G = torch.cuda.CUDAGraph


class KeptGraph(_G):          # keep the hipGraph_t after capture_end so that it can be inspected
    def __new__(cls, *a, **k):
        return _G.__new__(cls, keep_graph=True)

    def __init__(self, *a, **k):  # (the pybind constructor runs in __init__ with the arguments of the call)
        super().__init__(True)


torch.cuda.CUDAGraph = KeptGraph
if variant == "torchkl":
    def torch_tail(feat, teacher_feat, edge_index, kern, criterion="kld"):
        plan = OE.edge_plan(edge_index, feat.shape[0])
        p_s = OE._SegSoftmax.apply(OE._EdgeSim.apply(feat, plan, kern), plan.ptr_b)
        p_t = OE._SegSoftmax.apply(OE._EdgeSim.apply(teacher_feat, plan, kern), plan.ptr_b)
        return torch.nn.functional.kl_div(torch.log(p_s), p_t, log_target=False, reduction="mean")
    OE.lsp_loss = torch_tail

d = D.arxiv_like(scale=1.0, seed=0)
hp = dict(alpha=0.9, kd_T=4.0, beta=100.0 if training == "lpw" else 0.1, nce_T=0.075, max_samples=16384, kernel="rbf", proj_dim=256)
torch.manual_seed(0)
np.random.seed(0)
Net = PM.SAGE if os.environ.get("MODEL", "sage") == "sage" else PM.GCN
m = Net(d.num_features, 256, d.num_classes, 3, 0.5).to(dev)
A, tr = d.adj_t.to(dev), d.split_idx["train"].to(dev)
split = {k: v.to(dev) for k, v in d.split_idx.items()}
X, Y, T, TL = d.x.to(dev), d.y.to(dev), ops.pad_pitch(d.teacher_out_feat.to(dev)), d.teacher_logits.to(dev)
ei = sp = tp = None
params = list(m.parameters())
if training == "lpw":
    ei = subgraph(tr, torch.stack(A.coo()[:2]), relabel_nodes=True, num_nodes=d.num_nodes)[0]
else:
    sp, tp = PM.make_projection(256, 256).to(dev), PM.make_projection(750, 256).to(dev)
    params += list(sp.parameters()) + list(tp.parameters())
opt = torch.optim.Adam(params, lr=0.01, fused=True, capturable=True)
ge = PM.GraphedEpoch(m, X, A, Y, tr, opt, training, hp, T, TL, sp, tp, edge_index=ei, split_idx=split, warmup=3)
graph = ctypes.c_void_p(ge.graph.raw_cuda_graph())
n = ctypes.c_size_t(0)
assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) == 0
nodes = (ctypes.c_void_p * n.value)()
assert hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) == 0
kinds = collections.Counter()
type_of = {}
for nd in nodes:
    t = ctypes.c_int(-1)
    assert hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t)) == 0
    kinds[TYPES.get(t.value, str(t.value))] += 1
    type_of[nd] = TYPES.get(t.value, str(t.value))
ne = ctypes.c_size_t(0)
assert hip.hipGraphGetEdges(graph, None, None, ctypes.byref(ne)) == 0
src, dst = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
assert hip.hipGraphGetEdges(graph, src, dst, ctypes.byref(ne)) == 0
indeg, outdeg = collections.Counter(dst), collections.Counter(src)
roots = [x for x in nodes if indeg[x] == 0]
fan = sum(1 for x in nodes if indeg[x] > 1 or outdeg[x] > 1)
memset_edges = collections.Counter()
for a, b in zip(src, dst):
    if type_of.get(a) == "memset" or type_of.get(b) == "memset":
        memset_edges[(type_of.get(a), type_of.get(b))] += 1
print(f"# variant={variant} training={training}: {n.value} nodes {dict(kinds)}; {ne.value} edges; roots {len(roots)}; nodes with fan-in/out > 1: {fan}; "
      f"edges touching memset nodes {dict(memset_edges)}", flush=True)
ge.graph.instantiate()
torch.cuda.synchronize()          # first launch on an idle device
for s in range(4):
    l, a = ge.step()
    print(f"replay {s} loss {l[0]:.5f} cls {l[1]:.5f} aux {l[2]:.4e}", flush=True)
