"""Can an MFMA-bound GEMM hide under the gather-bound aggregation?  The K = 256 aggregation (no LDS, 64 VGPRs, request-path-bound) and the
teacher head's forward GEMM (90 941 x 256 x 750 on planes: matrix-pipe-bound) launched back to back on one stream vs on two streams."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import efficient_gnns_amd as E
import efficient_gnns_amd.data as D
import efficient_gnns_amd.ops as ops
dev = torch.device("cuda:0")
d = D.arxiv_like(1.0, seed=0)
adj = E.gcn_norm(d.adj_t.to(dev))
x = torch.randn(d.num_nodes, 256, device=dev)
tf = ops.pad_pitch(d.teacher_out_feat.to(dev))
idx = d.split_idx["train"].to(dev)
w = torch.randn(256, 750, device=dev) * 0.05
b = torch.zeros(256, device=dev)
wl = torch.randn(256, 256, device=dev) * 0.05
h = torch.randn(d.num_nodes, 256, device=dev)
def spmm(): return ops.spmm_raw(adj, x, "sum")[0]
def gemm_head():
    with torch.no_grad():
        return ops.linear_rows(tf, idx, w, b, const_input=True)
def gemm_layer(): return ops.gemm_raw(h, wl, False, False)
def bn_like(): return torch.relu(h)     # a streaming pass (read + write 173 MB)
side = torch.cuda.Stream()
def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6
def seq(a, b_):
    def f(): a(); b_()
    return f
def par(a, b_):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            b_()
        a()
        torch.cuda.current_stream().wait_stream(side)
    return f
for name, a, b_ in (("spmm K=256 | teacher head gemm", spmm, gemm_head), ("spmm K=256 | layer gemm", spmm, gemm_layer),
                    ("relu pass | teacher head gemm", bn_like, gemm_head), ("spmm K=256 | relu pass", spmm, bn_like)):
    ta, tb = timed(a), timed(b_)
    ts, tp = timed(seq(a, b_)), timed(par(a, b_))
    print(f"{name:36s} alone {ta:7.1f} + {tb:7.1f} us   back to back {ts:7.1f}   two streams {tp:7.1f} us", flush=True)
