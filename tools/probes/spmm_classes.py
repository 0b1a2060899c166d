"""Time the three row classes of the SpMM schedule separately (tools only): is a K=128 / K=40 call bound by the hubs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import efficient_gnns_amd as E, efficient_gnns_amd.data as D
from efficient_gnns_amd import _lib
from efficient_gnns_amd.sparse import gcn_norm
dev = "cuda"
d = D.arxiv_like(1.0, seed=0, with_teacher=False)
adj = gcn_norm(d.adj_t.to(dev))
rowptr, col, bits = adj._index_arrays()
short, mid, long_ = adj._row_plan()
lib = _lib.load()
n = adj.sparse_size(0)
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for K in (256, 128, 40):
    x = torch.randn(n, K, device=dev); y = torch.empty(n, K, device=dev)
    def call(s, m, l):
        def p(v): return (None, 0) if v is None else (_lib.ptr(v), v.numel())
        (ps, ns), (pm, nm), (pl, nl) = p(s), p(m), p(l)
        rc = lib.egnn_spmm_csr_f32(n, n, K, _lib.ptr(rowptr), _lib.ptr(col), bits, _lib.ptr(adj._value), None, None, _lib.ptr(x), K,
                                   _lib.ptr(y), K, 0, None, ps, ns, pm, nm, pl, nl, _lib.stream())
        assert rc == 0
    print(f"K={K:3d}  short {t(lambda: call(short, None, None)):7.1f} us   mid {t(lambda: call(None, mid, None)):7.1f} us   "
          f"long {t(lambda: call(None, None, long_)):7.1f} us   all-in-one-stream {t(lambda: call(short, mid, long_)):7.1f} us   "
          f"overlapped (ops.spmm_raw) {t(lambda: E.ops.spmm_raw(adj, x)):7.1f} us")
