"""Do the streaming BatchNorm kernels care where their tensors sit relative to each other?  (In the replayed graph bn_act_fwd /
bn_act_bwd_apply run 35-50 % slower than the same launches in eager epochs: profiles/r06_replay_vs_eager_kernels.txt.)
x, dy, dx / y are views of ONE buffer at controlled byte distances."""
import sys, os, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efficient_gnns_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
n, Cc = 169343, 256
nbytes = n * Cc * 4
big = torch.empty(4 * nbytes + (64 << 20), dtype=torch.uint8, device=dev)
base = big.data_ptr()
base_al = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
def view(off):
    o = base_al - base + off
    return big[o:o + nbytes].view(torch.float32).view(n, Cc)
mean = torch.zeros(Cc, device=dev); var = torch.ones(Cc, device=dev); g = torch.ones(Cc, device=dev); b = torch.zeros(Cc, device=dev)
nws = lib.egnn_bn_ws_floats(Cc); ws = torch.empty(nws, device=dev)
dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev); cs = torch.empty(Cc, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
span = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)        # next 2 MiB boundary after one tensor
print("tensor bytes", nbytes, "span", span, "base mod 2MiB", base % (2 << 20))
for d in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, nbytes - span + 0, nbytes - span + 4096):
    x = view(0); y = view(span + d); dy = view(2 * span + 2 * d); dx = view(3 * span + 3 * d)
    x.normal_(); dy.normal_()
    fwd = t(lambda: _lib.check(lib.egnn_bn_act_fwd_f32(_lib.ptr(x), Cc, n, Cc, _lib.ptr(mean), _lib.ptr(var), 1e-5, _lib.ptr(g), _lib.ptr(b), 1, 0.5, 123, None,
                                                        _lib.ptr(y), Cc, _lib.stream()), "fwd"))
    bwd = t(lambda: _lib.check(lib.egnn_bn_act_bwd_colsum_f32(_lib.ptr(x), Cc, _lib.ptr(dy), Cc, n, Cc, _lib.ptr(mean), _lib.ptr(var), 1e-5, _lib.ptr(g), _lib.ptr(b),
                                                               1, 0.5, 123, None, 1, _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dx), Cc, _lib.ptr(cs), _lib.ptr(ws), nws,
                                                               _lib.stream()), "bwd"))
    cp = t(lambda: y.copy_(x))
    print(f"distance between tensors = span + {d:8d} B : bn_act_fwd {fwd:7.1f} us   bwd(reduce+apply) {bwd:7.1f} us   copy {cp:7.1f} us", flush=True)
