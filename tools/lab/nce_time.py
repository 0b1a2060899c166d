"""Times the G-CRD entry points (S = 16384, P = 256) of whatever libegnn_hip.so EGNN_LIB points at (lab variants)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efficient_gnns_amd._lib as L
if os.environ.get("EGNN_LIB"):
    L.LIB_PATH = os.environ["EGNN_LIB"]
import efficient_gnns_amd.ops as ops
S, P = 16384, 256
g = torch.Generator().manual_seed(0)
f = torch.nn.functional.normalize(torch.randn(S, P, generator=g)).cuda().requires_grad_()
t = torch.nn.functional.normalize(torch.randn(S, P, generator=g)).cuda().requires_grad_()
def step():
    f.grad = t.grad = None
    ops.nce_unit(f, t, 0.075).backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print("nce fwd+bwd us", round(e0.elapsed_time(e1) * 100, 1), os.environ.get("EGNN_LIB", "product"))
