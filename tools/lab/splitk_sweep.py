"""Split-K sweep of the transposed weight-gradient GEMMs (dW = X^T dY, arxiv_pyg/gnn.py:192 via autograd) at the headline sizes:
ops.gemm_raw(trans_a=True) for the four shapes of one epoch, split_k from 32 to 256.    python tools/lab/splitk_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import efficient_gnns_amd.ops as ops  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
N, NTR = 169343, 90941
idx = torch.randperm(N, generator=g)[:NTR].to(dev)
cases = [("layer dW 256x256xN", torch.randn(N, 256, generator=g), torch.randn(N, 256, generator=g), None),
         ("layer-1 dW 128x256xN", torch.randn(N, 128, generator=g), torch.randn(N, 256, generator=g), None),
         ("student head dW 256x256xNtr (rows)", torch.randn(NTR, 256, generator=g), torch.randn(N, 256, generator=g), idx),
         ("teacher head dW 256x750xNtr (rows)", torch.randn(NTR, 256, generator=g), ops.pad_pitch(torch.randn(N, 750, generator=g).to(dev)), idx)]
for name, a, b, rows in cases:
    a, b = a.to(dev), b.to(dev)
    M, Nn = a.shape[1], b.shape[1]
    K = a.shape[0]
    flops = 2.0 * M * Nn * K
    ref = None
    out = []
    for sk in (None, 32, 48, 64, 85, 96, 128, 170, 256):
        def run():
            return ops.gemm_raw(a, b, True, False, split_k=sk, b_rows=rows)
        for _ in range(3):
            c = run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            c = run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        if ref is None:
            ref = c.double()
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        out.append(f"sk={sk}: {us:7.1f} us {flops / us / 1e6:6.1f} TF (dev {err:.1e})")
    print(f"{name}: " + " | ".join(out), flush=True)
