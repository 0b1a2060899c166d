"""Times the last-hidden-layer pair (BatchNorm + ReLU + dropout -> narrow Linear, with the projection head's rows tapped in) at the
headline size: the composed ops vs ops.bn_act_linear (forward: EGNN_TAIL_ONE_PASS, backward kernel variants: EGNN_DXBN).
Run:  python tools/lab/tail_time.py [N] [C] [Ks]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import efficient_gnns_amd.ops as ops  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 169343
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    Ks = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    x0 = (torch.randn(N, C, generator=g) * 2).to(dev)
    w = (torch.randn(C, Ks, generator=g) * 0.1).to(dev).requires_grad_(True)
    wp = (torch.randn(C, C, generator=g) * 0.05).to(dev).requires_grad_(True)
    idx = torch.randperm(N, generator=g)[: int(N * 0.537)].to(dev)
    g_xw = torch.randn(N, Ks, generator=g).to(dev)
    g_rows = torch.randn(idx.numel(), C, generator=g).to(dev)
    bn = torch.nn.BatchNorm1d(C).to(dev)
    bn.train()

    def run(fused, reps=20):
        tf = tb = 0.0
        for it in range(reps + 3):
            x = x0.clone().requires_grad_(True)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            if fused:
                h, xw = ops.bn_act_linear(x, bn, w, relu=True, p=0.5, training=True)
            else:
                h = ops.grad_tap(ops.bn_act(x, bn, relu=True, p=0.5, training=True))
                xw = ops.matmul(h, w)
            rows = ops.linear_rows(h, idx, wp)
            e[1].record()
            torch.autograd.backward([xw, rows], [g_xw, g_rows])
            e[2].record()
            torch.cuda.synchronize()
            if it >= 3:
                tf += e[0].elapsed_time(e[1])
                tb += e[1].elapsed_time(e[2])
        return tf / reps * 1e3, tb / reps * 1e3

    for name, fused in (("composed", False), ("bn_act_linear", True)):
        f, b = run(fused)
        print(f"{name:14s} N={N} C={C} Ks={Ks}  fwd {f:8.1f} us  bwd {b:8.1f} us  (both include the head's row GEMMs)  "
              f"EGNN_DXBN={os.environ.get('EGNN_DXBN', 'default')} EGNN_TAIL_ONE_PASS={os.environ.get('EGNN_TAIL_ONE_PASS', '1')}", flush=True)


if __name__ == "__main__":
    main()
