#!/usr/bin/env python3
"""Full-size (N = 169 343) forward + backward of the single operators of the train step against float64 torch ON THE GPU:
localises a gradient discrepancy that only shows at full size.  Prints max |err| / max |ref| per quantity."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import efficient_gnns_amd as E  # noqa: E402
import efficient_gnns_amd.data as D  # noqa: E402
import efficient_gnns_amd.ops as ops  # noqa: E402

DEV = "cuda"


def rel(a, b):
    b = b.double()
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))


def report(name, **kw):
    print(f"{name:44s} " + "  ".join(f"{k} {v:.2e}" for k, v in kw.items()), flush=True)


torch.manual_seed(0)
N, C = 169343, 256
x = (torch.randn(N, C, device=DEV) * 1.7 + torch.randn(C, device=DEV))
gy = torch.randn(N, C, device=DEV) * (torch.rand(N, 1, device=DEV) < 0.54)      # gradient only on "train" rows, like the CE gradient
# ---- fused BatchNorm + ReLU (+ dropout 0) ----------------------------------------------------------------------------------
for relu in (True, False):
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    xp = x.clone().requires_grad_(True)
    y = ops.bn_act(xp, bn, relu=relu, p=0.0, training=True)
    y.backward(gy)
    xd = x.double().requires_grad_(True)
    w, b = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    yr = F.batch_norm(xd, None, None, w, b, True, 0.0, bn.eps)
    yr = torch.relu(yr) if relu else yr
    yr.backward(gy.double())
    report(f"bn_act relu={relu} n={N}", fwd=rel(y, yr), dx=rel(xp.grad, xd.grad), dgamma=rel(bn.weight.grad, w.grad), dbeta=rel(bn.bias.grad, b.grad))
# ---- dense layers -----------------------------------------------------------------------------------------------------------
for (n, k, m, name) in ((N, 256, 256, "linear 169343x256 -> 256"), (N, 128, 256, "linear 169343x128 -> 256"), (N, 256, 40, "linear 169343x256 -> 40"),
                        (90941, 750, 256, "linear 90941x750 -> 256")):
    xx = torch.randn(n, k, device=DEV)
    ww = torch.randn(m, k, device=DEV) * 0.1
    bb = torch.randn(m, device=DEV)
    g = torch.randn(n, m, device=DEV)
    xp, wp, bp = xx.clone().requires_grad_(True), ww.clone().requires_grad_(True), bb.clone().requires_grad_(True)
    yp = ops.linear(xp, wp, bp)
    yp.backward(g)
    xd, wd, bd = xx.double().requires_grad_(True), ww.double().requires_grad_(True), bb.double().requires_grad_(True)
    yd = F.linear(xd, wd, bd)
    yd.backward(g.double())
    report(name, fwd=rel(yp, yd), dx=rel(xp.grad, xd.grad), dw=rel(wp.grad, wd.grad), db=rel(bp.grad, bd.grad))
    # matmul form (GCNConv weight [in, out])
    w2 = ww.t().contiguous()
    xp, wp = xx.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    yp = ops.matmul(xp, wp)
    yp.backward(g)
    report(name + " (matmul)", fwd=rel(yp, yd - bd), dx=rel(xp.grad, xd.grad), dw=rel(wp.grad, wd.grad.t()))
# ---- fused gather + linear (projection heads) ---------------------------------------------------------------------------------
idx = torch.randperm(N, device=DEV)[:90941]
for k in (256, 750):
    feat = ops.pad_pitch(torch.randn(N, k, device=DEV))
    ww = torch.randn(256, k, device=DEV) * 0.1
    bb = torch.randn(256, device=DEV)
    g = torch.randn(90941, 256, device=DEV)
    fp, wp, bp = feat.clone().requires_grad_(True), ww.clone().requires_grad_(True), bb.clone().requires_grad_(True)
    yp = ops.linear_rows(fp, idx, wp, bp)
    yp.backward(g)
    fd, wd, bd = feat.double().requires_grad_(True), ww.double().requires_grad_(True), bb.double().requires_grad_(True)
    yd = F.linear(fd[idx], wd, bd)
    yd.backward(g.double())
    report(f"linear_rows 90941 of 169343 x {k} -> 256", fwd=rel(yp, yd), dx=rel(fp.grad, fd.grad), dw=rel(wp.grad, wd.grad), db=rel(bp.grad, bd.grad))
# ---- CE + KD on the train rows ---------------------------------------------------------------------------------------------------
logits = torch.randn(N, 40, device=DEV)
teacher = torch.randn(N, 40, device=DEV) * 3
labels = torch.randint(0, 40, (N,), device=DEV)
lp = logits.clone().requires_grad_(True)
out = ops.take_rows(lp, idx)
lc, lk = ops.ce_and_kd(out, labels[idx], teacher[idx], 4.0)
(lk * (0.9 * 16) + lc * 0.1).backward()
ld = logits.double().requires_grad_(True)
od = ld[idx]
lcd = F.cross_entropy(od, labels[idx])
lkd = F.kl_div(F.log_softmax(od / 4.0, dim=1), F.softmax(teacher.double()[idx] / 4.0, dim=1), log_target=False)
(lkd * (0.9 * 16) + lcd * 0.1).backward()
report("take_rows + ce_and_kd", ce=abs(float(lc) - float(lcd)) / float(lcd), kd=abs(float(lk) - float(lkd)) / float(lkd), dlogits=rel(lp.grad, ld.grad))
# ---- aggregation with the BatchNorm statistics epilogue, K = 256 ------------------------------------------------------------------
d = D.arxiv_like(1.0, seed=0, with_teacher=False)
adj = E.gcn_norm(d.adj_t.to(DEV))
h = torch.randn(N, 256, device=DEV) + torch.randn(256, device=DEV) * 0.5
bn = torch.nn.BatchNorm1d(256).to(DEV)
y = ops.spmm(adj, h, "sum", want_bn_stats=True, bn_stats_shift=bn.running_mean)
st = getattr(y, "_egnn_bn_stats", None)
yd = y.double()
report("spmm epilogue statistics", mean=rel(st[0], yd.mean(0)), var=rel(st[1], yd.var(0, unbiased=False)))
# whole GCN layer + BN backward
conv = E.GCNConv(256, 256, cached=True).to(DEV)
hp = h.clone().requires_grad_(True)
z = conv(hp, d.adj_t.to(DEV), bn_stats_shift=bn.running_mean, want_bn_stats=True)
out = ops.bn_act(z, bn, relu=True, p=0.0, training=True)
out.backward(gy)
rowptr, col, val = adj.csr()
A = torch.sparse_csr_tensor(rowptr, col, val.double(), size=(N, N))
hd = h.double().requires_grad_(True)
Wd, bd_ = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
zd = torch.sparse.mm(A, hd @ Wd) + bd_
gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
od = torch.relu(F.batch_norm(zd, None, None, gam, bet, True, 0.0, bn.eps))
od.backward(gy.double())
report("GCNConv(256,256) + bn_act, fwd+bwd", fwd=rel(out, od), dx=rel(hp.grad, hd.grad), dW=rel(conv.weight.grad, Wd.grad), dgamma=rel(bn.weight.grad, gam.grad),
       dbeta=rel(bn.bias.grad, bet.grad))
