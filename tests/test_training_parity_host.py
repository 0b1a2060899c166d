"""CPU tests of the training-parity infrastructure (oracle/dropout.py, oracle/training_parity.py, efficient-gnns_amd/_audit.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.dropout import counter_mask, counter_uniform, injected_dropout
import oracle.training_parity as TP


def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _uniform_scalar(seed, idx):
    """csrc/bn_common.h uniform01, one element, plain Python integers."""
    h = _mix32((idx & 0xFFFFFFFF) ^ (seed & 0xFFFFFFFF))
    h = _mix32((h + (seed >> 32) + ((idx >> 32) * 0x9E3779B9)) & 0xFFFFFFFF)
    return np.float32(h >> 8) * np.float32(1.0 / 16777216.0)


@pytest.mark.parametrize("seed", [0, 1, 0x7FFFFFFFFFFFFFFF, 0xFEDCBA9876543210, 1234567890123456789])
def test_counter_uniform_equals_the_scalar_restatement(seed):
    n, C = 37, 24
    u = counter_uniform(seed, n, C)
    assert u.shape == (n, C) and u.dtype == np.float32
    for idx in (0, 1, 23, 24, 500, n * C - 1):
        assert u[idx // C, idx % C] == _uniform_scalar(seed, idx)
    assert 0.0 <= u.min() and u.max() < 1.0


def test_counter_mask_is_a_scaled_bernoulli_mask():
    m = counter_mask(42, 2000, 128, 0.25)
    vals = set(np.unique(m.numpy()).tolist())
    assert vals == {0.0, np.float32(1.0) / np.float32(0.75)}
    assert abs(float((m == 0).float().mean()) - 0.25) < 0.01
    assert not torch.equal(m, counter_mask(43, 2000, 128, 0.25))
    assert torch.equal(m, counter_mask(42, 2000, 128, 0.25))


def test_injected_dropout_hands_out_masks_in_order_and_restores():
    orig = F.dropout
    x = torch.ones(3, 4)
    m1, m2 = torch.full((3, 4), 2.0), torch.zeros(3, 4)
    with injected_dropout([m1, m2]) as st:
        assert torch.equal(F.dropout(x, 0.5, True), m1)
        assert torch.equal(F.dropout(x, 0.5, False), x)        # eval mode passes through, consumes nothing
        assert torch.equal(F.dropout(x, 0.0, True), x)
        assert st.left() == 1
        assert torch.equal(F.dropout(x, 0.5, True), m2)
        with pytest.raises(AssertionError):
            F.dropout(x, 0.5, True)
    assert F.dropout is orig
    with injected_dropout([torch.ones(2, 2)]):
        with pytest.raises(AssertionError):
            F.dropout(x, 0.5, True)                              # wrong shape
    assert F.dropout is orig


def test_oracle_steps_use_the_injected_masks():
    """Two oracle runs with the same injected masks give identical trajectories whatever torch's generator state is; different
    masks give different ones."""
    import efficient_gnns_amd.data as D
    import oracle.models as OM
    data = D.arxiv_like(scale=0.01, seed=2)
    dc = TP.oracle_data(data)
    hp = dict(alpha=0.9, kd_T=4.0, beta=0.1, nce_T=0.075, max_samples=64, kernel="cosine", proj_dim=16)

    def run(mask_seed, torch_seed):
        torch.manual_seed(0)
        om = OM.GCN(data.num_features, 32, data.num_classes, 3, 0.5)
        opt = torch.optim.Adam(om.parameters(), lr=0.01)
        torch.manual_seed(torch_seed)
        masks = [[counter_mask(mask_seed + 10 * s + l, data.num_nodes, 32, 0.5) for l in range(2)] for s in range(3)]
        return TP.oracle_steps((om, None, None, opt), dc, "kd", hp, None, masks, 5)
    a, b, c = run(1, 11), run(1, 22), run(2, 11)
    assert a == b and a != c
    assert TP.rel_errors(a, b) == 0.0 and TP.rel_errors(a, c) > 0


def test_capture_audit_flags_long_reductions_only():
    from efficient_gnns_amd._audit import CaptureAudit, LongReductionInCapture
    x = torch.rand(100000)
    with CaptureAudit(any_device=True) as a:
        x.view(100, 1000).sum(1)
        x.max()
        F.kl_div(torch.log(x + 1), x, reduction="mean")
        x.view(1000, 100).softmax(-1)
    names = [f[0] for f in a.flagged]
    assert names == ["max", "mean"], a.flagged
    with pytest.raises(LongReductionInCapture):
        a.check("unit test")
    with CaptureAudit(any_device=True) as b:
        (x * 2).view(1000, 100).sum(1)
    b.check("unit test")


def test_tensor_keyed_cache_is_lru_and_pins_inside_a_capture_block():
    """efficient-gnns_amd/_cache.py: identity-keyed entries, least-recently-used eviction one at a time, and values handed out inside
    ``pinning()`` stay referenced by the caller's list after the cache has evicted them (what a captured hipGraph needs)."""
    import gc
    import weakref
    from efficient_gnns_amd._cache import TensorKeyedCache, pinning
    c = TensorKeyedCache(capacity=2)
    keys = [torch.arange(4) + i for i in range(4)]
    built = []

    def make(i):
        def build():
            built.append(i)
            return torch.full((3,), float(i))
        return build
    a = c.get((keys[0],), (7,), make(0))
    assert c.get((keys[0],), (7,), make(0)) is a and built == [0]            # hit: same object, not rebuilt
    assert c.get((keys[0],), (8,), make(10)) is not a                         # another extra key: another entry
    c.get((keys[1],), (7,), make(1))                                          # capacity 2: evicts the least recently used = (keys[0], 7)
    assert len(c) == 2 and built == [0, 10, 1]
    c.get((keys[0],), (7,), make(0))
    assert built == [0, 10, 1, 0], "the evicted entry is rebuilt"
    keys[1].add_(1)                                                          # an in-place edit changes the version: not the same data
    c.get((keys[1],), (7,), make(11))
    assert built[-1] == 11
    with pinning() as pinned:
        v = c.get((keys[2],), (), make(2))
        ref = weakref.ref(v)
    for i in range(3):                                                       # push it out of the cache
        c.get((keys[3],), (i,), make(30 + i))
    del v
    gc.collect()
    assert ref() is not None, "a value handed out inside pinning() must outlive its eviction"
    del pinned
    gc.collect()
    assert ref() is None


def test_dropin_accel_keeps_torch_paths_for_cpu_inputs_and_restores():
    """dropin/accel.py on the CPU: CPU tensors keep torch's own BatchNorm1d / Linear (same values as before enable()), disable() restores
    the original methods; enable() twice is harmless."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("egnn_dropin_accel_cpu", os.path.join(ROOT, "efficient-gnns_amd", "dropin", "accel.py"))
    accel = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(accel)
    bn0, lin0, getitem0 = torch.nn.BatchNorm1d.forward, torch.nn.Linear.forward, torch.Tensor.__getitem__
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU())
    x = torch.randn(32, 8)
    want = net(x)
    accel.enable()
    accel.enable()
    try:
        assert accel.enabled() and torch.nn.BatchNorm1d.forward is not bn0
        # tensor indexing goes through accel's guard while it is enabled: CPU tensors (and every index kind but a big constant row gather
        # on the GPU) index exactly as before
        t = torch.arange(12.0).view(3, 4)
        assert torch.Tensor.__getitem__ is not getitem0 and torch.equal(t[torch.tensor([2, 0])], t.index_select(0, torch.tensor([2, 0])))
        assert float(t[1, 2]) == 6.0 and torch.equal(t[:, 1], torch.tensor([1.0, 5.0, 9.0])) and torch.equal(t[t > 10], torch.tensor([11.0]))
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU())
        net2.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
        net2[1].running_mean.zero_(); net2[1].running_var.fill_(1.0); net2[1].num_batches_tracked.zero_()
        got = net2(x)
    finally:
        accel.disable()
    assert torch.nn.BatchNorm1d.forward is bn0 and torch.nn.Linear.forward is lin0 and not accel.enabled()
    assert torch.Tensor.__getitem__ is getitem0
    assert torch.allclose(got, want, atol=1e-6)
    # round 6: under accel, Adam groups with equal options are stepped with ONE fused launch; parameters bit-equal to three launches, the
    # optimizer's param_groups / state_dict keep the script's three groups, Adam.step is restored by disable()
    torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))])       # (the first Optimizer ever built wraps the class's step with torch's profiling hook)
    step0 = torch.optim.Adam.step

    def run(on):
        torch.manual_seed(1)
        mods = [torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 2)]
        if on:
            accel.enable()
        try:
            opt = torch.optim.Adam([{"params": m.parameters(), "lr": 0.01} for m in mods], fused=True)
            for _ in range(3):
                opt.zero_grad()
                mods[2](mods[1](mods[0](torch.ones(5, 4)))).sum().backward()
                opt.step()
            assert len(opt.param_groups) == 3 and len(opt.state_dict()["param_groups"]) == 3
            assert (torch.optim.Adam.step is not step0) == on
        finally:
            if on:
                accel.disable()
        return [p.detach().clone() for m in mods for p in m.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(run(False), run(True))) and torch.optim.Adam.step is step0
    # groups with DIFFERENT options are never merged
    accel.enable()
    try:
        a, b = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
        opt = torch.optim.Adam([{"params": a.parameters(), "lr": 0.1}, {"params": b.parameters(), "lr": 0.0}], fused=True)
        b0 = b.weight.detach().clone()
        (a(torch.ones(1, 2)) + b(torch.ones(1, 2))).sum().backward()
        opt.step()
        assert torch.equal(b.weight, b0) and not torch.equal(a.weight, torch.zeros_like(a.weight))
    finally:
        accel.disable()


def test_deferred_activation_protocol_on_host():
    """efficient_gnns_amd/lazy.py without a GPU: which torch calls a deferred BatchNorm activation ABSORBS (F.relu, then F.dropout in
    training; the nn.ReLU module; F.dropout outside training is the identity) and that everything else would materialise (here: raise,
    there is no kernel on the CPU -- which is the point: nothing is skipped silently)."""
    import torch.nn.functional as F
    from efficient_gnns_amd import _lib
    from efficient_gnns_amd.lazy import LazyBnAct, LazyRows, materialise
    bn = torch.nn.BatchNorm1d(8)
    src = dict(x=torch.randn(16, 8), bn=bn, mean=torch.zeros(8), var=torch.ones(8), use_batch=True, training=True)
    a = LazyBnAct(src)
    assert tuple(a.shape) == (16, 8) and a.dim() == 2 and a.size(0) == 16 and len(a) == 16 and a.dtype == torch.float32
    r = F.relu(a)
    assert isinstance(r, LazyBnAct) and r._relu and r._p == 0.0 and r is not a and not a._relu
    assert torch.nn.ReLU()(a)._relu
    assert F.relu(r) is r
    d = F.dropout(r, p=0.5, training=True)
    assert isinstance(d, LazyBnAct) and d._relu and d._p == 0.5
    assert F.dropout(r, p=0.5, training=False) is r and F.dropout(r, p=0.0, training=True) is r
    for other in (lambda: torch.sigmoid(d), lambda: d + 1.0, lambda: d[torch.arange(4)], lambda: d.sum(), lambda: materialise(d)):
        with pytest.raises(RuntimeError):       # forming the tensor needs the kernels: no GPU here
            other()
    assert d._value is None
    plain = torch.randn(5, 3)
    assert materialise(plain) is plain and torch.equal(materialise(plain, torch.tensor([2, 0])), plain[[2, 0]])
    # inference: a deferred conv becomes a deferred fold under an eval-mode BatchNorm, which absorbs the ReLU (and F.dropout outside training)
    import types
    from efficient_gnns_amd.lazy import LazyConv, LazyFold
    conv = types.SimpleNamespace(out_channels=8)
    ran = []
    lc = LazyConv(conv, torch.randn(16, 4), "adj", lambda c, x, a, **kw: ran.append(kw) or torch.ones(16, 8))
    assert tuple(lc.shape) == (16, 8) and lc._value is None
    fold = LazyFold(lc, bn.eval())
    fr = F.dropout(F.relu(fold), p=0.5, training=False)
    assert isinstance(fr, LazyFold) and fr._relu and not fold._relu and fr._value is None and not ran
    assert torch.equal(materialise(lc), torch.ones(16, 8)) and ran == [{}]                  # any other consumer runs the conv as it is
    lr = LazyRows(plain, torch.tensor([4, 1, 3]))      # rows of a real tensor: a plain gather (torch indexing is plumbing, not a kernel)
    assert tuple(lr.shape) == (3, 3) and lr._value is None
    assert torch.equal(materialise(lr), plain[[4, 1, 3]]) and torch.equal(materialise(lr, torch.tensor([2])), plain[[3]])
