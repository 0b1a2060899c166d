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
