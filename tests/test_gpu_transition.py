"""GPU parity of msmbuilder_amd.msm._transition_counts (int64 pair counting on the device) against
the reference's own outputs (tests/golden/transition_golden.npz), the oracle, and the reference's
tests/test_transition_counts.py cases.  Integer work: results must be identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "transition_golden.npz"), allow_pickle=False)
    for name in ("ident", "lag7", "gaps", "nosw", "neg"):
        seqs = [g["%s_seq%d" % (name, i)] for i in range(int(g[name + "_nseq"]))]
        yield name, seqs, int(g[name + "_lag"]), bool(g[name + "_sw"]), g[name + "_counts"], g[name + "_keys"]
    yield "nan", [g["nan_seq0"], g["nan_seq0"][:100]], 2, True, g["nan_counts"], g["nan_keys"]
    yield "str", [g["str_seq0"]], 1, True, g["str_counts"], g["str_keys"]


def test_golden(gpu, golden_dir):
    import torch
    from msmbuilder_amd.msm import _transition_counts
    for name, seqs, lag, sw, counts, keys in _cases(golden_dir):
        c, m = _transition_counts(seqs, lag_time=lag, sliding_window=sw)
        assert c.dtype == np.float64 and np.array_equal(c, counts), name
        assert list(m.keys()) == list(keys) and list(m.values()) == list(range(len(keys))), name
        if seqs[0].dtype.kind == "i":      # the same labels resident in HBM, and mixed placement
            dev = [torch.from_numpy(s).cuda() for s in seqs]
            c2, m2 = _transition_counts(dev, lag_time=lag, sliding_window=sw)
            assert np.array_equal(c2, counts) and list(m2.keys()) == list(keys), name
            c3, _ = _transition_counts([dev[0]] + list(seqs[1:]), lag_time=lag, sliding_window=sw)
            assert np.array_equal(c3, counts), name
            c4, _ = _transition_counts([s.astype(np.int32) for s in seqs], lag_time=lag, sliding_window=sw)
            assert np.array_equal(c4, counts), name


def test_reference_test_cases(gpu):
    """/root/reference/msmbuilder/tests/test_transition_counts.py, case by case."""
    from msmbuilder_amd.msm import _transition_counts
    with pytest.raises(ValueError):
        _transition_counts([1, 2, 3])
    c, m = _transition_counts([np.arange(10)])
    np.testing.assert_array_equal(c, np.eye(10, k=1))
    assert list(m.keys()) == list(range(10)) and list(m.values()) == list(range(10))
    c, m = _transition_counts([range(10)], lag_time=2)
    np.testing.assert_array_equal(c, 0.5 * np.eye(10, k=2))
    c, m = _transition_counts([['alpha', 'b', 'b', 'b', 'c']])
    np.testing.assert_array_equal(c, 1.0 * np.array([[0, 1, 0], [0, 2, 1], [0, 0, 0]]))
    assert m == {'alpha': 0, 'b': 1, 'c': 2}
    c, m = _transition_counts([[100000000, 100000000, 100000001, 100000001]])
    np.testing.assert_array_equal(c, 1.0 * np.array([[1, 1], [0, 1]]))
    assert m == {100000000: 0, 100000001: 1}
    c, m = _transition_counts([np.array([100000000, 100000000, 9100000001, 9100000001])])   # span > dense limit
    np.testing.assert_array_equal(c, 1.0 * np.array([[1, 1], [0, 1]]))
    assert m == {100000000: 0, 9100000001: 1}
    c, m = _transition_counts([[0]])
    assert c.shape == (1, 1) and c[0, 0] == 0
    c, m = _transition_counts([[0, np.nan]])
    assert m == {0: 0}
    np.testing.assert_array_equal(c, np.zeros((1, 1)))
    c, m = _transition_counts([[np.nan]])
    assert m == {}
    np.testing.assert_array_equal(c, np.zeros((0, 0)))
    X = np.arange(6)
    C, _ = _transition_counts([X], lag_time=3)
    np.testing.assert_array_almost_equal(C, np.eye(6, k=3) / 3)
    X = np.arange(10)
    C1, m1 = _transition_counts([X], lag_time=3, sliding_window=False)
    C2, m2 = _transition_counts([X[::3]], sliding_window=True)
    np.testing.assert_array_almost_equal(C1, C2)
    assert m1 == m2


@pytest.mark.parametrize("k,lag", [(3, 1), (200, 10), (1000, 100)])
def test_vs_oracle_random(gpu, k, lag):
    from msmbuilder_amd.msm import _transition_counts
    from oracle.transition_oracle import transition_counts
    rs = np.random.RandomState(k + lag)
    seqs = []
    for n in (20000, 3, lag, lag + 1, 4097, 8192 + lag):
        y = np.empty(n, dtype=np.int64)
        y[0] = rs.randint(k)
        jump = rs.rand(n) > 0.9
        draws = rs.randint(0, k, size=n)
        for t in range(1, n):
            y[t] = draws[t] if jump[t] else y[t - 1]
        seqs.append(y)
    c, m = _transition_counts(seqs, lag_time=lag)
    co, mo = transition_counts(seqs, lag_time=lag)
    assert np.array_equal(c, co) and list(m.items()) == list(mo.items())


def test_full_size_labels_from_kcenters(gpu):
    """10M labels resident in HBM (the bench's frame count): conservation laws of the count matrix."""
    import torch
    from msmbuilder_amd.msm import _transition_counts
    torch.manual_seed(0)
    n_seq, T, K, lag = 1000, 10000, 200, 100
    stay = torch.rand(n_seq, T, device="cuda") < 0.98
    draws = torch.randint(0, K, (n_seq, T), device="cuda")
    idx = torch.arange(T, device="cuda").expand(n_seq, T)
    last_jump = torch.cummax(torch.where(~stay, idx, torch.zeros_like(idx)), dim=1).values
    labels = torch.gather(draws, 1, last_jump)                  # piecewise-constant (metastable) label streams
    seqs = list(labels.unbind(0))
    c, m = _transition_counts(seqs, lag_time=lag)
    assert c.shape == (K, K) and list(m.keys()) == list(range(K))
    raw = np.rint(c * lag).astype(np.int64)
    assert raw.sum() == n_seq * (T - lag)
    assert np.array_equal(raw.sum(1), torch.bincount(labels[:, :-lag].reshape(-1), minlength=K).cpu().numpy())
    assert np.array_equal(raw.sum(0), torch.bincount(labels[:, lag:].reshape(-1), minlength=K).cpu().numpy())
    pair = (labels[:, :-lag] * K + labels[:, lag:]).reshape(-1)
    assert np.array_equal(raw.reshape(-1), torch.bincount(pair, minlength=K * K).cpu().numpy())
