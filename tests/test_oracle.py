"""CPU tier: the oracle (test infrastructure) is pinned against the golden vectors that
tests/golden/make_golden.py captured from the REAL reference, against SURVEY.md's known
answers, and -- when oracle/_ref is present -- bit-for-bit against the reference's own
libdistance headers compiled from /root/reference."""
import os
import warnings

import numpy as np
import pytest

from oracle.libdistance_oracle import Oracle, Ref, VECTOR_METRICS
from oracle.tica_oracle import TicaOracle, lagged_moments

METRICS = sorted(set(VECTOR_METRICS))


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("metric", METRICS)
def test_libdistance_oracle_vs_golden(oracle, golden_dir, metric):
    g = np.load(os.path.join(golden_dir, "libdistance_golden.npz"))
    idx = g["idx"]
    for dn, dt in (("f32", np.float32), ("f64", np.float64)):
        for tag, (A, B) in (("g", (g["X"], g["Y"])), ("r", (g["Xr"], g["Yr"]))):
            A, B = A.astype(dt), B.astype(dt)
            p = "%s_%s_%s_" % (metric, dn, tag)
            with np.errstate(all="ignore"):
                assert _same(oracle.cdist(A, B, metric), g[p + "cdist"])
                lab, inertia = oracle.assign_nearest(A, B, metric)
                assert np.array_equal(lab, g[p + "assign"])
                assert _same(np.float64(inertia), g[p + "inertia"])
                lab, inertia = oracle.assign_nearest(A, B, metric, idx)
                assert np.array_equal(lab, g[p + "assign_idx"])
                assert _same(np.float64(inertia), g[p + "inertia_idx"])
                assert _same(oracle.dist(A, B[2], metric), g[p + "dist"])
                assert _same(oracle.dist(A, B[2], metric, idx), g[p + "dist_idx"])


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (needs /root/reference once)")
def test_libdistance_oracle_vs_compiled_reference(oracle):
    ref = Ref()
    rs = np.random.RandomState(0)
    for trial in range(12):
        n, k, f = rs.randint(1, 200), rs.randint(1, 30), rs.randint(1, 40)
        for dt in (np.float32, np.float64):
            X, Y = rs.randn(n, f).astype(dt), rs.randn(k, f).astype(dt)
            if trial % 3 == 0:
                X, Y = np.round(X).astype(dt), np.round(Y).astype(dt)
            if trial % 4 == 0:
                Y[: min(k, n)] = X[: min(k, n)]
            idx = rs.randint(0, n, size=7).astype(np.int64)
            for m in METRICS:
                with np.errstate(all="ignore"):
                    assert _same(oracle.cdist(X, Y, m), ref.cdist(X, Y, m)), (m, dt)
                    l1, i1 = oracle.assign_nearest(X, Y, m)
                    l2, i2 = ref.assign_nearest(X, Y, m)
                    assert np.array_equal(l1, l2) and _same(np.float64(i1), np.float64(i2)), (m, dt)
                    l1, i1 = oracle.assign_nearest(X, Y, m, idx)
                    l2, i2 = ref.assign_nearest(X, Y, m, idx)
                    assert np.array_equal(l1, l2) and _same(np.float64(i1), np.float64(i2))
                    assert _same(oracle.dist(X, Y[0], m), ref.dist(X, Y[0], m))
                    assert _same(oracle.dist(X, Y[0], m, idx), ref.dist(X, Y[0], m, idx))
                    if n <= 60:
                        assert _same(oracle.pdist(X, m), ref.pdist(X, m))
                        assert _same(oracle.pdist(X, m, idx), ref.pdist(X, m, idx))
                    pairs = rs.randint(0, n, size=(9, 2)).astype(np.int64)
                    assert _same(np.float64(oracle.sumdist(X, m, pairs)), np.float64(ref.sumdist(X, m, pairs)))


def test_libdistance_oracle_error_contract(oracle):
    X, Y = np.zeros((3, 2), np.float32), np.zeros((2, 2), np.float32)
    with pytest.raises(ValueError):
        oracle.assign_nearest(X, Y, "nope")
    with pytest.raises(TypeError):
        oracle.assign_nearest(X, Y.astype(np.float64), "euclidean")
    # all-NaN row keeps assignment 0 and adds DBL_MAX (assign.hpp:20-28)
    Xn = np.full((1, 2), np.nan, np.float32)
    lab, inertia = oracle.assign_nearest(Xn, Y, "euclidean")
    assert lab[0] == 0 and inertia == np.finfo(np.float64).max


def test_kcenters_oracle_vs_golden(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "kcenters_golden.npz"))
    rs = np.random.RandomState(1)
    X = np.concatenate([rs.randn(23, 2).astype(np.float32), rs.randn(10, 2).astype(np.float32)])
    ids, labels, dist = oracle.kcenters_fit(X, 3, "euclidean", 0)
    assert list(ids) == [0, 21, 16] == list(g["K1_ids"])
    assert np.array_equal(labels, g["K1_labels"]) and np.array_equal(dist, g["K1_distances"])
    assert np.sum(dist) == 29.00724663036992 == float(g["K1_inertia"])
    Xk = np.concatenate([g["K2_seq%d" % i] for i in range(3)])
    for metric in ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis"):
        for dt, dn in ((np.float32, "f32"), (np.float64, "f64")):
            p = "K2_%s_%s_" % (metric, dn)
            ids, labels, dist = oracle.kcenters_fit(Xk.astype(dt), 12, metric, int(g[p + "ids"][0]))
            assert np.array_equal(ids, g[p + "ids"]) and np.array_equal(labels, g[p + "labels"])
            assert np.array_equal(dist, g[p + "distances"])
            lab, _ = oracle.assign_nearest(Xk.astype(dt), np.ascontiguousarray(Xk.astype(dt)[ids]), metric)
            assert np.array_equal(lab, g[p + "predict"])


def test_tica_oracle_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "tica_golden.npz"))
    rs = np.random.RandomState(0)
    seqs = [rs.randn(1000, 6).astype(np.float32) for _ in range(3)] + [rs.randn(2, 6).astype(np.float32)]
    for tag, shr in (("A0", 0), ("An", None)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = TicaOracle(n_components=3, lag_time=2, shrinkage=shr).fit(seqs)
        assert [o.n_observations_, o.n_sequences_] == list(g[tag + "_n_obs_seq"]) == [3000, 3]
        np.testing.assert_allclose(o.eigenvalues_, g[tag + "_eigenvalues"], rtol=1e-12)
        np.testing.assert_allclose(o.timescales_, g[tag + "_timescales"], rtol=1e-11)
        np.testing.assert_array_equal(o.means_, g[tag + "_means"])
        np.testing.assert_array_equal(o.offset_correlation_, g[tag + "_offset_correlation"])
        np.testing.assert_allclose(o.covariance_, g[tag + "_covariance"], rtol=1e-14, atol=1e-17)
        np.testing.assert_allclose(o.shrinkage_, g[tag + "_shrinkage_"], rtol=1e-13)
        np.testing.assert_allclose(np.abs(o.eigenvectors_), np.abs(g[tag + "_eigenvectors"]), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(np.abs(o.transform(seqs[:1])[0]), np.abs(g[tag + "_transform0"]), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(o.score(seqs[1:3]), g[tag + "_score_test"], rtol=1e-10)
    # SURVEY.md section 8(c) known answers
    o = TicaOracle(n_components=3, lag_time=2, shrinkage=0).fit(seqs)
    np.testing.assert_allclose(o.eigenvalues_, [0.030889057382, 0.024348710243, 0.010004331246], rtol=1e-9)
    np.testing.assert_allclose(o.timescales_, [0.575150073765, 0.538317956886, 0.434335323084], rtol=1e-9)
    np.testing.assert_allclose(o.means_[:3], [-0.020693226642, -0.027541397681, -0.015588389737], rtol=1e-9)
    o = TicaOracle(n_components=3, lag_time=2).fit(seqs)
    np.testing.assert_allclose(o.eigenvalues_, [0.031653761128, 0.024321041356, 0.010195102652], rtol=1e-9)


def test_tica_oracle_ragged_and_mappings(golden_dir):
    g = np.load(os.path.join(golden_dir, "tica_golden.npz"))
    seqs = [g["B_seq%d" % i] for i in range(5)]
    for tag, kw in (("B", {}), ("Bk", dict(kinetic_mapping=True)), ("Bc", dict(commute_mapping=True))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = TicaOracle(n_components=4, lag_time=7, **kw).fit(seqs)
        np.testing.assert_allclose(o.eigenvalues_, g[tag + "_eigenvalues"], rtol=1e-12)
        Y, Yg = o.transform(seqs[1:2])[0], g[tag + "_transform1"]
        np.testing.assert_allclose(np.abs(Y), np.abs(Yg), rtol=1e-7, atol=1e-9)
        if tag == "B":
            np.testing.assert_array_equal(o.C, g["B_C"])
            np.testing.assert_array_equal(o.S0, g["B_S0"])
            np.testing.assert_array_equal(o.Stau, g["B_Stau"])
            np.testing.assert_array_equal(o.s0, g["B_s0"])
            np.testing.assert_array_equal(o.stau, g["B_stau"])
            assert [o.n_observations_, o.n_sequences_] == list(g["B_n_obs_seq"])
    assert lagged_moments(seqs[4], 7) is None      # len == lag: skipped
    with pytest.raises(ValueError):
        TicaOracle(lag_time=500).fit(seqs)


# ------------------------------------------------------------------ transition counts
def _transition_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "transition_golden.npz"), allow_pickle=False)
    for name in ("ident", "lag7", "gaps", "nosw", "neg"):
        seqs = [g["%s_seq%d" % (name, i)] for i in range(int(g[name + "_nseq"]))]
        yield name, seqs, int(g[name + "_lag"]), bool(g[name + "_sw"]), g[name + "_counts"], g[name + "_keys"]
    yield "nan", [g["nan_seq0"], g["nan_seq0"][:100]], 2, True, g["nan_counts"], g["nan_keys"]
    yield "str", [g["str_seq0"]], 1, True, g["str_counts"], g["str_keys"]


def test_transition_oracle_matches_reference_golden(golden_dir):
    from oracle.transition_oracle import transition_counts
    for name, seqs, lag, sw, counts, keys in _transition_cases(golden_dir):
        c, m = transition_counts(seqs, lag_time=lag, sliding_window=sw)
        assert np.array_equal(c, counts), name
        assert list(m.keys()) == list(keys) and list(m.values()) == list(range(len(keys))), name


def test_transition_oracle_reference_known_answers():
    """tests/test_transition_counts.py of the reference, restated."""
    from oracle.transition_oracle import transition_counts as tc
    with pytest.raises(ValueError):
        tc([1, 2, 3])
    c, m = tc([np.arange(10)])
    assert np.array_equal(c, np.eye(10, k=1)) and list(m.keys()) == list(range(10))
    assert np.array_equal(tc([range(10)], lag_time=2)[0], 0.5 * np.eye(10, k=2))
    c, m = tc([['alpha', 'b', 'b', 'b', 'c']])
    assert np.array_equal(c, [[0, 1, 0], [0, 2, 1], [0, 0, 0]]) and m == {'alpha': 0, 'b': 1, 'c': 2}
    c, m = tc([[100000000, 100000000, 100000001, 100000001]])
    assert np.array_equal(c, [[1, 1], [0, 1]]) and m == {100000000: 0, 100000001: 1}
    c, m = tc([[0, np.nan]])
    assert m == {0: 0} and np.array_equal(c, np.zeros((1, 1)))
    c, m = tc([[np.nan]])
    assert m == {} and c.shape == (0, 0)
    np.testing.assert_array_almost_equal(tc([np.arange(6)], lag_time=3)[0], np.eye(6, k=3) / 3)
