"""Building blocks of the LAPACK-free solve tail (csrc/toppairs.hip) against LAPACK on the host: the blocked Cholesky,
the k largest eigenpairs of a symmetric tridiagonal matrix (multisection + inverse iteration), and the whole
msm_tica_solve_topk path against the host dsygvx route on wide models (reference: tica.py:167-199)."""
import ctypes as C
import warnings

import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 5, 32, 33, 171, 512, 1000])
def test_potrf_matches_lapack(gpu, n):
    from msmbuilder_amd import _lib
    rs = np.random.RandomState(n)
    M = rs.randn(n, n + 5)
    B = M.dot(M.T) / n + 0.1 * np.eye(n)
    out = B.copy()
    info = C.c_int(-1)
    _lib.check(_lib.lib().msm_potrf(out.ctypes.data, n, C.byref(info), 0))
    assert info.value == 0
    U = np.linalg.cholesky(B).T
    np.testing.assert_allclose(np.triu(out), U, rtol=1e-11, atol=1e-13)
    assert np.array_equal(np.tril(out, -1), np.tril(B, -1))          # the other triangle is left alone, like LAPACK
    # not positive definite from column p on: LAPACK's info convention
    if n >= 5:
        bad = B.copy()
        p = n // 2
        bad[p, p] = -1.0
        _lib.check(_lib.lib().msm_potrf(bad.ctypes.data, n, C.byref(info), 0))
        assert info.value == p + 1


def _tridiag_cases():
    rs = np.random.RandomState(0)
    cases = []
    for n, k in ((3, 3), (6, 4), (64, 8), (300, 5), (512, 10), (512, 40), (1024, 12)):
        cases.append(("random", rs.randn(n), rs.randn(n - 1), k))
    # a real reduction: clustered top eigenvalues (three equal to rounding, one 1e-9 below)
    n = 300
    Q, _ = np.linalg.qr(rs.randn(n, n))
    w = np.linspace(-1, 0.9, n)
    w[-3:] = 0.95
    w[-4] = 0.95 - 1e-9
    T = scipy.linalg.hessenberg((Q * w).dot(Q.T))
    cases.append(("cluster", np.diag(T).copy(), np.diag(T, -1).copy(), 6))
    # tICA-like: 16 slow modes above a noise bulk
    n = 512
    Q, _ = np.linalg.qr(rs.randn(n, n))
    w = np.r_[rs.uniform(-0.05, 0.05, n - 16), np.exp(-100 / np.logspace(np.log10(20), np.log10(5000), 16))]
    T = scipy.linalg.hessenberg((Q * w).dot(Q.T))
    cases.append(("tica", np.diag(T).copy(), np.diag(T, -1).copy(), 10))
    # zero couplings: the matrix splits into blocks (and exact duplicates across blocks)
    d = np.r_[rs.randn(40), rs.randn(40)]
    e = rs.randn(79)
    e[39] = 0.0
    d[40:] = d[:40]
    e[40:] = e[:39]
    cases.append(("split", d, e, 7))
    cases.append(("diagonal", np.arange(20.0), np.zeros(19), 5))
    cases.append(("scaled", 1e-150 * rs.randn(100), 1e-150 * rs.randn(99), 4))
    cases.append(("big", 1e120 * rs.randn(100), 1e120 * rs.randn(99), 4))
    return cases


@pytest.mark.parametrize("case", _tridiag_cases(), ids=lambda c: "%s-%d-%d" % (c[0], len(c[1]), c[3]))
def test_tridiag_topk_matches_lapack(gpu, case):
    from msmbuilder_amd import _lib
    name, d, e, k = case
    n = len(d)
    d = np.ascontiguousarray(d)
    e = np.ascontiguousarray(e)
    vals, vecs = np.empty(k), np.empty((k, n))
    _lib.check(_lib.lib().msm_tridiag_topk(d.ctypes.data, e.ctypes.data, n, k, vals.ctypes.data, vecs.ctypes.data, 0))
    w = scipy.linalg.eigh_tridiagonal(d, e, eigvals_only=True)[::-1][:k]
    nrm = max(np.abs(d).max(), np.abs(e).max() if n > 1 else 0.0)
    np.testing.assert_allclose(vals, w, rtol=0, atol=4 * n * np.finfo(float).eps * nrm)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    res = np.abs(T.dot(vecs.T) - vecs.T * vals).max()
    assert res <= 1e-13 * n * nrm, res
    np.testing.assert_allclose(vecs.dot(vecs.T), np.eye(k), rtol=0, atol=1e-12)


@pytest.mark.parametrize("F,k,flat", [(200, 10, False), (512, 10, False), (512, 64, False), (700, 3, False), (512, 10, True),
                                      (130, 8, False)])
def test_topk_solve_matches_host_route(gpu, monkeypatch, F, k, flat):
    """msm_tica_solve_topk on wide models against the all-host numpy / dsygvx route of the same accumulators: through the
    subspace iteration where the spectrum has a gap (k <= 16), through the tridiagonalisation where it has none (`flat`:
    white-noise features -- the iteration must notice that it stalls and hand over) or k is large, and with the subspace
    iteration switched off."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    rs = np.random.RandomState(F + k)
    if flat:
        X = rs.randn(6000, F) + rs.randn(F)
    else:
        z = np.cumsum(rs.randn(6000, 12), axis=0) * 0.02
        z -= z.mean(0)
        X = (np.tanh(z).dot(rs.randn(12, F)) + 0.4 * rs.randn(6000, F) + rs.randn(F)).astype(np.float64)
    seqs = [X[:3500], X[3500:]]
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, env in (("host", {"MSMBUILDER_AMD_DEVICE_SOLVE": "0"}),
                          ("topk", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid", "MSMBUILDER_AMD_DEVICE_TOPK": "1", "MSM_SOLVE_SUBSPACE": "1"}),
                          ("direct", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid", "MSMBUILDER_AMD_DEVICE_TOPK": "1", "MSM_SOLVE_SUBSPACE": "0"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            m = tICA(n_components=k, lag_time=5).fit(seqs)
            out[name] = (m.eigenvalues_.copy(), m.eigenvectors_.copy(), m.covariance_.copy(), getattr(m, "_solve_route", None))
    assert out["direct"][3][0] == "tridiagonal" and out["direct"][3][2] == 0
    route = out["topk"][3]
    assert route[2] == 0
    if k <= 16 and F >= 128:
        assert route[0] == ("tridiagonal" if flat else "subspace"), route
    else:
        assert route[0] == "tridiagonal"
    for name in ("topk", "direct"):
        np.testing.assert_allclose(out[name][0], out["host"][0], rtol=1e-10)
        V, Vh, S = out[name][1], out["host"][1], out["host"][2]
        np.testing.assert_allclose(V.T.dot(S).dot(V), np.eye(k), rtol=0, atol=1e-9)      # B-orthonormal like dsygvx
        # eigenvectors of well separated eigenvalues agree up to sign; inside near-degenerate groups compare the subspace
        gaps = np.abs(np.diff(out["host"][0]))
        for j in range(k):
            lo = gaps[j - 1] if j > 0 else np.inf
            hi = gaps[j] if j < k - 1 else np.inf
            if min(lo, hi) > 1e-4:
                sg = np.sign(V[:, j].dot(S).dot(Vh[:, j]))
                np.testing.assert_allclose(V[:, j] * sg, Vh[:, j], rtol=0, atol=1e-8 * np.abs(Vh[:, j]).max() / min(lo, hi, 1.0))
