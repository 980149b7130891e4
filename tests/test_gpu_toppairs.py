"""The device pieces of tICA._solve against LAPACK on the host: the blocked Cholesky (csrc/toppairs.hip) and the whole
msm_tica_solve_topk path (subspace iteration, verified; LAPACK on the reduced matrix behind it) against the host dsygvx
route on wide models (reference: tica.py:167-199)."""
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


@pytest.mark.parametrize("F,k,flat", [(200, 10, False), (512, 10, False), (512, 64, False), (700, 3, False), (512, 10, True),
                                      (130, 8, False), (100, 5, False)])
def test_topk_solve_matches_host_route(gpu, monkeypatch, F, k, flat):
    """The hybrid solve on wide models against the all-host numpy / dsygvx route of the same accumulators: through the
    subspace iteration where the spectrum has a gap (k <= 16, F >= 128), through LAPACK on the reduced matrix where k is
    large or the model narrow; with no gap at all (`flat`: white-noise features) the iteration must either converge to
    verified pairs or notice that it stalls and hand over."""
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
        for name, env in (("host", {"MSMBUILDER_AMD_DEVICE_SOLVE": "0"}), ("hybrid", {"MSMBUILDER_AMD_DEVICE_SOLVE": "hybrid"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            m = tICA(n_components=k, lag_time=5).fit(seqs)
            out[name] = (m.eigenvalues_.copy(), m.eigenvectors_.copy(), m.covariance_.copy(), getattr(m, "_solve_route", None))
    route = out["hybrid"][3]
    if k <= 16 and F >= 128 and not flat:
        assert route[0] == "subspace" and route[2] == 0, route
    elif flat:
        # white noise: the iteration either gets there with its longer, chunked filters (verified pairs, status 0) or
        # notices that it stalls and hands the reduced matrix to LAPACK (status 1) -- the numbers below must hold either way
        assert (route[0] == "subspace" and route[2] == 0) or (route[0] == "lapack" and route[2] == 1), route
    else:
        assert route[0] == "lapack" and route[2] != 1, route
    for name in ("hybrid",):
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



@pytest.mark.parametrize("F,n_slow,neg", [(256, 3, False), (512, 6, False), (512, 2, True), (1024, 2, False)])
def test_topk_solve_components_reaching_into_the_noise_bulk(gpu, monkeypatch, F, n_slow, neg):
    """Fewer slow processes than components: eigenvalues n_slow + 1 .. k sit at the upper edge of the noise bulk, a few
    percent of the bulk's width apart.  The subspace iteration narrows its damped interval to the bulk (hard mode,
    csrc/subspace.hip) instead of handing over to LAPACK; `neg` adds an oscillating feature whose autocorrelation at the lag
    is about -0.9, i.e. an eigenvalue far BELOW the bulk that the guessed lower bound misses and the iteration has to
    find.  Whatever route is taken, the numbers must be the host route's (reference: tica.py:188-194, dsygvx)."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    rs = np.random.RandomState(F + n_slow)
    n, lag, k = 12000, 5, 10
    ts = np.logspace(np.log10(30.0), np.log10(600.0), n_slow)
    a = np.exp(-1.0 / ts)
    z = np.zeros((n, n_slow))
    e = rs.randn(n, n_slow) * np.sqrt(1 - a * a)
    for t in range(1, n):
        z[t] = a * z[t - 1] + e[t]
    X = z.dot(rs.randn(n_slow, F) / np.sqrt(n_slow)) + 0.5 * rs.randn(n, F) + rs.randn(F)
    if neg:
        X[:, 7] += 3.0 * np.cos(np.pi * 0.95 / lag * np.arange(n) + 0.3)
    seqs = [X[:7000], X[7000:]]
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, env in (("host", "0"), ("hybrid", "hybrid")):
            monkeypatch.setenv("MSMBUILDER_AMD_DEVICE_SOLVE", env)
            m = tICA(n_components=k, lag_time=lag).fit(seqs)
            out[name] = (m.eigenvalues_.copy(), m.eigenvectors_.copy(), m.covariance_.copy(), getattr(m, "_solve_route", None))
    ev = out["host"][0]
    assert ev[n_slow - 1] > 2 * ev[n_slow] > 0, ev            # the construction: n_slow processes, then the bulk's edge
    route = out["hybrid"][3]
    assert route[0] == "subspace" and route[2] == 0, route   # no LAPACK hand-over for these
    np.testing.assert_allclose(out["hybrid"][0], ev, rtol=0, atol=2e-11)
    V, Vh, S = out["hybrid"][1], out["host"][1], out["host"][2]
    np.testing.assert_allclose(V.T.dot(S).dot(V), np.eye(k), rtol=0, atol=1e-9)
    gaps = np.abs(np.diff(ev))
    for j in range(k):
        lo_ = gaps[j - 1] if j > 0 else np.inf
        hi_ = gaps[j] if j < k - 1 else np.inf
        g = min(lo_, hi_, 1.0)
        if g > 1e-4:
            sg = np.sign(V[:, j].dot(S).dot(Vh[:, j]))
            np.testing.assert_allclose(V[:, j] * sg, Vh[:, j], rtol=0, atol=1e-8 * np.abs(Vh[:, j]).max() / g)
    # the span of all k vectors (near-degenerate pairs included): projectors agree in the Sigma inner product
    P = V.T.dot(S).dot(Vh)
    np.testing.assert_allclose(P.dot(P.T), np.eye(k), rtol=0, atol=1e-6)
