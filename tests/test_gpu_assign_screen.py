"""assign_nearest on float64 rows of at most 16 features (KCenters.predict on a tICA projection) runs a float32 screen over
the centres and the reference's exact arithmetic only for the centres the screen cannot rule out (csrc/distance_small.hip,
assign_screen_kernel).  Labels AND distances must be those of assign.hpp:6-91 bit for bit -- first index on ties, NaN rows,
duplicated centres -- whatever the screen thinks: against the C oracle and against the exact kernel (MSM_ASSIGN_SCREEN=0)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle.libdistance_oracle import Oracle
    return Oracle()


def _assign(X, Y):
    """labels, inertia, distances through the C ABI (host arrays)."""
    from msmbuilder_amd import _lib
    from msmbuilder_amd._lib import check
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    n, m = X.shape
    lab = np.empty(n, dtype=np.int64)
    dist = np.empty(n, dtype=np.float64)
    inertia = C.c_double(0.0)
    check(_lib.lib().msm_assign_nearest_f64(C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), b"euclidean", None,
                                               n, Y.shape[0], m, n, C.c_void_p(lab.ctypes.data), C.c_void_p(dist.ctypes.data),
                                               C.byref(inertia), 0))
    return lab, float(inertia.value), dist


def _check(X, Y, oracle, monkeypatch):
    with np.errstate(all="ignore"):
        lab_o, inertia_o, mind_o = oracle.assign_nearest(X, Y, "euclidean", return_distances=True)
    monkeypatch.setenv("MSM_ASSIGN_SCREEN", "1")
    lab, inertia, dist = _assign(X, Y)
    monkeypatch.setenv("MSM_ASSIGN_SCREEN", "0")
    lab_e, inertia_e, dist_e = _assign(X, Y)
    assert np.array_equal(lab_e, lab_o)
    assert np.array_equal(lab, lab_o), np.flatnonzero(lab != lab_o)[:10]
    ok = np.isfinite(mind_o)
    assert np.array_equal(dist[ok], mind_o[ok])
    assert np.array_equal(dist, dist_e, equal_nan=True)
    assert (np.isnan(inertia) and np.isnan(inertia_o)) or inertia == inertia_o or abs(inertia - inertia_o) <= 1e-13 * abs(inertia_o)
    assert inertia == inertia_e or (np.isnan(inertia) and np.isnan(inertia_e))


@pytest.mark.parametrize("n,k,f", [(16384, 2, 1), (20000, 9, 2), (17001, 37, 3), (30000, 200, 10), (16500, 256, 7), (16500, 257, 10),
                                   (16400, 1000, 5), (20000, 50, 16), (20000, 64, 15), (40000, 200, 9)])
def test_random_rows_with_hits_and_duplicates(gpu, oracle, monkeypatch, n, k, f):
    rs = np.random.RandomState(n + k + f)
    X = rs.randn(n, f)
    Y = X[rs.choice(n, k, replace=False)].copy() if k > 40 else rs.randn(k, f)
    Y[: min(k, 3)] = X[: min(k, 3)]               # exact hits
    if k > 4:
        Y[4] = Y[1]                                # identical centres: the lower index wins
    _check(X, Y, oracle, monkeypatch)


@pytest.mark.parametrize("what", ["offset", "lattice", "nonfinite", "huge", "tiny", "scales", "one_cluster", "constant"])
def test_adversarial_rows(gpu, oracle, monkeypatch, what):
    rs = np.random.RandomState(len(what))
    n, k, f = 20000, 60, 10
    X = rs.randn(n, f)
    Y = X[rs.choice(n, k, replace=False)].copy()
    if what == "offset":                            # far from the origin: the float copies are centred on centre 0
        X += 3e6
        Y += 3e6
    elif what == "lattice":                         # ties everywhere: nearly every row is undecidable for the screen
        X = rs.randint(-3, 4, size=(n, f)).astype(np.float64)
        Y = rs.randint(-3, 4, size=(k, f)).astype(np.float64)
    elif what == "nonfinite":
        X[5, 3] = np.nan
        X[700, 0] = np.inf
        X[701, 9] = -np.inf
        X[16000] = np.nan
    elif what == "huge":                            # rows beyond the float32 range, and near its end
        X[11] = 1e200
        X[12] = 1e19
        X[13, 2] = -3e38
        X[14] = 1.7e308
    elif what == "tiny":
        X *= 1e-30
        Y *= 1e-30
        X[40:80] *= 1e-20                           # float32 denormals / zeros
    elif what == "scales":                          # columns of very different scale
        s = 10.0 ** rs.randint(-6, 7, size=f)
        X *= s
        Y *= s
    elif what == "one_cluster":                     # all centres within 1e-7 of each other relative to their norm
        Y = 1.0 + 1e-7 * rs.randn(k, f)
        X = 1.0 + 1e-7 * rs.randn(n, f)
    elif what == "constant":
        X[:] = 0.25
        Y[:] = 0.25
        Y[7] = 0.5
    _check(X, Y, oracle, monkeypatch)


def test_nan_centre_disables_the_screen(gpu, oracle, monkeypatch):
    rs = np.random.RandomState(3)
    X = rs.randn(20000, 6)
    Y = rs.randn(30, 6)
    Y[11, 2] = np.nan
    _check(X, Y, oracle, monkeypatch)
    Y[11, 2] = 1e30
    _check(X, Y, oracle, monkeypatch)


def test_kcenters_predict_on_a_projection(gpu, oracle, monkeypatch):
    """The bench's shape in small: KCenters on correlated 10-dimensional rows, predict == the labels of fit."""
    torch = pytest.importorskip("torch")
    from msmbuilder_amd import KCenters
    rs = np.random.RandomState(0)
    Z = np.cumsum(rs.randn(60000, 10), axis=0) * 0.01 + rs.randn(60000, 10)
    km = KCenters(n_clusters=100, random_state=0).fit([Z])
    lab = km.predict([Z])[0]
    assert np.array_equal(lab, km.labels_[0])
    lab_o, _, _ = oracle.assign_nearest(Z, km.cluster_centers_, "euclidean", return_distances=True)
    assert np.array_equal(lab, lab_o)
    Zd = torch.from_numpy(Z).cuda()
    labd = km.predict([Zd])[0]
    assert np.array_equal(labd.cpu().numpy(), lab_o)
