"""oracle/kpp_oracle.py -- TEST INFRASTRUCTURE, never imported by the product (msmbuilder_amd/).

A numpy restatement of scikit-learn's greedy k-means++ seeding, ``_kmeans_plusplus`` (sklearn/cluster/_kmeans.py:163-259:
the third-party arithmetic behind msmbuilder/cluster/__init__.py:67-69, version unpinned by the reference -- SURVEY 8(c)),
draw for draw: first centre by ``choice``, then ``2 + log(k)`` candidates per round by inverse-CDF sampling of the current
squared distances, keeping the candidate with the lowest potential.  It was the product's host path in rounds 1-3
(float32 ``||x||^2 - 2 x.c + ||c||^2`` through a BLAS sgemm); round 4 moved the rounds to the device
(msmbuilder_amd/csrc/kpp.hip) and this file became the checker: pinned against ``sklearn.cluster.kmeans_plusplus`` itself
in tests/test_host.py, and the device seeding is compared with both in tests/test_gpu_kmeans.py.
"""
import numpy as np


def kmeans_plusplus(X, n_clusters, random_state):
    """Greedy k-means++ seeding on a (small, host) sample, mirroring scikit-learn's
    ``_kmeans_plusplus`` draw for draw (sklearn/cluster/_kmeans.py:163-259): first centre by
    ``choice``, then ``2 + log(k)`` candidates per round drawn by inverse-CDF sampling of the
    current squared distances, keeping the candidate with the lowest potential."""
    n_samples, n_features = X.shape
    centers = np.empty((n_clusters, n_features), dtype=X.dtype)
    n_local_trials = 2 + int(np.log(n_clusters))
    sample_weight = np.ones(n_samples, dtype=X.dtype)
    center_id = random_state.choice(n_samples, p=sample_weight / sample_weight.sum())
    centers[0] = X[center_id]
    xsq = np.einsum("ij,ij->i", X, X)
    Xm2 = np.ascontiguousarray(-2.0 * X)  # -2 x.c as c.(-2 x): an exact scaling, one pass less per round
    d = np.empty((n_local_trials, n_samples), dtype=X.dtype)

    def sqdist(c, out):  # [len(c), n_samples], same ||x||^2 - 2 x.c + ||c||^2 form as sklearn
        np.dot(c, Xm2.T, out=out)
        out += xsq[None, :]
        out += np.einsum("ij,ij->i", c, c)[:, None]
        np.maximum(out, 0, out=out)
        return out

    closest = sqdist(centers[0:1], np.empty((1, n_samples), dtype=X.dtype))[0].copy()
    current_pot = closest @ sample_weight
    cum = np.empty(n_samples)
    _kpp_rounds(X, centers, n_clusters, n_local_trials, sample_weight, random_state, sqdist, d, closest, current_pot, cum)
    return centers


def _kpp_rounds(X, centers, n_clusters, n_local_trials, sample_weight, random_state, sqdist, d, closest, current_pot, cum):
    n_samples = X.shape[0]
    for c in range(1, n_clusters):
        rand_vals = random_state.uniform(size=n_local_trials) * current_pot
        np.cumsum(closest, dtype=np.float64, out=cum)   # sample_weight == 1
        candidate_ids = np.searchsorted(cum, rand_vals)
        np.clip(candidate_ids, None, n_samples - 1, out=candidate_ids)
        d_cand = sqdist(X[candidate_ids], d)
        np.minimum(closest, d_cand, out=d_cand)
        pots = d_cand @ sample_weight
        best = np.argmin(pots)
        current_pot = pots[best]
        closest = d_cand[best].copy()
        centers[c] = X[candidate_ids[best]]
    return centers


def kmeans_plusplus_f64(X, n_clusters, random_state):
    """The device seeding's arithmetic (msmbuilder_amd/csrc/kpp.hip) in numpy: scikit-learn's draws and its float64-upcast
    distances rounded to float32, but float64 potentials (scikit-learn: a float32 dot over the rows, whose rounding --
    ~1e-5 relative over 2e5 rows -- moves every inverse-CDF draw by a row or two, so its picks at such sizes depend on the
    BLAS build).  What the device seeds are compared with beyond a few thousand rows."""
    n, F = X.shape
    L = 2 + int(np.log(n_clusters))
    X64 = X.astype(np.float64)
    xx = np.einsum("ij,ij->i", X64, X64)
    w = np.ones(n, dtype=X.dtype)
    first = int(random_state.choice(n, p=w / w.sum()))
    u = random_state.uniform(size=(max(n_clusters - 1, 0), L))

    def sq(rows):
        c = X64[rows]
        d = -2.0 * (c @ X64.T)
        d += np.einsum("ij,ij->i", c, c)[:, None]
        d += xx[None, :]
        return np.maximum(d, 0.0).astype(np.float32)
    ids = [first]
    closest = sq([first])[0]
    pot = float(np.float32(closest.astype(np.float64).sum()))
    for c in range(1, n_clusters):
        cum = np.cumsum(closest, dtype=np.float64)
        cand = np.minimum(np.searchsorted(cum, u[c - 1] * pot), n - 1)
        d = np.minimum(closest[None, :], sq(cand))
        pots = d.astype(np.float64).sum(axis=1).astype(np.float32)
        b = int(np.argmin(pots))
        pot, closest = float(pots[b]), d[b]
        ids.append(int(cand[b]))
    return X[ids].copy(), np.asarray(ids)
