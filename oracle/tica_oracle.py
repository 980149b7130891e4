"""oracle/tica_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/scipy restatement of the arithmetic of the reference's tICA estimator,
``/root/reference/msmbuilder/decomposition/tica.py``.  Every function cites the
reference lines it follows.  Only tests/, ``__graft_entry__.smoke()`` and
bench.py's ``cpu_baseline`` leg may import this module; the product path
(msmbuilder_amd/) never does.

Parity status: PINNED.  tests/golden/make_golden.py imports the reference's
own tica.py (by file path, with an ``mdtraj`` stub and an ``eigh(eigvals=)``
shim -- SURVEY.md section 8(c)) in the dev container and stores its outputs in
tests/golden/tica_*.npz; tests/test_oracle_tica.py checks this restatement
against those vectors and against SURVEY.md's captured known answers.
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.linalg


def lagged_moments(X, lag_time):
    """One trajectory's contribution to the six accumulators, tica.py:401-422.

    Returns ``None`` for a trajectory with ``len(X) <= lag_time`` (tica.py:410-412:
    warned about and skipped, counters untouched), otherwise a dict with

    * ``C``    = X[:-tau].T @ X[tau:]            (tica.py:417)
    * ``s0``   = X[:-tau].sum(0)                 (tica.py:418)
    * ``stau`` = X[tau:].sum(0)                  (tica.py:419)
    * ``sall`` = X.sum(0)                        (tica.py:420, never read again)
    * ``S0``   = X[:-tau].T @ X[:-tau]           (tica.py:421)
    * ``Stau`` = X[tau:].T @ X[tau:]             (tica.py:422)

    all float64 on the float64 up-cast of X (tica.py:402).
    """
    X = np.asarray(np.atleast_2d(X), dtype=np.float64)
    if not len(X) > lag_time:
        return None
    a, b = X[:-lag_time], X[lag_time:]
    return dict(C=np.dot(a.T, b), s0=a.sum(axis=0), stau=b.sum(axis=0), sall=X.sum(axis=0),
                S0=np.dot(a.T, a), Stau=np.dot(b.T, b), n=X.shape[0])


def rao_blackwell_ledoit_wolf(S, n):
    """tica.py:492-524 (Chen, Wiesel & Hero 2009)."""
    p = len(S)
    assert S.shape == (p, p)
    alpha = (n - 2) / (n * (n + 2))
    beta = ((p + 1) * n - 2) / (n * (n + 2))
    trace_S2 = np.sum(S * S)
    U = ((p * trace_S2 / np.trace(S) ** 2) - 1)
    rho = min(alpha + beta / U, 1)
    F = (np.trace(S) / p) * np.eye(p)
    return (1 - rho) * S + rho * F, rho


class TicaOracle:
    """State machine of tica.py:108-259 with the reference's attribute names."""

    def __init__(self, n_components=None, lag_time=1, shrinkage=None,
                 kinetic_mapping=False, commute_mapping=False):
        # tica.py:108-148
        self.n_components = n_components
        self.lag_time = lag_time
        self.shrinkage = shrinkage
        self.shrinkage_ = None
        self.kinetic_mapping = kinetic_mapping
        self.commute_mapping = commute_mapping
        if kinetic_mapping and commute_mapping:
            raise ValueError("Can't have both kinetic mapping and commute mapping. "
                             "Please only use one.")
        self.n_features = None
        self.n_observations_ = None
        self.n_sequences_ = None
        self._initialized = False

    def _initialize(self, n_features):
        # tica.py:150-165
        if self._initialized:
            return
        if self.n_components is None:
            self.n_components = n_features
        self.n_features = n_features
        self.n_observations_ = 0
        self.n_sequences_ = 0
        self.C = np.zeros((n_features, n_features))
        self.s0 = np.zeros(n_features)
        self.stau = np.zeros(n_features)
        self.sall = np.zeros(n_features)
        self.S0 = np.zeros((n_features, n_features))
        self.Stau = np.zeros((n_features, n_features))
        self._initialized = True

    def partial_fit(self, X):
        # tica.py:401-424
        X = np.asarray(np.atleast_2d(X), dtype=np.float64)
        self._initialize(X.shape[1])
        mom = lagged_moments(X, self.lag_time)
        if mom is None:
            warnings.warn("length of data (%d) is too short for the lag time (%d)"
                          % (len(X), self.lag_time))
            return self
        self.n_observations_ += mom["n"]
        self.n_sequences_ += 1
        self.C += mom["C"]
        self.s0 += mom["s0"]
        self.stau += mom["stau"]
        self.sall += mom["sall"]
        self.S0 += mom["S0"]
        self.Stau += mom["Stau"]
        return self

    def fit(self, sequences):
        # tica.py:261-290
        self._initialized = False
        for X in sequences:
            self.partial_fit(X)
        if self.n_sequences_ == 0:
            raise ValueError('All sequences were shorter than the lag time, %d' % self.lag_time)
        return self

    # ---- finalisation: tica.py:228-259 ----
    @property
    def two_N(self):
        return 2 * (self.n_observations_ - self.lag_time * self.n_sequences_)

    @property
    def means_(self):
        return (self.s0 + self.stau) / float(self.two_N)

    @property
    def offset_correlation_(self):
        term = (self.C + self.C.T) / self.two_N
        mu = self.means_
        return term - np.outer(mu, mu)

    @property
    def covariance_(self):
        term = (self.S0 + self.Stau) / self.two_N
        mu = self.means_
        S = term - np.outer(mu, mu)
        if self.shrinkage is None:
            sigma, self.shrinkage_ = rao_blackwell_ledoit_wolf(S, n=self.n_observations_)
        else:
            self.shrinkage_ = self.shrinkage
            p = self.n_features
            F = (np.trace(S) / p) * np.eye(p)
            sigma = (1 - self.shrinkage) * S + self.shrinkage * F
        return sigma

    def solve(self):
        """tica.py:167-199: top-k generalized symmetric-definite eigenpairs, descending."""
        lhs, rhs = self.offset_correlation_, self.covariance_
        F, k = self.n_features, self.n_components
        vals, vecs = scipy.linalg.eigh(lhs, b=rhs, subset_by_index=[F - k, F - 1])
        ind = np.argsort(vals)[::-1]
        return vals[ind], vecs[:, ind]

    @property
    def eigenvalues_(self):
        return self.solve()[0]

    @property
    def eigenvectors_(self):
        return self.solve()[1]

    @property
    def timescales_(self):
        # tica.py:219-222
        return -1. * self.lag_time / np.log(self.eigenvalues_)

    def transform(self, sequences):
        # tica.py:312-354
        vals, vecs = self.solve()
        out = []
        for X in sequences:
            X = np.asarray(np.atleast_2d(X))
            Y = np.dot(X - self.means_, vecs)
            if self.kinetic_mapping:
                Y *= vals
            if self.commute_mapping:
                ts = -1. * self.lag_time / np.log(vals)
                reg = 0.5 * ts * np.tanh(np.pi * ((ts - self.lag_time) / self.lag_time) + 1)
                Y *= np.sqrt(reg / 2)
                Y = np.nan_to_num(Y)
            out.append(Y)
        return out

    def score(self, sequences):
        # tica.py:426-467 (GMRQ)
        V = self.eigenvectors_
        m2 = TicaOracle(shrinkage=self.shrinkage, n_components=self.n_components,
                        lag_time=self.lag_time)
        for X in sequences:
            m2.partial_fit(X)
        num = V.T.dot(m2.offset_correlation_).dot(V)
        den = V.T.dot(m2.covariance_).dot(V)
        try:
            return np.trace(num.dot(np.linalg.inv(den)))
        except np.linalg.LinAlgError:
            return np.nan


def reference_faithful_fit_seconds(X_list, lag_time):
    """The CPU baseline bench.py times: exactly the reference's operation
    sequence (f64 up-cast + three dgemm + three column sums, tica.py:402-422)
    through numpy's BLAS.  Returns (seconds, frames)."""
    import time
    t0 = time.perf_counter()
    n = 0
    o = TicaOracle(lag_time=lag_time)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for X in X_list:
            o.partial_fit(X)
            n += len(X)
    return time.perf_counter() - t0, n
