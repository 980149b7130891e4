"""The subclass seam of tICA: the reference's SparseTICA / KSparseTICA / KernelTICA override ``_solve``
(reading ``offset_correlation_`` / ``covariance_``, writing ``_eigenvalues_`` / ``_eigenvectors_`` / ``_is_dirty``:
sparsetica.py:139-167, ksparsetica.py:155-191) or ``fit`` / ``partial_fit`` / ``transform`` (ktica.py:195-212) and rely
on ``_fit``, ``_initialized`` and the lazily solved properties of the base class.  Restated subclasses of that shape on
the new tICA, default f32 mode, against the float64 oracle; plus the import of a reference object's accumulators."""
import os
import warnings

import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sequences(seed, F=24, lengths=(900, 1400, 650)):
    rs = np.random.RandomState(seed)
    M = rs.randn(5, F)
    out = []
    for n in lengths:
        z = np.zeros((n, 5))
        a = np.exp(-1.0 / np.array([40.0, 25.0, 12.0, 6.0, 3.0]))
        e = rs.randn(n, 5)
        for t in range(1, n):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * e[t]
        out.append((z.dot(M) + 0.5 * rs.randn(n, F) + 2.0).astype(np.float32))
    return out


def _thresholded_pairs(A, B, k, cut):
    """What the toy subclass computes: dense pencil, top k, small loadings zeroed, B-renormalised."""
    w, V = scipy.linalg.eigh(A, B)
    w, V = w[::-1][:k], V[:, ::-1][:, :k]
    V = np.where(np.abs(V) < cut * np.abs(V).max(axis=0), 0.0, V)
    V = V / np.sqrt(np.einsum("ik,ij,jk->k", V, B, V))
    vals = np.einsum("ik,ij,jk->k", V, A, V)
    order = np.argsort(vals)[::-1]
    return vals[order], V[:, order]


def _make_sparse_class():
    from msmbuilder_amd import tICA

    class ThresholdTICA(tICA):
        """SparseTICA-shaped: extra hyper-parameters, positional super().__init__, its own _solve."""

        def __init__(self, n_components=None, lag_time=1, rho=0.01, kinetic_mapping=False, commute_mapping=False,
                     shrinkage=None):
            super(ThresholdTICA, self).__init__(n_components, lag_time=lag_time, kinetic_mapping=kinetic_mapping,
                                                commute_mapping=commute_mapping)
            self.rho = rho
            self.shrinkage = shrinkage
            self.solves = 0

        def _solve(self):
            if not self._is_dirty:
                return
            if self.rho <= 0:
                return super(ThresholdTICA, self)._solve()
            self.solves += 1
            A = self.offset_correlation_
            B = self.covariance_
            self._eigenvalues_, self._eigenvectors_ = _thresholded_pairs(A, B, self.n_components, self.rho)
            self._is_dirty = False

    return ThresholdTICA


def test_solve_override_matches_oracle(gpu):
    from oracle.tica_oracle import TicaOracle
    cls = _make_sparse_class()
    seqs = _sequences(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = cls(n_components=3, lag_time=5, rho=0.2).fit(seqs)
        o = TicaOracle(n_components=3, lag_time=5).fit(seqs)
        assert m._initialized and m._is_dirty and m.solves == 0
        vals, V = _thresholded_pairs(o.offset_correlation_, o.covariance_, 3, 0.2)
        np.testing.assert_allclose(m.eigenvalues_, vals, rtol=1e-5)
        assert m.solves == 1 and not m._is_dirty
        assert (m.eigenvectors_ == 0).sum() == (V == 0).sum() > 0          # the override's vectors are what is served
        sign = np.sign((m.eigenvectors_ * V).sum(0))
        np.testing.assert_allclose(m.eigenvectors_ * sign, V, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(m.timescales_, -5.0 / np.log(vals), rtol=1e-4)
        assert m.solves == 1                                               # cached until the accumulators change
        # transform goes through the override's vectors and the base class's means
        Y = m.transform(seqs[:1])[0]
        Yr = (seqs[0].astype(np.float64) - o.means_).dot(V) * sign
        np.testing.assert_allclose(Y, Yr, rtol=1e-3, atol=2e-4)
        assert m.summarize().startswith("time-structure based Independent Components Analysis")

        # partial_fit dirties the model; the override runs again on the new moments, and means_ follow the data
        extra = _sequences(5, lengths=(800,))[0] + np.float32(3.0)
        m.partial_fit(extra)
        o.partial_fit(extra)
        assert m._is_dirty
        vals2, V2 = _thresholded_pairs(o.offset_correlation_, o.covariance_, 3, 0.2)
        np.testing.assert_allclose(m.eigenvalues_, vals2, rtol=1e-5)
        assert m.solves == 2
        np.testing.assert_allclose(m.means_, o.means_, rtol=1e-6)

        # rho <= 0: the base class solve through super()
        b = cls(n_components=3, lag_time=5, rho=0.0).fit(seqs)
        o0 = TicaOracle(n_components=3, lag_time=5).fit(seqs)
        np.testing.assert_allclose(b.eigenvalues_, o0.eigenvalues_, rtol=1e-5)
        assert b.solves == 0 and not b._is_dirty


def test_means_after_device_solve_then_partial_fit_then_override(gpu):
    """a mean cached by the device-side solve must not survive new data when a later solve does not refresh it
    (a subclass's host _solve, or MSMBUILDER_AMD_DEVICE_SOLVE=0)."""
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    seqs = _sequences(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=2, lag_time=3).fit(seqs)
        m.eigenvalues_                                   # device solve: leaves its mean behind
        shifted = seqs[0] + np.float32(10.0)
        m.partial_fit(shifted)
        m._eigenvalues_ = np.zeros(2)                    # what a subclass's _solve does: results set by hand ...
        m._eigenvectors_ = np.zeros((m.n_features, 2))
        m._is_dirty = False                              # ... and the flag cleared, without a device solve
        o = TicaOracle(n_components=2, lag_time=3).fit(seqs)
        o.partial_fit(shifted)
        np.testing.assert_allclose(m.means_, o.means_, rtol=1e-6)
        np.testing.assert_allclose(m.covariance_, o.covariance_, rtol=0, atol=1e-5 * np.abs(o.covariance_).max())


def test_kernel_tica_shaped_subclass(gpu):
    """KernelTICA's shape (ktica.py:195-212): fit / partial_fit / transform feed the base class through a feature map."""
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle

    def lift(X):
        X = np.asarray(X, dtype=np.float64)
        return np.concatenate([X, np.tanh(X[:, :6])], axis=1)

    class MappedTICA(tICA):
        def fit(self, sequences, y=None):
            super(MappedTICA, self).fit([lift(s) for s in sequences], y=y)      # (returns None, like the reference)

        def partial_fit(self, X):
            super(MappedTICA, self).partial_fit(lift(X))

        def transform(self, sequences):
            return super(MappedTICA, self).transform([lift(s) for s in sequences])

    seqs = _sequences(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MappedTICA(n_components=3, lag_time=4)
        assert m.fit(seqs) is None
        m.partial_fit(seqs[1])
        o = TicaOracle(n_components=3, lag_time=4).fit([lift(s) for s in seqs])
        o.partial_fit(lift(seqs[1]))
        assert m.n_features == 30 and m.n_sequences_ == 4
        np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=1e-9)    # float64 inputs: the fp64 kernel
        Y, Yr = m.transform(seqs[:1])[0], o.transform([lift(seqs[0])])[0]
        np.testing.assert_allclose(Y * np.sign((Y * Yr).sum(0)), Yr, rtol=1e-6, atol=1e-8)


def test_import_of_reference_accumulators(gpu):
    """A reference object's attribute dict (the raw accumulators the real tica.py produced: golden B_*) through
    msm_tica_import: the fitted attributes are the reference's, and partial_fit carries on from them."""
    from msmbuilder_amd import tICA
    from oracle.tica_oracle import TicaOracle
    g = np.load(os.path.join(HERE, "golden", "tica_golden.npz"), allow_pickle=True)
    F = g["B_C"].shape[0]
    state = dict(n_components=4, lag_time=7, shrinkage=None, kinetic_mapping=False, commute_mapping=False,
                 n_features=F, n_observations_=int(g["B_n_obs_seq"][0]), n_sequences_=int(g["B_n_obs_seq"][1]),
                 _outer_0_to_T_lagged=g["B_C"], _sum_0_to_TminusTau=g["B_s0"], _sum_tau_to_T=g["B_stau"],
                 _sum_0_to_T=None, _outer_0_to_TminusTau=g["B_S0"], _outer_offset_to_T=g["B_Stau"],
                 _initialized=True, _is_dirty=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA.from_reference_state(state)
        assert (m.n_observations_, m.n_sequences_) == tuple(int(v) for v in g["B_n_obs_seq"])
        np.testing.assert_allclose(m.eigenvalues_, g["B_eigenvalues"], rtol=1e-10)
        np.testing.assert_allclose(m.means_, g["B_means"], rtol=1e-12)
        np.testing.assert_allclose(m.shrinkage_, float(g["B_shrinkage_"]), rtol=1e-10)
        Y = m.transform([g["B_seq1"]])[0]
        np.testing.assert_allclose(Y * np.sign((Y * g["B_transform1"]).sum(0)), g["B_transform1"], rtol=1e-6, atol=1e-8)
        # carrying on: the imported state + new data == the oracle fed everything
        seqs = [g["B_seq%d" % i] for i in range(5)]
        extra = seqs[0][::-1].copy()
        m.partial_fit(extra)
        o = TicaOracle(n_components=4, lag_time=7).fit(seqs)
        o.partial_fit(extra)
        tol = 1e-5 if extra.dtype == np.float32 else 1e-10
        np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=tol)
        # a subclass can be the target too, and an unfitted state gives an unfitted model
        cls = _make_sparse_class()
        s = cls.from_reference_state(state, rho=0.0)
        np.testing.assert_allclose(s.eigenvalues_, g["B_eigenvalues"], rtol=1e-10)
        assert not tICA.from_reference_state(dict(n_components=2, lag_time=3))._initialized
    with pytest.raises(ValueError):
        tICA.from_reference_state(dict(state, _sum_tau_to_T=np.zeros(F + 1)))
