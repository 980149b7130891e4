"""Un-centred features (|mean| / std of 10 - 100, the normal case for contact and atom-pair distances: BASELINE
configs[2] is ContactFeaturizer, the reference's tests/workflows/basic.sh is AtomPairsFeaturizer).

The covariance is G / 2N - mu mu^T (/root/reference/msmbuilder/decomposition/tica.py:228-259), so an fp32
accumulation error of eps * |G| is a RELATIVE covariance error of eps * (mean / std)^2.  The reference is immune
because it up-casts every trajectory to float64 (tica.py:402); the fp32 / bf16 kernels here are immune because they
accumulate the moments of x - r (r = column means of the first launch) and restore the raw moments in fp64 at export
(csrc/tica.hip, "mean shift").  Everything below is compared with the float64 oracle at the STATED tolerance -- rtol
1e-5 on eigenvalues, and 1e-5 of their own scale (not of max|G|) on covariance_ / offset_correlation_."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _uncentred(seed, n_seq, n_frames, F, ratio, k=6):
    """AR(1) slow modes mixed into F features, every feature scaled to std 1, then offset by +-ratio."""
    rs = np.random.RandomState(seed)
    M = rs.randn(k, F)
    a = np.exp(-1.0 / (8.0 * (1 + np.arange(k))))
    sign = np.where(rs.rand(F) < 0.5, -1.0, 1.0)
    off = sign * ratio * rs.uniform(0.8, 1.2, size=F)
    out = []
    for _ in range(n_seq):
        eps = rs.randn(n_frames, k)
        z = np.zeros((n_frames, k))
        z[0] = eps[0]
        for t in range(1, n_frames):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * eps[t]
        x = z.dot(M) + 0.5 * rs.randn(n_frames, F)
        x /= np.sqrt((M * M).sum(0) + 0.25)      # unit variance per feature
        out.append((x + off).astype(np.float32))
    return out


def _fit(seqs, mode, monkeypatch, shift=True, **kw):
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    if not shift:
        monkeypatch.setenv("MSM_TICA_SHIFT", "0")
    else:
        monkeypatch.delenv("MSM_TICA_SHIFT", raising=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return tICA(**kw).fit(seqs)


def _oracle(seqs, **kw):
    from oracle.tica_oracle import TicaOracle
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return TicaOracle(**kw).fit(seqs)


@pytest.mark.parametrize("ratio", [10, 30, 100])
@pytest.mark.parametrize("F", [128, 171, 512])
def test_f32_default_on_uncentred_features(gpu, monkeypatch, F, ratio):
    """VERDICT r1 task 1: default f32 mode, eigenvalues rtol 1e-5, covariance_ / offset_correlation_ to 1e-5 of
    their own scale.  F = 128 and 171 take the C/G kernel (one tile; F % 4 != 0), F = 512 the sum/difference kernel."""
    lag = 20
    seqs = _uncentred(1000 * F + ratio, 4, 25000, F, ratio)
    seqs[1] = seqs[1][:8191]       # ragged, not a multiple of the 32-frame step
    m = _fit(seqs, "f32", monkeypatch, n_components=4, lag_time=lag)
    o = _oracle(seqs, n_components=4, lag_time=lag)
    np.testing.assert_allclose(m.means_, o.means_, rtol=1e-12)
    cov, oc = o.covariance_, o.offset_correlation_
    np.testing.assert_allclose(m.covariance_, cov, rtol=0, atol=1e-5 * np.abs(cov).max())
    np.testing.assert_allclose(m.offset_correlation_, oc, rtol=0, atol=1e-5 * np.abs(oc).max())
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=1e-5)
    np.testing.assert_allclose(m.timescales_, o.timescales_, rtol=1e-4)
    # raw accumulators restored at export: the fp32 part of their error is relative to the CENTRED second moment
    # (2 N' sigma^2 with sigma = 1 here), the rest is fp64 rounding of the mu^2-dominated totals
    m._pull()
    G = o.S0 + o.Stau
    atol = 1e-6 * o.two_N + 1e-11 * np.abs(G).max()
    np.testing.assert_allclose(m._outer_gram_sum, G, rtol=0, atol=atol)
    C = 0.5 * (o.C + o.C.T) if m._lagged_symmetrised else o.C
    np.testing.assert_allclose(m._outer_0_to_T_lagged, C, rtol=0, atol=atol)
    assert np.array_equal(m._outer_gram_sum, m._outer_gram_sum.T)


def test_shift_is_what_makes_it_pass(gpu, monkeypatch, capsys):
    """The same fit with the shift switched off (MSM_TICA_SHIFT=0): the raw fp32 accumulation misses the stated
    tolerance at ratio 100 (VERDICT r1's emulation predicted 1e-3), the shifted one meets it with margin."""
    F, lag, ratio = 512, 20, 100
    seqs = _uncentred(77, 4, 25000, F, ratio)
    o = _oracle(seqs, n_components=4, lag_time=lag)
    on = _fit(seqs, "f32", monkeypatch, True, n_components=4, lag_time=lag)
    e_on = np.abs(on.eigenvalues_ / o.eigenvalues_ - 1).max()
    off = _fit(seqs, "f32", monkeypatch, False, n_components=4, lag_time=lag)
    try:
        e_off = np.abs(off.eigenvalues_ / o.eigenvalues_ - 1).max()
    except np.linalg.LinAlgError:      # measured: the raw fp32 covariance is not even positive definite here
        e_off = np.inf
    S = (o.S0 + o.Stau) / o.two_N - np.outer(o.means_, o.means_)
    off._pull()
    S_off = off._outer_gram_sum / o.two_N - np.outer(off.means_, off.means_)
    on._pull()
    S_on = on._outer_gram_sum / o.two_N - np.outer(on.means_, on.means_)
    c_on, c_off = np.abs(S_on - S).max() / np.abs(S).max(), np.abs(S_off - S).max() / np.abs(S).max()
    with capsys.disabled():
        print("\n[uncentred ratio 100, F=512] eigenvalue rel. error: shift on %.2e, shift off %s; "
              "covariance error / max|cov|: on %.2e, off %.2e" % (e_on, e_off, c_on, c_off))
    assert e_on <= 1e-6 and c_on <= 1e-6
    assert e_off > 10 * e_on and c_off > 10 * c_on


@pytest.mark.parametrize("mode,rtol", [("bf16x2", 1e-5), ("bf16", 5e-3)])
@pytest.mark.parametrize("F", [128, 512])
def test_bf16_modes_on_uncentred_features(gpu, monkeypatch, mode, rtol, F):
    """bf16 rounding after the shift: the 8 (16) significant bits go to x - r.  Without it ratio 30 leaves bf16 with
    ~3 bits of signal."""
    lag, ratio = 20, 30
    seqs = _uncentred(5 + F, 4, 25000, F, ratio)
    m = _fit(seqs, mode, monkeypatch, n_components=4, lag_time=lag)
    o = _oracle(seqs, n_components=4, lag_time=lag)
    np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=rtol)
    cov = o.covariance_
    np.testing.assert_allclose(m.covariance_, cov, rtol=0, atol=(1e-5 if mode == "bf16x2" else 1e-2) * np.abs(cov).max())


@pytest.mark.parametrize("F", [128, 256])
def test_partial_fit_mixed_and_reset(gpu, monkeypatch, F):
    """r is fixed by the first shifted launch and later launches (other trajectories, float64 input through the fp64
    kernel, a second model re-using the pooled handle after reset) add up."""
    from msmbuilder_amd import tICA
    lag, ratio = 7, 50
    seqs = _uncentred(9 + F, 5, 6000, F, ratio)
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    o = _oracle(seqs, n_components=3, lag_time=lag)
    for rep in range(2):      # second round re-uses the parked handle: the shift must have been reset with it
        m = tICA(n_components=3, lag_time=lag)
        m.partial_fit(seqs[0])
        m.partial_fit(seqs[1].astype(np.float64))
        m.partial_fit(seqs[2] if rep == 0 else seqs[2].copy())
        m.partial_fit(seqs[3])
        m.partial_fit(seqs[4])
        np.testing.assert_allclose(m.eigenvalues_, o.eigenvalues_, rtol=1e-5)
        cov = o.covariance_
        np.testing.assert_allclose(m.covariance_, cov, rtol=0, atol=1e-5 * np.abs(cov).max())
        del m


@pytest.mark.parametrize("F", [128, 512])
def test_segments_on_uncentred_features(gpu, monkeypatch, F):
    """One long un-centred trajectory cut into rank-owned pieces (each model has its OWN r): the exported raw
    moments add up to the unsplit fit -- the shifted pairs' right-frame sums come from their own column-sum pass."""
    from msmbuilder_amd import tICA
    from msmbuilder_amd import parallel
    lag, ratio = 25, 40
    X = _uncentred(21 + F, 1, 30000, F, ratio)[0]
    Y = _uncentred(22 + F, 1, 7000, F, ratio)[0]
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f32")
    o = _oracle([X, Y], n_components=3, lag_time=lag)
    world = 3
    parts = []
    for r in range(world):
        m = tICA(n_components=3, lag_time=lag)
        pieces = []
        for i, ob, oe in parallel.split_frames([len(X), len(Y)], lag, rank_=r, world=world):
            seq = (X, Y)[i]
            end = min(oe + lag, len(seq))
            pieces.append((seq[ob:end], len(seq), ob, ob, oe))
        m.partial_fit_segments(pieces)
        m._pull()
        parts.append(m)
    C = sum(p._outer_0_to_T_lagged for p in parts)
    G = sum(p._outer_gram_sum for p in parts)
    s0 = sum(p._sum_0_to_TminusTau for p in parts)
    st = sum(p._sum_tau_to_T for p in parts)
    nobs = sum(p.n_observations_ for p in parts)
    nseq = sum(p.n_sequences_ for p in parts)
    assert (nobs, nseq) == (o.n_observations_, o.n_sequences_)
    npairs = nobs - lag * nseq
    mu = (s0 + st) / (2.0 * npairs)
    np.testing.assert_allclose(mu, o.means_, rtol=1e-12)
    cov = G / (2.0 * npairs) - np.outer(mu, mu)
    oc = (C + C.T) / (2.0 * npairs) - np.outer(mu, mu)
    S_ref = (o.S0 + o.Stau) / (2.0 * npairs) - np.outer(o.means_, o.means_)
    np.testing.assert_allclose(cov, S_ref, rtol=0, atol=1e-5 * np.abs(S_ref).max())
    np.testing.assert_allclose(oc, o.offset_correlation_, rtol=0, atol=1e-5 * np.abs(o.offset_correlation_).max())
