"""Host-side finalisation of the tICA accumulators (float64 numpy, O(F^2) / O(F^3)).

The device hands back four sums over all lagged pairs (t, t+tau) of all trajectories:

    C    = sum x_t x_{t+tau}^T                     (lagged second moment)
    G    = sum x_t x_t^T + x_{t+tau} x_{t+tau}^T   (instantaneous, both ends)
    s0   = sum x_t,   stau = sum x_{t+tau}
    n_pairs = n_observations - tau * n_sequences

from which the symmetrised estimators of the reference are formed
(/root/reference/msmbuilder/decomposition/tica.py:228-259):

    mu = (s0 + stau) / (2 n_pairs)
    offset_correlation = (C + C^T) / (2 n_pairs) - mu mu^T
    S                  =  G        / (2 n_pairs) - mu mu^T
    covariance         = (1 - rho) S + rho (tr S / p) I

with rho either given or the Rao-Blackwellised Ledoit-Wolf estimate
(tica.py:492-524, Chen/Wiesel/Hero ICASSP 2009) evaluated with n = n_observations.
"""
import numpy as np
import scipy.linalg


def pair_count(n_observations, n_sequences, lag_time):
    return n_observations - lag_time * n_sequences


def mean_vector(s0, stau, n_pairs):
    return (s0 + stau) / float(2 * n_pairs)


def offset_correlation(C, mu, n_pairs):
    return (C + C.T) / (2 * n_pairs) - np.outer(mu, mu)


def sample_covariance(G, mu, n_pairs):
    return G / (2 * n_pairs) - np.outer(mu, mu)


def rblw_shrinkage(S, n):
    """Rao-Blackwell Ledoit-Wolf intensity rho in [0, 1] for a p x p sample covariance."""
    p = S.shape[0]
    tr = np.trace(S)
    alpha = (n - 2) / (n * (n + 2))
    beta = ((p + 1) * n - 2) / (n * (n + 2))
    U = p * np.sum(S * S) / tr ** 2 - 1
    return min(alpha + beta / U, 1)


def shrink(S, rho):
    """(1 - rho) S + rho * (tr S / p) * I"""
    p = S.shape[0]
    out = (1 - rho) * S
    out[np.diag_indices(p)] += rho * np.trace(S) / p
    return out


def rao_blackwell_ledoit_wolf(S, n):
    """Same call signature and return value as the reference's helper: (sigma, shrinkage)."""
    p = len(S)
    assert S.shape == (p, p)
    rho = rblw_shrinkage(S, n)
    F = (np.trace(S) / p) * np.eye(p)
    return (1 - rho) * S + rho * F, rho


def _blas_threads(F):
    """BLAS/LAPACK worker threads for an F x F solve: a handful.  Measured for a 512 x 512 dsygvx
    on the GPU boxes' hosts: 1 thread 10-26 ms (depends on the CPU), 2-8 threads 9.4-9.7 ms, all
    256 threads 17 ms -- and, more importantly, a large OpenBLAS pool keeps spinning for tens of
    ms after the call, which starves the thread that is about to enqueue hundreds of small
    kernel launches (the 200 k-centers launches went from 31 ms to 48-90 ms right after a
    256-thread solve)."""
    return 4 if F <= 768 else 8


DEVICE_SOLVE_MIN_FEATURES = 1536   # measured: F = 1024 hybrid 25.3 ms vs all-device 26.8 ms (host dsygvx 49.6); F = 2048 all-device 62 ms vs host 175 ms


def _use_device_solve(F):
    """MSMBUILDER_AMD_DEVICE_SOLVE = 1 / 0 forces the device / host solver; default: by size."""
    import os
    env = os.environ.get("MSMBUILDER_AMD_DEVICE_SOLVE", "auto")
    if env in ("0", "1"):
        return env == "1"
    return F >= DEVICE_SOLVE_MIN_FEATURES


def solve_mode(F):
    """Where ``tICA._solve`` runs: "hybrid" (device finalise + Cholesky reduction + back-substitution, host dsyevr on
    the reduced matrix; default below DEVICE_SOLVE_MIN_FEATURES), "device" (rocSOLVER dsyevd as well; default from
    there), or "host" (round 1's numpy finalisation + dsygvx).  MSMBUILDER_AMD_DEVICE_SOLVE = 0 -> host, 1 -> device,
    hybrid -> hybrid."""
    import os
    env = os.environ.get("MSMBUILDER_AMD_DEVICE_SOLVE", "auto")
    if env == "0":
        return "host"
    if env == "1":
        return "device"
    if env == "hybrid":
        return "hybrid"
    return "device" if F >= DEVICE_SOLVE_MIN_FEATURES else "hybrid"


def top_standard_eigenpairs(Cs, k):
    """k largest eigenpairs of the symmetric ``Cs`` (destroyed), eigenvalues descending, eigenvectors as ROWS of a
    C-contiguous k x F array.  LAPACK dsyevr on ONE thread: at F = 512 the call is its memory-bound tridiagonalisation
    (5.7 ms; 6.9 ms on 4 threads, measured on the GPU boxes' hosts)."""
    F = Cs.shape[0]
    with _blas_limit(1 if F <= 768 else _blas_threads(F)):
        vals, vecs = scipy.linalg.eigh(Cs, subset_by_index=[F - k, F - 1], driver="evr", overwrite_a=True,
                                       check_finite=False)
    order = np.argsort(vals)[::-1]
    return vals[order], np.ascontiguousarray(vecs[:, order].T)


def use_subspace_solve(F, k):
    """The device route of the hybrid solve (csrc/subspace.hip behind ``msm_tica_solve_topk``) applies to up to 16
    components of 128 .. 1,024 features; everything else takes LAPACK's dsyevr on the reduced matrix."""
    return 128 <= F <= 1024 and 1 <= k <= 16


def device_generalized_eigenpairs(lhs, rhs, k):
    """The same k largest eigenpairs through ``msm_sygv_top`` (rocSOLVER dsygvd on the GPU)."""
    import ctypes as C
    from .. import _lib
    F = lhs.shape[0]
    a = np.ascontiguousarray(lhs, dtype=np.float64)
    b = np.ascontiguousarray(rhs, dtype=np.float64)
    vals = np.empty(k)
    vecs = np.empty((k, F))
    _lib.ensure_device()
    rc = _lib.lib().msm_sygv_top(a.ctypes.data, b.ctypes.data, F, k, vals.ctypes.data, vecs.ctypes.data, 0)
    if rc == _lib.MSM_ERR_INVALID and "positive definite" in _lib.last_error():
        raise np.linalg.LinAlgError(_lib.last_error())
    _lib.check(rc)
    return vals, np.ascontiguousarray(vecs.T)


def top_generalized_eigenpairs(lhs, rhs, k):
    """k largest solutions of lhs v = lambda rhs v, eigenvalues descending
    (LAPACK dsygvx through scipy, as tica.py:188-194; rocSOLVER dsygvd on the device for large F)."""
    F = lhs.shape[0]
    if _use_device_solve(F):
        return device_generalized_eigenpairs(lhs, rhs, k)
    with _blas_limit(_blas_threads(F)):
        vals, vecs = scipy.linalg.eigh(lhs, b=rhs, subset_by_index=[F - k, F - 1])
    order = np.argsort(vals)[::-1]
    return vals[order], vecs[:, order]


_controller = None


def _blas_limit(n_threads):
    """threadpoolctl context LOWERING the BLAS thread count to ``n_threads``.  The controller is built ONCE:
    constructing it walks every loaded shared object (1.3 ms per solve when done per call).  The count is never
    raised: under ``torchrun`` every rank starts with OMP_NUM_THREADS=1, OpenBLAS sizes its per-thread buffers for
    that, and asking it for 4 threads afterwards crashed dsygvx with SIGSEGV on both ranks of a 2-rank bench."""
    global _controller
    import contextlib
    try:
        if _controller is None:
            from threadpoolctl import ThreadpoolController
            _controller = ThreadpoolController()
        current = [int(lib.num_threads) for lib in _controller.lib_controllers if lib.user_api == "blas"]
        if not current or min(current) <= int(n_threads):
            return contextlib.nullcontext()
        return _controller.limit(limits=int(n_threads), user_api="blas")
    except Exception:  # threadpoolctl missing: solve with whatever BLAS does
        return contextlib.nullcontext()


def is_symmetric(a, rtol=1e-05, atol=1e-08):
    """``np.allclose(a, a.T)`` (tica.py:183-186), cheaper: the moments built here are symmetric
    bit for bit, so an exact comparison settles it; otherwise the same element-wise criterion
    |a - a.T| <= atol + rtol |a.T| (NaN fails it, as it does there) on contiguous copies."""
    if np.array_equal(a, a.T):
        return True
    at = np.ascontiguousarray(a.T)
    d = a - at
    np.abs(d, out=d)
    np.abs(at, out=at)
    at *= rtol
    at += atol
    return bool((d <= at).all())
