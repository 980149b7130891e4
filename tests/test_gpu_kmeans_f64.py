"""Round 6: MiniBatchKMeans on FLOAT64 rows.

msmbuilder.cluster.MiniBatchKMeans is scikit-learn's estimator (/root/reference/msmbuilder/cluster/__init__.py:67-69);
scikit-learn computes in the type of X, and the reference pipeline hands it the float64 output of tICA.transform
(decomposition/tica.py:329-352).  Rounds 1-5 narrowed such input to fp32; here the float64 kernels
(kmeans_label_f64_kernel on the fp64 matrix pipe, the typed step / seeding kernels) are held against

* brute-force float64 arithmetic (labels equal off exact ties, inertia rtol 1e-12),
* scikit-learn 1.7.2's own results captured in tests/golden/mbkm_f64_golden.npz (inputs regenerated from seeds),
* scikit-learn live on the GPU box.

Stated tolerance for float64 rows: equal ``n_steps_``, centres rtol 1e-9, inertia rtol 1e-9, labels equal off exact ties."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from seeds import mbkm_f64_data  # noqa: E402


def _brute(X, C):
    d = ((X[:, None, :] - C[None]) ** 2).sum(-1)
    return d.argmin(1), d


@pytest.mark.parametrize("n,f,k", [(1, 3, 1), (1000, 8, 6), (5000, 512, 1000), (3001, 130, 257), (777, 31, 129), (40000, 10, 1000),
                                   (129, 1, 2), (70000, 17, 300)])
def test_label_kernel_f64(gpu, n, f, k):
    """Every shape goes through kmeans_label_f64_kernel: one row, feature counts that are not multiples of the instruction's
    K-step (4) or the LDS step (8), several K-steps, centre counts off the 128-tile, small batches (centres split over
    workgroups, candidates merged) and large ones."""
    import torch
    from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
    rs = np.random.RandomState(n + k)
    C = rs.randn(k, f) * 3
    X = C[rs.randint(0, k, n)] + rs.randn(n, f)
    assert X.dtype == np.float64
    for rows in (X, torch.from_numpy(X).cuda()):
        lab, inertia = label_inertia(rows, C)
        lab = lab.cpu().numpy() if hasattr(lab, "cpu") else lab
        assert lab.dtype == np.int32 and lab.shape == (n,)
        if n * k <= 4_000_000:
            ref, d = _brute(X, C)
            bad = np.nonzero(lab != ref)[0]
            for i in bad:   # a different label only between distances that agree to float64 rounding
                assert abs(d[i, lab[i]] - d[i, ref[i]]) <= 1e-12 * max(d[i, ref[i]], 1e-300), (i, d[i, lab[i]], d[i, ref[i]])
            assert len(bad) <= 1
        ref_inertia = ((X - C[lab]) ** 2).sum()
        np.testing.assert_allclose(inertia, ref_inertia, rtol=1e-12)


def test_label_f64_resolves_what_fp32_cannot(gpu):
    """Centres 1e-5 apart around rows of magnitude 1: squared distances differ by ~1e-10 of ||c||^2, below fp32 resolution
    (every distance ties in the GEMM form) and well inside float64's -- the property the fp32 down-cast of rounds 1-5
    silently lost.  (scikit-learn's float64 GEMM form resolves exactly this much and no more.)"""
    from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
    rs = np.random.RandomState(0)
    base = rs.randn(1, 12)
    C = base + 1e-5 * rs.randn(64, 12)
    pick = rs.randint(0, 64, 500)
    X = C[pick] + 1e-7 * rs.randn(500, 12)
    lab, _ = label_inertia(X, C)
    ref, _ = _brute(X - base, C - base)    # (shifted: the brute-force distances themselves need the headroom)
    assert (lab == ref).mean() > 0.99 and (lab == pick).mean() > 0.99
    lab32, _ = label_inertia(X.astype(np.float32), C.astype(np.float32))
    assert (lab32 == pick).mean() < 0.5


def test_label_f64_ties_lowest_index_and_nan(gpu):
    from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
    X = np.zeros((300, 16))
    C = np.ones((200, 16))                      # all centres identical -> label 0 everywhere
    lab, _ = label_inertia(X, C)
    assert np.all(lab == 0)
    C[150] = 0                                   # unique best, in the second centre tile
    lab, inertia = label_inertia(X, C)
    assert np.all(lab == 150) and inertia == 0.0
    X[7] = np.nan                                # an all-NaN row: scikit-learn's argmin returns 0
    lab, _ = label_inertia(X, C)
    assert lab[7] == 0 and lab[8] == 150


def test_dtype_rule_is_scikit_learns(gpu):
    """float32 rows -> float32 centres; float64, integer and list input -> float64 centres (sklearn validate_data)."""
    sk = pytest.importorskip("sklearn.cluster")
    from msmbuilder_amd import MiniBatchKMeans
    rs = np.random.RandomState(1)
    base = (rs.randn(4, 6) * 20).round()
    Xi = (base[rs.randint(0, 4, 3000)] + rs.randint(-3, 4, (3000, 6))).astype(np.int64)
    for X, want in ((Xi.astype(np.float32), np.float32), (Xi.astype(np.float64), np.float64), (Xi, np.float64)):
        kw = dict(n_clusters=4, init=Xi[:4].astype(np.float64), n_init=1, batch_size=256, max_iter=2, random_state=0)
        mine = MiniBatchKMeans(**kw).fit([X])
        ref = sk.MiniBatchKMeans(**kw).fit(X)
        assert mine.cluster_centers_.dtype == want == ref.cluster_centers_.dtype
        assert mine._counts.dtype == want
        assert mine.n_steps_ == ref.n_steps_
        np.testing.assert_allclose(mine.cluster_centers_, ref.cluster_centers_, rtol=1e-9 if want == np.float64 else 1e-4)
        assert mine.predict([X[:50]])[0].dtype == np.int32


def test_minibatch_f64_golden_sklearn(gpu, golden_dir):
    """scikit-learn 1.7.2's float64 results, captured (no live scikit-learn behind this test): explicit init (a), k-means++
    init (b) on 20,000 x 16; host rows and device rows."""
    import torch
    from msmbuilder_amd import MiniBatchKMeans
    g = np.load(os.path.join(golden_dir, "mbkm_f64_golden.npz"))
    X, init = mbkm_f64_data("small")
    for rows in ([X[:7000], X[7000:]], [torch.from_numpy(X).cuda()]):
        for tag, kw in (("a_", dict(init=init)), ("b_", {})):
            m = MiniBatchKMeans(n_clusters=25, n_init=1, batch_size=512, max_iter=3, random_state=5, **kw).fit(rows)
            assert m.cluster_centers_.dtype == np.float64
            assert m.n_steps_ == int(g[tag + "n_steps"])
            np.testing.assert_allclose(m.cluster_centers_, g[tag + "centers"], rtol=1e-9, atol=1e-12)
            np.testing.assert_array_equal(m._counts, g[tag + "counts"])
            np.testing.assert_allclose(m.inertia_, float(g[tag + "inertia"]), rtol=1e-9)
            labels = np.concatenate([l.cpu().numpy() if hasattr(l, "cpu") else l for l in m.labels_])
            assert labels.dtype == np.int32
            np.testing.assert_array_equal(labels, g[tag + "labels"])


def test_minibatch_f64_golden_k1000_projection(gpu, golden_dir):
    """VERDICT r5 #1 'done' clause: K = 1000 on a 100,000 x 10 float64 projection (k-means++ seeding included), against
    scikit-learn's captured result."""
    import torch
    from msmbuilder_amd import MiniBatchKMeans
    from msmbuilder_amd.cluster.minibatchkmeans import kmeans_plusplus
    g = np.load(os.path.join(golden_dir, "mbkm_f64_golden.npz"))
    Y, _ = mbkm_f64_data("proj")
    seeds = kmeans_plusplus(torch.from_numpy(Y[:3072]).cuda(), 1000, np.random.RandomState(7))
    assert seeds.dtype == np.float64
    np.testing.assert_array_equal(seeds, Y[:3072][g["c_kpp_ids"]])
    m = MiniBatchKMeans(n_clusters=1000, n_init=1, batch_size=1024, max_iter=2, random_state=3).fit([torch.from_numpy(Y).cuda()])
    assert m.n_steps_ == int(g["c_n_steps"])
    np.testing.assert_allclose(m.cluster_centers_, g["c_centers"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(m._counts, g["c_counts"])
    np.testing.assert_allclose(m.inertia_, float(g["c_inertia"]), rtol=1e-9)
    np.testing.assert_array_equal(m.labels_[0].cpu().numpy(), g["c_labels"])


def test_minibatch_f64_vs_sklearn_live(gpu):
    sk = pytest.importorskip("sklearn.cluster")
    import torch
    from msmbuilder_amd import MiniBatchKMeans
    rs = np.random.RandomState(3)
    cent = rs.randn(20, 32) * 5
    X = cent[rs.randint(0, 20, 20000)] + rs.randn(20000, 32)
    init = X[rs.choice(20000, 20, replace=False)].copy()
    for kw in (dict(n_clusters=20, init=init, n_init=1, batch_size=512, max_iter=3, random_state=5),
               dict(n_clusters=20, n_init=1, random_state=1),                                   # k-means++, default batch
               dict(n_clusters=300, n_init=1, batch_size=4096, max_iter=2, random_state=2),     # general update kernel, splits
               dict(n_clusters=12, init="random", n_init=3, batch_size=700, max_iter=2, random_state=4, tol=1e-4)):
        ref = sk.MiniBatchKMeans(**kw).fit(X)
        for rows in (X, torch.from_numpy(X).cuda()):
            mine = MiniBatchKMeans(**kw).fit([rows])
            assert mine.n_steps_ == ref.n_steps_          # same minibatch stream, same early-stopping decisions
            np.testing.assert_allclose(mine.cluster_centers_, ref.cluster_centers_, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(mine.inertia_, ref.inertia_, rtol=1e-9)
            lab = mine.labels_[0].cpu().numpy() if hasattr(mine.labels_[0], "cpu") else mine.labels_[0]
            np.testing.assert_array_equal(lab, ref.labels_)
        np.testing.assert_array_equal(mine.predict([X[:100]])[0], lab[:100])
        np.testing.assert_allclose(mine.score(X[:1000]), ref.score(X[:1000]), rtol=1e-9)


@pytest.mark.parametrize("n,F,k", [(500, 8, 10), (3072, 64, 50), (1000, 3, 25), (3072, 512, 200), (7, 2, 7), (2500, 33, 40),
                                   (20000, 10, 200), (1500, 4000, 30)])
def test_device_kmeans_plusplus_f64_draws_scikit_learns_seeds(gpu, n, F, k):
    """msm_kmeans_plusplus_f64 against sklearn.cluster.kmeans_plusplus live: float64 potentials on both sides, so the picks
    agree also on samples where the float32 path's do not (20,000 rows), and the generator ends in the same state."""
    sk = pytest.importorskip("sklearn.cluster")
    import torch
    from msmbuilder_amd.cluster.minibatchkmeans import kmeans_plusplus
    rs = np.random.RandomState(n + k)
    X = rs.randn(n, F) * rs.uniform(0.5, 3, F) + rs.randn(F)
    g_ref = np.random.RandomState(7)
    ref, _ = sk.kmeans_plusplus(X, k, random_state=g_ref)
    tail = g_ref.randint(0, 1 << 30, 5)
    for rows in (X, torch.from_numpy(X).cuda()):
        g_mine = np.random.RandomState(7)
        mine = kmeans_plusplus(rows, k, g_mine)
        assert mine.dtype == np.float64
        np.testing.assert_array_equal(mine, ref)
        np.testing.assert_array_equal(g_mine.randint(0, 1 << 30, 5), tail)


@pytest.mark.parametrize("K,B,mni", [(200, 128, 10), (300, 1024, None), (40, 4096, 3)])
def test_minibatch_f64_queued_runs_equal_step_by_step(gpu, monkeypatch, K, B, mni):
    """The queued runs (msm_mbk_run_begin/_end on a float64 handle) against the step-by-step path and scikit-learn."""
    sk = pytest.importorskip("sklearn.cluster")
    import torch
    from msmbuilder_amd import MiniBatchKMeans
    rs = np.random.RandomState(K + B)
    cent = rs.randn(max(K // 4, 2), 24) * 4
    X = cent[rs.randint(0, len(cent), 60000)] + rs.randn(60000, 24)
    Xd = torch.from_numpy(X).cuda()
    out = {}
    for runs in ("1", "0"):
        monkeypatch.setenv("MSMBUILDER_AMD_MBK_RUNS", runs)
        gen = np.random.RandomState(11)
        m = MiniBatchKMeans(n_clusters=K, batch_size=B, max_iter=4, n_init=1, max_no_improvement=mni, random_state=gen).fit([Xd])
        out[runs] = (m.cluster_centers_.copy(), m.n_steps_, m.inertia_, gen.randint(0, 1 << 30, 4))
    a, b = out["1"], out["0"]
    assert a[1] == b[1]
    np.testing.assert_array_equal(a[0], b[0])
    assert a[2] == b[2]
    np.testing.assert_array_equal(a[3], b[3])
    gen = np.random.RandomState(11)
    ref = sk.MiniBatchKMeans(n_clusters=K, batch_size=B, max_iter=4, n_init=1, max_no_improvement=mni, random_state=gen).fit(X)
    assert ref.n_steps_ == a[1]
    np.testing.assert_array_equal(gen.randint(0, 1 << 30, 4), a[3])
    np.testing.assert_allclose(a[0], ref.cluster_centers_, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a[2], ref.inertia_, rtol=1e-9)


def test_partial_fit_f64(gpu):
    sk = pytest.importorskip("sklearn.cluster")
    from msmbuilder_amd.cluster.minibatchkmeans import _MiniBatchKMeans
    rs = np.random.RandomState(9)
    cent = rs.randn(5, 7) * 6
    init = cent + 0.1
    mine = _MiniBatchKMeans(n_clusters=5, init=init, n_init=1, random_state=0)
    ref = sk.MiniBatchKMeans(n_clusters=5, init=init, n_init=1, random_state=0)
    for i in range(4):
        X = cent[rs.randint(0, 5, 400)] + rs.randn(400, 7)
        mine.partial_fit(X if i != 2 else X.astype(np.float32))   # a float32 batch is converted to the centres' type
        ref.partial_fit(X if i != 2 else X.astype(np.float32).astype(np.float64))
        assert mine.cluster_centers_.dtype == np.float64
        np.testing.assert_allclose(mine.cluster_centers_, ref.cluster_centers_, rtol=1e-9, atol=1e-12)
        np.testing.assert_array_equal(mine.labels_, ref.labels_)


def test_stateless_step_f64_c_abi(gpu):
    """msm_mbk_step_f64 / msm_kmeans_label_f64 straight through ctypes (no estimator): one scikit-learn mini-batch step."""
    import ctypes as C
    from msmbuilder_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(4)
    K, m, n, B = 37, 9, 5000, 1500
    X = rs.randn(n, m) * 2
    cen = X[rs.choice(n, K, replace=False)].copy()
    cnt = rs.randint(1, 50, K).astype(np.float64)
    idx = np.ascontiguousarray(rs.randint(0, n, B), dtype=np.int64)
    c0, w0 = cen.copy(), cnt.copy()
    inertia = C.c_double(0.0)
    sums, cnts = np.empty((K, m)), np.empty(K)
    _lib.check(L.msm_mbk_step_f64(X.ctypes.data, n, m, idx.ctypes.data, B, cen.ctypes.data, cnt.ctypes.data, K, C.byref(inertia),
                                  sums.ctypes.data, cnts.ctypes.data, 1, 0))
    Xb = X[idx]
    lab, d = _brute(Xb, c0)
    np.testing.assert_allclose(inertia.value, d[np.arange(B), lab].sum(), rtol=1e-12)
    want = c0.copy()
    for j in range(K):
        mem = Xb[lab == j]
        if len(mem):
            acc = c0[j] * w0[j]
            for x in mem:      # batch order, like _k_means_minibatch.pyx
                acc = acc + x
            want[j] = acc * (1.0 / (w0[j] + len(mem)))
    np.testing.assert_array_equal(cnt, w0 + np.bincount(lab, minlength=K))
    np.testing.assert_array_equal(cen, want)          # the same float64 operations in the same order: bit for bit
    np.testing.assert_allclose(sums, np.stack([Xb[lab == j].sum(0) for j in range(K)]), rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(cnts, np.bincount(lab, minlength=K))
