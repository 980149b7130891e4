#!/usr/bin/env python
"""tests/golden/make_golden.py -- generate the committed golden vectors.

Runs ONLY in the dev container (it needs /root/reference).  It imports the
reference's own python files *by file path* -- nothing of the reference is
copied into this repository -- and stores inputs + the reference's outputs as
small ``.npz`` fixtures next to this script:

* tica_golden.npz        msmbuilder/decomposition/tica.py  (tICA)
* kcenters_golden.npz    msmbuilder/cluster/kcenters.py + cluster/base.py over
                         the reference's libdistance headers (oracle/_ref)
* libdistance_golden.npz msmbuilder/libdistance/src/*.hpp compiled (oracle/_ref)
* transition_golden.npz  msmbuilder/msm/core.py (_transition_counts)
* mbkm_f64_golden.npz    the same on FLOAT64 rows (round 6; inputs regenerated from seeds)
* mbkm_golden.npz        scikit-learn MiniBatchKMeans (the third-party
                         arithmetic behind msmbuilder.cluster.MiniBatchKMeans,
                         cluster/__init__.py:67-69; unpinned upstream)

Loader recipe: synthetic ``msmbuilder`` packages in
sys.modules, a stub ``mdtraj`` exposing an empty ``Trajectory`` class (the
reference imports it for isinstance checks only), and a wrapper around
``scipy.linalg.eigh`` translating the removed ``eigvals=(lo, hi)`` kwarg
(tica.py:188-189) to ``subset_by_index``.

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import scipy.linalg

REF = "/root/reference/msmbuilder"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Return (tICA, KCenters) classes of the real reference."""
    # --- stubs -----------------------------------------------------------
    md = types.ModuleType("mdtraj")

    class Trajectory(object):
        pass
    md.Trajectory = Trajectory
    sys.modules["mdtraj"] = md

    _eigh = scipy.linalg.eigh

    def eigh_compat(a, b=None, eigvals=None, **kw):
        if eigvals is not None:
            kw["subset_by_index"] = list(eigvals)
        return _eigh(a, b=b, **kw)
    scipy.linalg.eigh = eigh_compat

    pkg = types.ModuleType("msmbuilder")
    pkg.__path__ = [REF]
    sys.modules["msmbuilder"] = pkg
    _load("msmbuilder.base", os.path.join(REF, "base.py"), "msmbuilder")

    utils = types.ModuleType("msmbuilder.utils")
    utils.__path__ = [os.path.join(REF, "utils")]
    sys.modules["msmbuilder.utils"] = utils
    val = _load("msmbuilder.utils.validation", os.path.join(REF, "utils", "validation.py"),
                "msmbuilder.utils")
    utils.check_iter_of_sequences = val.check_iter_of_sequences
    utils.array2d = val.array2d

    dec = types.ModuleType("msmbuilder.decomposition")
    dec.__path__ = [os.path.join(REF, "decomposition")]
    sys.modules["msmbuilder.decomposition"] = dec
    tica_mod = _load("msmbuilder.decomposition.tica", os.path.join(REF, "decomposition", "tica.py"),
                     "msmbuilder.decomposition")

    # libdistance: fake module backed by the reference headers compiled in oracle/_ref
    from oracle.libdistance_oracle import Ref
    ref = Ref()
    fake = types.ModuleType("msmbuilder.libdistance")
    fake.assign_nearest = ref.assign_nearest
    fake.dist = ref.dist
    fake.cdist = ref.cdist
    sys.modules["msmbuilder.libdistance"] = fake
    pkg.libdistance = fake

    clu = types.ModuleType("msmbuilder.cluster")
    clu.__path__ = [os.path.join(REF, "cluster")]
    sys.modules["msmbuilder.cluster"] = clu
    base = _load("msmbuilder.cluster.base", os.path.join(REF, "cluster", "base.py"),
                 "msmbuilder.cluster")
    clu.MultiSequenceClusterMixin = base.MultiSequenceClusterMixin
    kc = _load("msmbuilder.cluster.kcenters", os.path.join(REF, "cluster", "kcenters.py"),
               "msmbuilder.cluster")
    return tica_mod.tICA, kc.KCenters, ref


def ar1_sequences(seed, n_seq, n_frames, n_features, n_slow=4, tau=10.0):
    """Small slow-mode synthetic (SURVEY.md section 8(d) recipe, scaled down)."""
    rs = np.random.RandomState(seed)
    M = rs.randn(n_slow, n_features)
    b = rs.uniform(-1, 1, size=n_features)
    a = np.exp(-1.0 / (tau * (1 + np.arange(n_slow))))
    out = []
    for _ in range(n_seq):
        z = np.zeros((n_frames, n_slow))
        z[0] = rs.randn(n_slow)
        eps = rs.randn(n_frames, n_slow)
        for t in range(1, n_frames):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * eps[t]
        X = z.dot(M) + 0.5 * rs.randn(n_frames, n_features) + b
        out.append(X.astype(np.float32))
    return out


def main():
    tICA, KCenters, ref = load_reference()
    warnings.simplefilter("ignore")

    # ------------------------------------------------------------------ tICA
    g = {}
    # case A: SURVEY.md known-answer (4th sequence is skipped: N=2 <= tau)
    rs = np.random.RandomState(0)
    seqsA = [rs.randn(1000, 6).astype(np.float32) for _ in range(3)] + [rs.randn(2, 6).astype(np.float32)]
    for tag, shr in (("A0", 0), ("An", None)):
        m = tICA(n_components=3, lag_time=2, shrinkage=shr).fit(seqsA)
        g[tag + "_eigenvalues"] = m.eigenvalues_
        g[tag + "_eigenvectors"] = m.eigenvectors_
        g[tag + "_timescales"] = m.timescales_
        g[tag + "_means"] = m.means_
        g[tag + "_offset_correlation"] = m.offset_correlation_
        g[tag + "_covariance"] = m.covariance_
        g[tag + "_shrinkage_"] = np.float64(m.shrinkage_)
        g[tag + "_score_"] = np.float64(m.score_)
        g[tag + "_n_obs_seq"] = np.array([m.n_observations_, m.n_sequences_])
        g[tag + "_transform0"] = m.transform(seqsA[:1])[0]
        g[tag + "_score_test"] = np.float64(m.score(seqsA[1:3]))
    # case B: slow-mode AR(1) data, ragged lengths, lag 7, all mapping variants
    seqsB = ar1_sequences(11, 5, 400, 12)
    seqsB[2] = seqsB[2][:123]
    seqsB[4] = seqsB[4][:7]  # == lag -> skipped
    for i, s in enumerate(seqsB):
        g["B_seq%d" % i] = s
    for tag, kw in (("B", {}), ("Bk", dict(kinetic_mapping=True)), ("Bc", dict(commute_mapping=True))):
        m = tICA(n_components=4, lag_time=7, **kw).fit(seqsB)
        g[tag + "_eigenvalues"] = m.eigenvalues_
        g[tag + "_eigenvectors"] = m.eigenvectors_
        g[tag + "_timescales"] = m.timescales_
        g[tag + "_means"] = m.means_
        g[tag + "_shrinkage_"] = np.float64(m.shrinkage_)
        g[tag + "_transform1"] = m.transform(seqsB[1:2])[0]
        if tag == "B":
            g["B_C"] = m._outer_0_to_T_lagged
            g["B_S0"] = m._outer_0_to_TminusTau
            g["B_Stau"] = m._outer_offset_to_T
            g["B_s0"] = m._sum_0_to_TminusTau
            g["B_stau"] = m._sum_tau_to_T
            g["B_n_obs_seq"] = np.array([m.n_observations_, m.n_sequences_])
            g["B_summarize"] = np.array(m.summarize())
    np.savez_compressed(os.path.join(HERE, "tica_golden.npz"), **g)
    print("tica_golden.npz:", len(g), "arrays; A0 eigenvalues", g["A0_eigenvalues"])

    # -------------------------------------------------------------- KCenters
    g = {}
    rs = np.random.RandomState(1)
    seqs = [rs.randn(23, 2).astype(np.float32), rs.randn(10, 2).astype(np.float32)]
    m = KCenters(n_clusters=3, random_state=0).fit(seqs)
    g["K1_ids"] = np.array(m.cluster_ids_)
    g["K1_inertia"] = np.float64(m.inertia_)
    g["K1_labels"] = np.concatenate(m.labels_)
    g["K1_distances"] = np.concatenate(m.distances_)
    g["K1_centers"] = m.cluster_centers_
    g["K1_predict"] = np.concatenate(m.predict(seqs))
    # a larger case per metric and dtype
    rs = np.random.RandomState(5)
    Xk = [rs.randn(n, 7) for n in (301, 57, 160)]
    for i, s in enumerate(Xk):
        g["K2_seq%d" % i] = s
    for metric in ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis"):
        for dt, dn in ((np.float32, "f32"), (np.float64, "f64")):
            seqs = [s.astype(dt) for s in Xk]
            m = KCenters(n_clusters=12, metric=metric, random_state=3).fit(seqs)
            p = "K2_%s_%s_" % (metric, dn)
            g[p + "ids"] = np.array(m.cluster_ids_)
            g[p + "inertia"] = np.float64(m.inertia_)
            g[p + "labels"] = np.concatenate(m.labels_)
            g[p + "distances"] = np.concatenate(m.distances_)
            g[p + "predict"] = np.concatenate(m.predict(seqs))
            g[p + "summarize"] = np.array(m.summarize())
    np.savez_compressed(os.path.join(HERE, "kcenters_golden.npz"), **g)
    print("kcenters_golden.npz:", len(g), "arrays; K1 ids", g["K1_ids"], "inertia", g["K1_inertia"])

    # ----------------------------------------------------------- libdistance
    g = {}
    rs = np.random.RandomState(7)
    X = rs.randn(64, 5)
    Y = rs.randn(9, 5)
    X[10] = Y[3]            # exact hit
    X[11] = 0.0             # zero row (braycurtis/jaccard/canberra edge)
    Xr, Yr = np.round(X), np.round(Y)  # integer-valued: ties, hamming/jaccard structure
    idx = rs.randint(0, 64, size=13).astype(np.int64)
    g.update(X=X, Y=Y, Xr=Xr, Yr=Yr, idx=idx)
    for metric in ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra",
                   "braycurtis", "hamming", "jaccard"):
        for dn, dt in (("f32", np.float32), ("f64", np.float64)):
            for tag, (A, B) in (("g", (X, Y)), ("r", (Xr, Yr))):
                A, B = A.astype(dt), B.astype(dt)
                p = "%s_%s_%s_" % (metric, dn, tag)
                with np.errstate(all="ignore"):
                    g[p + "cdist"] = ref.cdist(A, B, metric)
                    lab, inertia = ref.assign_nearest(A, B, metric)
                    g[p + "assign"] = lab
                    g[p + "inertia"] = np.float64(inertia)
                    lab, inertia = ref.assign_nearest(A, B, metric, idx)
                    g[p + "assign_idx"] = lab
                    g[p + "inertia_idx"] = np.float64(inertia)
                    g[p + "dist"] = ref.dist(A, B[2], metric)
                    g[p + "dist_idx"] = ref.dist(A, B[2], metric, idx)
    np.savez_compressed(os.path.join(HERE, "libdistance_golden.npz"), **g)
    print("libdistance_golden.npz:", len(g), "arrays")

    # ------------------------------------------------------- MiniBatchKMeans
    import sklearn
    from sklearn.cluster import MiniBatchKMeans
    g = {"sklearn_version": np.array(sklearn.__version__)}
    rs = np.random.RandomState(42)
    cent = rs.randn(6, 8) * 4
    Xm = (cent[rs.randint(0, 6, size=3000)] + rs.randn(3000, 8)).astype(np.float32)
    init = Xm[rs.choice(3000, 6, replace=False)].copy()
    g["X"] = Xm
    g["init"] = init
    mb = MiniBatchKMeans(n_clusters=6, init=init, n_init=1, batch_size=256, max_iter=5,
                         random_state=0, max_no_improvement=None, reassignment_ratio=0.0,
                         tol=0.0).fit(Xm)
    g["centers"] = mb.cluster_centers_
    g["labels"] = mb.labels_
    g["inertia"] = np.float64(mb.inertia_)
    g["n_steps"] = np.int64(mb.n_steps_)
    g["counts"] = mb._counts
    np.savez_compressed(os.path.join(HERE, "mbkm_golden.npz"), **g)
    print("mbkm_golden.npz: sklearn", sklearn.__version__, "inertia", g["inertia"], "steps", g["n_steps"])


def kpp_golden():
    """Round 5: k-means++ seeds of a MID-SIZE sample captured from scikit-learn itself, so that "the device
    seeding is scikit-learn's, row for row" is a test with no live scikit-learn behind it: 6,000 x 10 float32, K = 200,
    RandomState(7).  Appended to mbkm_golden.npz (its other arrays are left as they are).  Why 6,000: scikit-learn's current
    potential is a float32 sum whose rounding depends on the summation order of its build, and every inverse-CDF draw is
    scaled by it -- against the float64 restatement (oracle/kpp_oracle.py, = the device kernel) its picks agreed on every
    sample tried up to 6,000 rows and departed on one of three 8,000-row samples and on most samples from 20,000 rows
    (scan below, printed)."""
    import sklearn
    from sklearn.cluster import kmeans_plusplus as sk_kpp
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.kpp_oracle import kmeans_plusplus_f64

    def sample(n, seed):
        rs = np.random.RandomState(seed)
        return (rs.randn(n, 10) * np.linspace(3, 0.3, 10)).astype(np.float32)
    path = os.path.join(HERE, "mbkm_golden.npz")
    g = dict(np.load(path, allow_pickle=False))
    X = sample(6000, 5)
    centers, ids = sk_kpp(X, 200, random_state=np.random.RandomState(7))
    g["kpp_n"], g["kpp_data_seed"], g["kpp_k"], g["kpp_stream_seed"] = np.int64(6000), np.int64(5), np.int64(200), np.int64(7)
    g["kpp_ids"] = np.asarray(ids, dtype=np.int64)
    g["kpp_centers"] = np.asarray(centers, dtype=np.float32)
    g["kpp_sklearn_version"] = np.array(sklearn.__version__)
    scan = []
    for seed in (5, 6, 7):
        for n in (3072, 6000, 8000, 15000, 20000, 50000):
            Xs = sample(n, seed)
            _, a = sk_kpp(Xs, 200, random_state=np.random.RandomState(7))
            _, b = kmeans_plusplus_f64(Xs, 200, np.random.RandomState(7))
            same = np.asarray(a) == np.asarray(b)
            scan.append((seed, n, int(same.sum())))
            print("kpp scan: data seed %d, %6d rows: %3d of 200 picks equal scikit-learn's" % scan[-1])
    g["kpp_scan_seed_rows_equal"] = np.asarray(scan, dtype=np.int64)
    np.savez_compressed(path, **g)
    print("mbkm_golden.npz: + k-means++ seeds of 6000 x 10, K = 200 from scikit-learn", sklearn.__version__)


from seeds import mbkm_f64_data  # noqa: E402  (tests/golden/seeds.py: shared with the tests)


def mbkm_f64_golden():
    """Round 6: scikit-learn MiniBatchKMeans on FLOAT64 rows -- the type the reference pipeline feeds it
    (cluster/__init__.py:67-69 <- tica.py:329-352) -- captured so that the float64 GPU path is pinned without a live
    scikit-learn: explicit init and k-means++ init on 20,000 x 16, and K = 1000 with k-means++ on a 100,000 x 10 projection.
    Stored: centres, counts, inertia, n_steps, labels (and the k-means++ seeds' row ids); the inputs come from
    `mbkm_f64_data` seeds."""
    import sklearn
    from sklearn.cluster import MiniBatchKMeans, kmeans_plusplus as sk_kpp
    g = {"sklearn_version": np.array(sklearn.__version__)}
    X, init = mbkm_f64_data("small")
    assert X.dtype == np.float64
    a = MiniBatchKMeans(n_clusters=25, init=init, n_init=1, batch_size=512, max_iter=3, random_state=5).fit(X)
    b = MiniBatchKMeans(n_clusters=25, n_init=1, batch_size=512, max_iter=3, random_state=5).fit(X)
    for tag, m in (("a_", a), ("b_", b)):
        assert m.cluster_centers_.dtype == np.float64
        g[tag + "centers"], g[tag + "counts"] = m.cluster_centers_, m._counts
        g[tag + "inertia"], g[tag + "n_steps"] = np.float64(m.inertia_), np.int64(m.n_steps_)
        g[tag + "labels"] = m.labels_.astype(np.int32)
    cs, ids = sk_kpp(X, 25, random_state=np.random.RandomState(7))
    g["kpp_ids"] = np.asarray(ids, dtype=np.int64)
    Y, _ = mbkm_f64_data("proj")
    c = MiniBatchKMeans(n_clusters=1000, n_init=1, batch_size=1024, max_iter=2, random_state=3).fit(Y)
    g["c_centers"], g["c_counts"] = c.cluster_centers_, c._counts
    g["c_inertia"], g["c_n_steps"] = np.float64(c.inertia_), np.int64(c.n_steps_)
    g["c_labels"] = c.labels_.astype(np.int32)
    _, ids = sk_kpp(Y[:3072], 1000, random_state=np.random.RandomState(7))
    g["c_kpp_ids"] = np.asarray(ids, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "mbkm_f64_golden.npz"), **g)
    print("mbkm_f64_golden.npz: sklearn", sklearn.__version__, "steps", g["a_n_steps"], g["b_n_steps"], g["c_n_steps"],
          "inertia", g["a_inertia"], g["b_inertia"], g["c_inertia"])


def transition_golden():
    """msmbuilder.msm._transition_counts itself (msm/core.py:487-596), loaded by file path with a stub
    for the compiled _ratematrix extension and the numpy aliases (np.int / np.float) it still uses."""
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    if "msmbuilder" not in sys.modules:
        pkg = types.ModuleType("msmbuilder")
        pkg.__path__ = [REF]
        sys.modules["msmbuilder"] = pkg
    if "msmbuilder.utils" not in sys.modules or not hasattr(sys.modules["msmbuilder.utils"], "list_of_1d"):
        utils = sys.modules.get("msmbuilder.utils") or types.ModuleType("msmbuilder.utils")
        utils.__path__ = [os.path.join(REF, "utils")]
        sys.modules["msmbuilder.utils"] = utils
        if "mdtraj" not in sys.modules:
            md = types.ModuleType("mdtraj")
            md.Trajectory = type("Trajectory", (object,), {})
            sys.modules["mdtraj"] = md
        val = sys.modules.get("msmbuilder.utils.validation") or _load(
            "msmbuilder.utils.validation", os.path.join(REF, "utils", "validation.py"), "msmbuilder.utils")
        utils.list_of_1d = val.list_of_1d
    msm = types.ModuleType("msmbuilder.msm")
    msm.__path__ = [os.path.join(REF, "msm")]
    sys.modules["msmbuilder.msm"] = msm
    sys.modules["msmbuilder.msm._ratematrix"] = types.ModuleType("msmbuilder.msm._ratematrix")
    core = _load("msmbuilder.msm.core", os.path.join(REF, "msm", "core.py"), "msmbuilder.msm")
    tc = core._transition_counts

    g = {}
    rs = np.random.RandomState(21)

    def metastable(n, k, stay=0.97):
        y = np.empty(n, dtype=np.int64)
        y[0] = rs.randint(k)
        for t in range(1, n):
            y[t] = y[t - 1] if rs.rand() < stay else rs.randint(k)
        return y

    cases = {
        "ident": ([metastable(5000, 12), metastable(37, 12), metastable(1, 12), metastable(9000, 12)], 1, True),
        "lag7": ([metastable(5000, 12), metastable(7, 12), metastable(8, 12), metastable(4100, 12)], 7, True),
        "gaps": ([metastable(3000, 9) * 5 + 100, metastable(2000, 9) * 5 + 100], 3, True),          # non-identity mapping
        "nosw": ([metastable(5000, 6), metastable(1234, 6)], 4, False),
        "neg": ([metastable(2500, 7) - 3], 2, True),
    }
    for name, (seqs, lag, sw) in cases.items():
        c, m = tc(seqs, lag_time=lag, sliding_window=sw)
        for i, y in enumerate(seqs):
            g["%s_seq%d" % (name, i)] = y
        g[name + "_nseq"] = np.int64(len(seqs))
        g[name + "_lag"] = np.int64(lag)
        g[name + "_sw"] = np.bool_(sw)
        g[name + "_counts"] = c
        g[name + "_keys"] = np.array(list(m.keys()), dtype=np.int64)
        g[name + "_vals"] = np.array(list(m.values()), dtype=np.int64)
    # float labels with NaN, and strings
    yf = metastable(4000, 5).astype(float)
    yf[rs.randint(0, 4000, 60)] = np.nan
    c, m = tc([yf, yf[:100]], lag_time=2)
    g["nan_seq0"], g["nan_counts"] = yf, c
    g["nan_keys"] = np.array(list(m.keys()), dtype=float)
    ys = np.array(["s%02d" % v for v in metastable(500, 4)])
    c, m = tc([ys], lag_time=1)
    g["str_seq0"], g["str_counts"], g["str_keys"] = ys, c, np.array(list(m.keys()))
    np.savez_compressed(os.path.join(HERE, "transition_golden.npz"), **g)
    print("transition_golden.npz:", len(g), "arrays; ident counts sum", g["ident_counts"].sum())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "transition":
        transition_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "kpp":
        kpp_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "mbkm_f64":
        mbkm_f64_golden()
    else:
        main()
        transition_golden()
        kpp_golden()
        mbkm_f64_golden()
