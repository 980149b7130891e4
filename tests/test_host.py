"""CPU tier: the C ABI (library loads, exports every declared symbol, fails loudly without a
GPU), host-side logic (validation, finalisation math, sharding) and the world_size-2 gloo
path of the multi-GPU reductions."""
import ctypes as C
import os
import re
import subprocess
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for hdr in ("msmhip.h", "msmhip_libdistance.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b((?:msm_|assign_nearest_|dist_|cdist_)\w+)\s*\(", txt)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    from msmbuilder_amd import _lib
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(L, s), "libmsmhip.so does not export %s" % s
    assert b"gfx950" in L.msm_version()


def test_built_for_gfx950_only(tmp_path):
    # llvm-objdump --offloading drops the unbundled code objects next to its input: give it a link in a scratch dir
    so = str(tmp_path / "libmsmhip.so")
    os.symlink(os.path.join(ROOT, "msmbuilder_amd", "libmsmhip.so"), so)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", so], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("llvm-objdump --offloading unavailable")
    archs = set(re.findall(r"gfx[0-9a-f]+", out.stdout))
    assert archs == {"gfx950"}, archs


def test_no_device_is_a_loud_error_not_a_fallback():
    from msmbuilder_amd import _lib, tICA, KCenters, libdistance
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.NoDeviceError):
        tICA().fit([np.random.randn(50, 3)])
    with pytest.raises(_lib.NoDeviceError):
        KCenters(n_clusters=2).fit([np.random.randn(50, 3)])
    with pytest.raises(_lib.NoDeviceError):
        libdistance.assign_nearest(np.zeros((4, 2)), np.zeros((2, 2)), "euclidean")
    # argument validation still happens before any device work, with the reference's error types
    with pytest.raises(ValueError):
        libdistance.assign_nearest(np.zeros((4, 2)), np.zeros((2, 2)), "rmsd")
    with pytest.raises(TypeError):
        libdistance.cdist(np.zeros((4, 2), np.float32), np.zeros((2, 2)), "euclidean")
    with pytest.raises(ValueError):
        tICA(kinetic_mapping=True, commute_mapping=True)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "msmbuilder_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace("# oracle", ""), f


def test_validation_semantics():
    from msmbuilder_amd.utils import array2d, check_iter_of_sequences
    check_iter_of_sequences([np.zeros((3, 2)), np.zeros((1, 2))])
    with pytest.raises(ValueError, match="list of sequences"):
        check_iter_of_sequences([np.zeros(3)])
    with pytest.raises(ValueError, match="list of sequences"):
        check_iter_of_sequences([np.zeros((3, 2)), [[1, 2]]])
    check_iter_of_sequences(iter([np.zeros((3, 2))] * 10), max_iter=3)
    assert array2d([1.0, 2.0]).shape == (1, 2)
    with pytest.raises(ValueError, match="NaN"):
        array2d(np.array([[1.0, np.nan]]))
    big = np.full((2, 2), 1e308)   # the sum overflows but every element is finite: accepted
    assert array2d(big) is big


def test_moments_match_oracle_formulas():
    from msmbuilder_amd.decomposition import _moments
    from oracle.tica_oracle import TicaOracle, rao_blackwell_ledoit_wolf
    rs = np.random.RandomState(4)
    seqs = [rs.randn(200, 7) + 2 for _ in range(3)]
    for shr in (None, 0.0, 0.3):
        o = TicaOracle(n_components=3, lag_time=5, shrinkage=shr).fit(seqs)
        npairs = _moments.pair_count(o.n_observations_, o.n_sequences_, 5)
        mu = _moments.mean_vector(o.s0, o.stau, npairs)
        np.testing.assert_array_equal(mu, o.means_)
        np.testing.assert_array_equal(_moments.offset_correlation(o.C, mu, npairs), o.offset_correlation_)
        S = _moments.sample_covariance(o.S0 + o.Stau, mu, npairs)
        rho = _moments.rblw_shrinkage(S, o.n_observations_) if shr is None else shr
        np.testing.assert_allclose(_moments.shrink(S, rho), o.covariance_, rtol=1e-14, atol=1e-16)
        sig, r2 = _moments.rao_blackwell_ledoit_wolf(S, o.n_observations_)
        sig_o, r_o = rao_blackwell_ledoit_wolf(S, o.n_observations_)
        assert r2 == r_o and np.array_equal(sig, sig_o)
        vals, vecs = _moments.top_generalized_eigenpairs(o.offset_correlation_, o.covariance_, 3)
        np.testing.assert_allclose(vals, o.eigenvalues_, rtol=1e-12)


def test_shard_sequences_balanced_and_complete():
    from msmbuilder_amd.parallel import shard_sequences
    rs = np.random.RandomState(0)
    seqs = [np.zeros((int(n), 1)) for n in rs.randint(10, 5000, size=57)]
    for world in (1, 2, 4, 8):
        owned = [shard_sequences(seqs, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(57))
        loads = [sum(len(seqs[i]) for i in o) for o in owned]
        assert max(loads) - min(loads) <= 5000


def test_sharded_sum_equals_unsharded_oracle():
    """Single-process simulation of the multi-rank tICA: per-rank partial accumulators summed
    == one pass over everything (lagged pairs never cross a trajectory, reference tica.py:417)."""
    from msmbuilder_amd.parallel import shard_sequences
    from oracle.tica_oracle import TicaOracle
    rs = np.random.RandomState(1)
    seqs = [rs.randn(int(n), 5) for n in rs.randint(3, 400, size=23)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = TicaOracle(lag_time=4).fit(seqs)
        parts = []
        for r in range(4):
            o = TicaOracle(lag_time=4)
            for i in shard_sequences(seqs, r, 4):
                o.partial_fit(seqs[i])
            parts.append(o)
    for name in ("C", "S0", "Stau", "s0", "stau"):
        np.testing.assert_allclose(sum(getattr(p, name) for p in parts), getattr(full, name), rtol=1e-12, atol=1e-10)
    assert sum(p.n_observations_ for p in parts) == full.n_observations_
    assert sum(p.n_sequences_ for p in parts) == full.n_sequences_


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from msmbuilder_amd import parallel
rank, world, local = parallel.init_from_env(backend="gloo")
assert parallel.active() and parallel.world_size() == 2
# 1) packed-accumulator all-reduce (the tICA exchange step) on oracle partials
from oracle.tica_oracle import TicaOracle
rs = np.random.RandomState(7)
seqs = [rs.randn(int(n), 4) for n in rs.randint(20, 300, size=9)]
mine = parallel.shard_sequences(seqs)
o = TicaOracle(lag_time=3)
for i in mine:
    o.partial_fit(seqs[i])
packed = np.concatenate([o.C.ravel(), (o.S0 + o.Stau).ravel(), o.s0, o.stau, [o.n_observations_, o.n_sequences_]])
tot = parallel.allreduce_array(packed)
full = TicaOracle(lag_time=3).fit(seqs)
ref = np.concatenate([full.C.ravel(), (full.S0 + full.Stau).ravel(), full.s0, full.stau,
                      [full.n_observations_, full.n_sequences_]])
assert np.allclose(tot, ref, rtol=1e-12, atol=1e-10), np.abs(tot - ref).max()
# 2) row sharding bookkeeping + distributed row gather (MiniBatchKMeans batches)
X = np.arange(40, dtype=np.float32).reshape(10, 4) + 100 * rank
shard = parallel.RowShard(10)
assert shard.n_total == 20 and shard.offset == 10 * rank
idx = np.array([0, 19, 10, 9, 3])
rows = shard.gather_rows(lambda loc: X[loc], idx, 4)
expect = np.stack([(np.arange(40, dtype=np.float32).reshape(10, 4) + 100 * (i // 10))[i % 10] for i in idx])
assert np.array_equal(rows, expect)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gloo_world2_reductions(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_split_frames_covers_every_usable_row_once():
    from msmbuilder_amd.parallel import split_frames
    lengths, lag = [5, 1000, 37, 12, 4000, 13], 12
    for world in (1, 2, 3, 8):
        seen = {i: np.zeros(n, dtype=int) for i, n in enumerate(lengths)}
        loads = []
        for r in range(world):
            pieces = split_frames(lengths, lag, r, world)
            loads.append(sum(e - b for _, b, e in pieces))
            for i, b, e in pieces:
                assert lengths[i] > lag and 0 <= b < e <= lengths[i]
                seen[i][b:e] += 1
        for i, n in enumerate(lengths):
            assert (seen[i] == (1 if n > lag else 0)).all()
        assert max(loads) - min(loads) <= 1


def test_eigensolve_survives_torchrun_thread_settings():
    """torchrun starts every rank with OMP_NUM_THREADS=1; raising OpenBLAS's thread count afterwards (the solve asks
    for a handful of threads) crashed dsygvx with SIGSEGV.  The limiter must only lower."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from msmbuilder_amd.decomposition import _moments\n"
        "rs = np.random.RandomState(0)\n"
        "A = rs.randn(512, 512); A = A + A.T\n"
        "B = rs.randn(700, 512); B = B.T @ B\n"
        "vals, vecs = _moments.top_generalized_eigenpairs(A, B, 5)\n"
        "assert vals.shape == (5,) and np.all(np.diff(vals) <= 0)\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stderr[-2000:])


def _header_arities():
    """{function name: number of parameters} parsed from include/msmhip.h (plain C declarations)."""
    txt = open(os.path.join(ROOT, "include", "msmhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    out = {}
    for name, params in re.findall(r"\b(msm_\w+)\s*\(([^()]*)\)\s*;", txt):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_ctypes_prototypes_match_the_header():
    """Every prototype the Python host declares must have as many arguments as the C declaration in include/msmhip.h:
    a drifted ctypes signature passes garbage in registers instead of failing."""
    from msmbuilder_amd import _lib
    L = _lib.lib()
    arity = _header_arities()
    assert len(arity) >= 60
    checked = 0
    for name, n in sorted(arity.items()):
        fn = getattr(L, name)
        if fn.argtypes is None:     # exported and declared in the header, not bound by the Python host
            continue
        assert len(fn.argtypes) == n, "%s: header declares %d parameters, _lib.py binds %d" % (name, n, len(fn.argtypes))
        checked += 1
    assert checked >= 50


def test_kmeans_plusplus_oracle_matches_sklearn_golden(golden_dir):
    """The float64 restatement of k-means++ (oracle/kpp_oracle.py: what the device kernel computes) against scikit-learn
    1.7.2's captured seeds of 6,000 x 10, K = 200 (tests/golden/make_golden.py kpp), and the stored scan of where the two
    stop agreeing: every sample up to 6,000 rows agrees, none of the three 20,000-row samples with data seeds 5 / 6 does."""
    from oracle.kpp_oracle import kmeans_plusplus_f64
    g = np.load(os.path.join(golden_dir, "mbkm_golden.npz"))
    n, k = int(g["kpp_n"]), int(g["kpp_k"])
    rs = np.random.RandomState(int(g["kpp_data_seed"]))
    X = (rs.randn(n, 10) * np.linspace(3, 0.3, 10)).astype(np.float32)
    mine, ids = kmeans_plusplus_f64(X, k, np.random.RandomState(int(g["kpp_stream_seed"])))
    np.testing.assert_array_equal(np.asarray(ids), g["kpp_ids"])
    np.testing.assert_array_equal(mine, g["kpp_centers"])
    scan = g["kpp_scan_seed_rows_equal"]
    assert all(e == 200 for s_, n_, e in scan if n_ <= 6000) and any(e < 200 for s_, n_, e in scan if n_ >= 8000)


@pytest.mark.parametrize("n,F,k", [(500, 8, 10), (3072, 64, 50), (1000, 3, 25)])
def test_kmeans_plusplus_oracle_draws_scikit_learns_seeds(n, F, k):
    """oracle/kpp_oracle.py (the numpy restatement the device seeding is checked against on the GPU tier) walks
    scikit-learn's RandomState call sequence: the SAME rows are chosen and the generator ends in the same state
    (MiniBatchKMeans then draws the same minibatches)."""
    sk = pytest.importorskip("sklearn.cluster")
    from oracle.kpp_oracle import kmeans_plusplus
    rs = np.random.RandomState(n + k)
    X = (rs.randn(n, F) * rs.uniform(0.5, 3, F) + rs.randn(F)).astype(np.float32)
    g_mine, g_ref = np.random.RandomState(7), np.random.RandomState(7)
    mine = kmeans_plusplus(X, k, g_mine)
    ref, _ = sk.kmeans_plusplus(X, k, random_state=g_ref)
    np.testing.assert_array_equal(mine, ref)
    np.testing.assert_array_equal(g_mine.randint(0, 1 << 30, 5), g_ref.randint(0, 1 << 30, 5))


def test_minibatch_host_bookkeeping_matches_scikit_learn():
    """The host restatement of scikit-learn's per-step bookkeeping (_mini_batch_convergence, _random_reassign:
    sklearn/cluster/_kmeans.py:1963-2043) against the private methods themselves, on synthetic inertia sequences."""
    sk = pytest.importorskip("sklearn.cluster")
    from msmbuilder_amd.cluster.minibatchkmeans import _MiniBatchKMeans
    rs = np.random.RandomState(0)
    for trial in range(20):
        mni = [None, 3, 10][trial % 3]
        kw = dict(n_clusters=20, batch_size=64, max_no_improvement=mni, tol=0.0)
        mine, ref = _MiniBatchKMeans(**kw), sk.MiniBatchKMeans(**kw)
        for m in (mine, ref):
            m._batch_size, m._tol = 64, 0.0
            m._ewa_inertia = m._ewa_inertia_min = None
            m._no_improvement = 0
            m._n_since_last_reassign = 0
            m._counts = np.zeros(20, dtype=np.float32)
        base = rs.uniform(50, 500)
        n_samples = int(rs.randint(500, 50000))
        for step in range(200):
            inertia = base * (0.9 ** min(step, 20)) * rs.uniform(0.97, 1.03) * 64
            a = mine._mini_batch_convergence(step, 200, n_samples, 0, inertia)
            b = ref._mini_batch_convergence(step, 200, n_samples, 0, inertia)
            assert a == b and mine._ewa_inertia == ref._ewa_inertia and mine._no_improvement == ref._no_improvement
            counts = rs.randint(0 if step < 3 else 1, 9, 20).astype(np.float32)
            mine._counts, ref._counts = counts.copy(), counts.copy()
            assert mine._random_reassign() == ref._random_reassign()
            assert mine._n_since_last_reassign == ref._n_since_last_reassign
            if a:
                break


def test_dir_npy_dataset_host_protocol(tmp_path):
    """The dir-npy container without a device: file naming, key order, modes, provenance chain, mmap reads and the
    estimator hooks (reference behaviour: msmbuilder/dataset.py:30-97, 158-237, 290-331)."""
    from msmbuilder_amd.dataset import dataset, NumpyDirDataset
    rs = np.random.RandomState(1)
    arrays = [rs.randn(n, 5).astype(np.float32) for n in (40, 1, 300)]
    path = str(tmp_path / "trajs")
    with dataset(path, mode="w", fmt="dir-npy") as ds:
        assert isinstance(ds, NumpyDirDataset)
        for i, a in enumerate(arrays):
            ds[i] = a
    with pytest.raises(ValueError):
        dataset(path, mode="w", fmt="dir-npy")              # exists
    ro = dataset(path, mode="r")
    assert list(ro.keys()) == [0, 1, 2] and len(ro) == 3
    assert sorted(os.listdir(path)) == ["00000000.npy", "00000001.npy", "00000002.npy", "PROVENANCE.txt"]
    umask = os.umask(0)
    os.umask(umask)
    for name in ("00000000.npy", "00000002.npy"):          # np.save's permissions (umask), not mkstemp's 0600
        assert (os.stat(os.path.join(path, name)).st_mode & 0o777) == (0o666 & ~umask)
    for a, b in zip(arrays, ro):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ro.get(2, mmap=True)[7], arrays[2][7])
    with pytest.raises(IndexError):
        ro[5]
    with pytest.raises(IOError):
        ro[0] = arrays[0]

    class Halver(object):
        def fit(self, sequences):
            self.n_ = sum(len(s) for s in sequences)
            return self

        def partial_transform(self, X):
            return X[:, :2] * 0.5

    est = Halver()
    out = ro.fit_transform_with(est, str(tmp_path / "halved"), fmt="dir-npy")
    assert est.n_ == 341 and list(out.keys()) == [0, 1, 2]
    np.testing.assert_array_equal(out[2], arrays[2][:, :2] * 0.5)
    assert "Derived from" in out.provenance and path in out.provenance


@pytest.mark.parametrize("n", [1000, 70_000, 10_000_000, 2**31 + 5, 2**33])
def test_randomstate_randint_batched_equals_sequential(n):
    """MiniBatchKMeans._run draws the S batches of a queued run with ONE ``RandomState.randint(0, n, (S, B))`` call and,
    when the convergence criterion fires after `done` steps, rewinds by restoring the state and re-drawing (done, B).  Both
    rely on legacy ``randint`` consuming the bit stream element by element: the batched call must give the same numbers
    AND leave the same generator state as scikit-learn's S calls of size B."""
    S, B = 7, 1024
    a, b = np.random.RandomState(5), np.random.RandomState(5)
    seq = np.stack([a.randint(0, n, B) for _ in range(S)])
    bat = b.randint(0, n, (S, B))
    assert np.array_equal(seq, bat)
    assert a.randint(0, 2**31 - 1, 16).tolist() == b.randint(0, 2**31 - 1, 16).tolist()   # same state afterwards
    # rewind: restore + re-draw `done` batches == the state after `done` sequential calls
    c, d = np.random.RandomState(9), np.random.RandomState(9)
    st = c.get_state()
    c.randint(0, n, (S, B))
    c.set_state(st)
    c.randint(0, n, (3, B))
    for _ in range(3):
        d.randint(0, n, B)
    assert c.uniform(size=4).tolist() == d.uniform(size=4).tolist()


def _bf16_rne(x32):
    """round-to-nearest-even bfloat16 image of float32 values, returned as float32 (ksc_bf16_rne in distance.hip)"""
    u = np.ascontiguousarray(x32, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1.0, 3.0e6), (1e-20, 0.0), (1e18, 0.0), (1.0, -2.5e3)])
@pytest.mark.parametrize("m", [3, 10, 16])
def test_kcenters_screen_margin_is_safe(scale, offset, m):
    """The screened k-centers passes (distance.hip, kcenters_screen_pass_kernel) leave a row alone when
    ``d~ - eps >= curf`` with d~ evaluated IN FLOAT32 on a bfloat16 copy of the row centred on c0 and
    ``eps = 1.02 * 2^-8 ||x~|| + 2^-19 (||x~|| + ||yc||) + [2^-48 (R + ||y|| + 2 ||c0||) + 1e-37]``.  Numpy emulation of
    exactly that arithmetic (float32 subtract / multiply / add / sqrt, each rounded -- two roundings per product-sum where
    the kernel has one fma: the bound covers both) on adversarial `distances_` values (the largest float32 the screen still
    accepts): the float64 distance the reference would compute must then never be below it.  (With a margin of 2^-9 --
    bfloat16 has 8 significand bits, not 9 -- this test fails.)"""
    f32 = np.float32
    rs = np.random.RandomState(m + int(abs(offset)) % 97)
    n = 200_000
    X = rs.randn(n, m) * scale + offset
    X[::7] = (rs.randn(len(X[::7]), m) * 4.0) * scale + offset            # a wider shell as well
    c0 = X[0].copy()
    Y = X[rs.randint(0, n, 64)]                                           # centres are data rows
    R = np.sqrt((X * X).sum(1).max())
    G2 = ((X - c0) ** 2).sum(1).max()
    xc = (X - c0).astype(np.float32)
    xt = _bf16_rne(xc)                                                    # float32 values with 8 significant bits
    n2 = np.zeros(n, dtype=f32)
    for f in range(m):
        n2 = (n2 + xt[:, f] * xt[:, f]).astype(f32)
    nrm = np.sqrt(n2).astype(f32)
    worst = np.inf
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        for y in Y:
            yc = y - c0
            ycn2 = float((yc * yc).sum())
            eps0 = (R + np.sqrt((y * y).sum()) + 2.0 * np.sqrt((c0 * c0).sum())) * 2.0 ** -48 + 1e-37
            if not (G2 < 1e36 and ycn2 < 1e36):
                continue                                                  # the kernel switches the screen off: nothing passes
            eps0f = f32(eps0 * 1.000001)
            ycnf = f32(np.sqrt(ycn2) * 1.000001)
            ycf = yc.astype(f32)
            a = np.zeros(n, dtype=f32)
            for f in range(m):
                t = (xt[:, f] - ycf[f]).astype(f32)
                a = (a + (t * t).astype(f32)).astype(f32)
            dt = np.sqrt(a).astype(f32)
            eps = (nrm * f32(2.0 ** -8 * 1.02) + (f32(2.0 ** -19) * (nrm + ycnf)).astype(f32) + eps0f).astype(f32)
            # the reference's distance: float64, features in order, separately rounded multiply and add, sqrt
            ar = np.zeros(n)
            for f in range(m):
                d = X[:, f] - y[f]
                ar = ar + d * d
            dref = np.sqrt(ar)
            # adversarial distances_: curf = the largest float32 the test `d~ - eps >= curf` (in float32) still accepts
            cur = (dt - eps).astype(f32).astype(np.float64)
            ok = cur > 0
            assert np.all(dref[ok] >= cur[ok]), (scale, offset, m)
            worst = min(worst, float(np.min((dref[ok] - cur[ok]) / np.maximum(eps[ok].astype(np.float64), 1e-300))))
    assert worst >= 0.0


@pytest.mark.parametrize("scale,offset,m", [(1.0, 0.0, 10), (1e-3, 250.0, 7), (3e4, -1e6, 16), (1.0, 3e6, 2), (1e-12, 0.0, 5)])
def test_kcenters_byte_screen_margin_is_safe(scale, offset, m):
    """The default screen copy of round 3: signed bytes q_j with one bfloat16 scale sf per row (127 sf >= max |x_j - c0_j|,
    q_j = rint((x_j - c0_j) / sf) in float64), d~ = sqrtf(sum fl32(q_j sf - yc_j)^2) and
    ``eps = sf (0.51 * 1.02 sqrt(m') + 2^-19 * 127 sqrt(m') * 1.001) + 2^-19 ||yc|| + eps0`` (m' = m rounded up to even).
    Numpy emulation of that arithmetic on adversarial `distances_`, as in the bfloat16 test above: the reference's float64
    distance must never be below the largest float32 the screen still accepts."""
    f32 = np.float32
    rs = np.random.RandomState(m + int(abs(offset)) % 89)
    n = 200_000
    X = rs.randn(n, m) * scale + offset
    X[::5] = (rs.randn(len(X[::5]), m) * 6.0) * scale + offset
    X[1::11, 0] = offset                                                  # coordinates that sit exactly on the origin's
    c0 = X[0].copy()
    Y = X[rs.randint(0, n, 48)]
    R = np.sqrt((X * X).sum(1).max())
    G2 = ((X - c0) ** 2).sum(1).max()
    xd = X - c0
    smax = np.abs(xd).max(1)
    sf32 = (smax * (1.0000002 / 127.0)).astype(f32)
    bits = ((sf32.view(np.uint32).astype(np.uint64) + 0xffff) >> 16).astype(np.uint32) << 16   # smallest bfloat16 >= sf32
    sf = bits.view(f32).copy()
    sf[smax == 0] = 0
    small = (smax > 0) & (smax < 1e-30)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(sf[:, None] > 0, np.rint(xd / sf[:, None].astype(np.float64)), 0.0)
    assert np.abs(q).max() <= 127
    q = q.astype(f32)
    mp = m + (m & 1)
    qsq = {2: 1.4143, 4: 2.0, 6: 2.4495, 8: 2.8285, 10: 3.1623, 12: 3.4642, 14: 3.7417, 16: 4.0}[mp]
    QA = f32(f32(0.51) * f32(1.02) * f32(qsq) + f32(2.0 ** -19) * f32(127.0) * f32(qsq) * f32(1.001))
    worst = np.inf
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        for y in Y:
            yc = y - c0
            ycn2 = float((yc * yc).sum())
            eps0 = (R + np.sqrt((y * y).sum()) + 2.0 * np.sqrt((c0 * c0).sum())) * 2.0 ** -48 + 1e-37
            if not (G2 < 1e36 and ycn2 < 1e36):
                continue
            eps0f = f32(eps0 * 1.000001)
            ycnf = f32(np.sqrt(ycn2) * 1.000001)
            ycf = yc.astype(f32)
            a = np.zeros(n, dtype=f32)
            for f in range(m):
                t = ((q[:, f] * sf).astype(f32) - ycf[f]).astype(f32)     # q sf is exact in float32; the kernel's fma rounds once too
                a = (a + (t * t).astype(f32)).astype(f32)
            dt = np.sqrt(a).astype(f32)
            eps = ((sf * QA).astype(f32) + (f32(2.0 ** -19) * ycnf + eps0f).astype(f32)).astype(f32)
            ar = np.zeros(n)
            for f in range(m):
                d = X[:, f] - y[f]
                ar = ar + d * d
            dref = np.sqrt(ar)
            cur = (dt - eps).astype(f32).astype(np.float64)
            ok = (cur > 0) & ~small                                        # (rows with a scale below 1e-30 carry a NaN scale: never screened)
            assert np.all(dref[ok] >= cur[ok]), (scale, offset, m)
            worst = min(worst, float(np.min((dref[ok] - cur[ok]) / np.maximum(eps[ok].astype(np.float64), 1e-300)))) if ok.any() else worst
    assert worst >= 0.0


@pytest.mark.parametrize("kind", ["gaussian", "lattice", "duplicates", "trajectory"])
def test_kcenters_batched_selection_equals_the_sequential_scan(kind):
    """The rule behind several k-centers centres per pass (csrc/distance.hip, kcb_select_kernel), played in numpy: list the
    rows whose distance exceeds theta; the listed row of largest distance (lowest row on ties) is the next centre; the other
    listed rows take d = min(d, dist(row, centre)); the next largest is the centre after that IF it is strictly above theta
    (every unlisted row is at most theta, and distances only shrink) -- else a new pass.  The centre sequence must be the
    one-centre-per-pass loop's (kcenters.py:79-102: argmax of distances_, first occurrence), also on integer lattices and
    duplicated rows where ties are everywhere, whatever theta is."""
    rs = np.random.RandomState({"gaussian": 1, "lattice": 2, "duplicates": 3, "trajectory": 4}[kind])
    n, m, K = 6000, 4, 60
    if kind == "gaussian":
        X = rs.randn(n, m)
    elif kind == "lattice":
        X = rs.randint(-4, 5, size=(n, m)).astype(np.float64)
    elif kind == "duplicates":
        X = rs.randn(n // 6, m)[rs.randint(0, n // 6, n)]
    else:
        X = np.cumsum(rs.randn(n, m) * 0.1, axis=0)
    def dist_to(c):
        d = X - X[c]
        return np.sqrt((d * d).sum(1))
    # the reference loop
    ref = [0]
    dref = dist_to(0)
    for _ in range(K - 1):
        c = int(np.argmax(dref))
        ref.append(c)
        dref = np.minimum(dref, dist_to(c))
    for theta_rule in (0.97, 0.8, 1.5, 0.0):
        ids = [0]
        dist = dist_to(0)
        passes = 0
        while len(ids) < K:
            passes += 1
            vmax = dist.max()
            theta = theta_rule * vmax                                     # any threshold will do for correctness
            listed = np.flatnonzero(dist > theta)
            if len(listed) == 0 or len(listed) > 512:                      # empty / overflowed list: the partials' argmax alone
                batch = [int(np.argmax(dist))]
            else:
                cv = dist[listed].copy()
                alive = np.ones(len(listed), bool)
                batch = []
                while len(ids) + len(batch) < K and len(batch) < 16:
                    if not alive.any():
                        break
                    v = np.where(alive, cv, -1.0)
                    j = int(np.flatnonzero(v == v.max())[0])               # largest value, lowest row (listed is ascending)
                    if batch and not (v[j] > theta):
                        break
                    batch.append(int(listed[j]))
                    alive[j] = False
                    d = X[listed] - X[listed[j]]
                    cv = np.minimum(cv, np.sqrt((d * d).sum(1)))
            for c in batch:                                                # the pass applies the batch in order
                dist = np.minimum(dist, dist_to(c))
            ids += batch
        assert ids == ref, (kind, theta_rule)
        if theta_rule == 0.97 and kind in ("gaussian", "trajectory"):
            assert passes < K // 2                                         # and it does batch


@pytest.mark.parametrize("scale,offset,m", [(1.0, 0.0, 10), (1e-3, 250.0, 7), (3e4, -1e6, 16), (1.0, 3e6, 2), (1e-12, 0.0, 5), (1e15, 0.0, 9)])
def test_assign_screen_bound_holds(scale, offset, m):
    """The float32 sweep of the screened assign_nearest (csrc/distance_small.hip, assign_screen_kernel): rows and centres
    centred on centre 0 and rounded to float32, ``w_k = |c~_k|^2 - 2 x~.c~_k + (|x~|^2 + 2 Eh)`` with two partial dot products
    (even / odd features), ``Eh = 1.01 (m' + 8) 2^-24 ((|x~| + max|c~|) 1.001)^2``.  What the kernel relies on, checked in
    numpy with every float32 operation rounded (two roundings where the kernel has one fma: the bound covers both) against
    extended-precision distances of the float64 data: ``|w_k - 2 Eh - D_k^2| <= Eh``, ``w_k >= 0``, and therefore for every
    row ``v2 - 4 Eh <= min over k != k1 of D_k^2`` with v2 the second smallest key value, its low 8 bits cleared."""
    f32 = np.float32
    rs = np.random.RandomState(m + int(abs(offset)) % 83)
    n, K = 20000, 48
    X = rs.randn(n, m) * scale + offset
    X[::9] = (rs.randn(len(X[::9]), m) * 5.0) * scale + offset
    Y = X[rs.randint(0, n, K)].copy()
    Y[5] = Y[2]                                                            # a duplicated centre
    o = Y[0]
    mp = m + (m & 1)
    xt = np.zeros((n, mp), f32); xt[:, :m] = (X - o).astype(f32)
    ct = np.zeros((K, mp), f32); ct[:, :m] = (Y - o).astype(f32)
    def sq(v):
        a = np.zeros(len(v), f32)
        for f in range(mp):
            a = (a + (v[:, f] * v[:, f]).astype(f32)).astype(f32)
        return a
    nx, nc = sq(xt), sq(ct)
    if not (nc.max() <= 3e37 and np.all(nx <= 3e37)):
        pytest.skip("beyond the float32 range: the kernel does not screen such data")
    rnc = np.sqrt(nc.max()).astype(f32)
    cE = f32((mp + 8) * 2.0 ** -24)
    S = ((np.sqrt(nx).astype(f32) + rnc).astype(f32) * f32(1.001)).astype(f32)
    Eh = (((cE * S).astype(f32) * S).astype(f32) * f32(1.01)).astype(f32)
    off = (nx + (f32(2.0) * Eh).astype(f32)).astype(f32)
    W = np.zeros((n, K), f32)
    for k in range(K):
        sx = (f32(-0.5) * off).astype(f32)
        sy = np.zeros(n, f32)
        for g in range(mp // 2):
            sx = (sx + (xt[:, 2 * g] * ct[k, 2 * g]).astype(f32)).astype(f32)
            sy = (sy + (xt[:, 2 * g + 1] * ct[k, 2 * g + 1]).astype(f32)).astype(f32)
        w = (nc[k] + (f32(-2.0) * (sx + sy).astype(f32)).astype(f32)).astype(f32)
        W[:, k] = np.maximum(w, f32(0.0))
    Xl, Yl = X.astype(np.longdouble), Y.astype(np.longdouble)
    D2 = np.stack([((Xl - Yl[k]) ** 2).sum(1) for k in range(K)], 1)
    Ehl = Eh.astype(np.longdouble)[:, None]
    err = np.abs(W.astype(np.longdouble) - 2 * Ehl - D2)
    assert np.all(err <= Ehl), float((err / np.maximum(Ehl, np.longdouble(1e-300))).max())
    assert np.all(W >= 0)
    # the two smallest keys (index in the low 8 bits) and the bound the fast path uses
    keys = (W.view(np.uint32) & np.uint32(0xffffff00)) | np.arange(K, dtype=np.uint32)[None, :]
    order = np.argsort(keys, axis=1, kind="stable")
    k1 = order[:, 0]
    v2 = np.take_along_axis(keys, order[:, 1:2], 1)[:, 0]
    v2 = (v2 & np.uint32(0xffffff00)).view(f32).astype(np.longdouble)
    D2o = D2.copy()
    D2o[np.arange(n), k1] = np.inf
    assert np.all(v2 - 4 * Ehl[:, 0] <= D2o.min(1))


def test_dir_npy_dataset_payload_reader(tmp_path):
    """The container's host reads go through the native header parser (msm_npy_info) + memmap / fromfile; payloads the
    parser does not describe (structured dtypes) fall back to numpy's reader; writes are atomic renames; keys iterate in
    numeric order whatever order the files were created in; stray files are ignored."""
    from msmbuilder_amd.dataset import dataset
    rs = np.random.RandomState(3)
    path = str(tmp_path / "d")
    arrays = {7: rs.randn(5, 3), 0: rs.randint(0, 9, (4, 2)).astype(np.int32), 11: np.zeros((0, 3), np.float32),
              3: np.asfortranarray(rs.randn(4, 5)), 2: np.array([True, False, True]),
              5: np.array([(1, 2.0)], dtype=[("a", "<i4"), ("b", "<f8")]), 6: rs.randn(6).astype(np.float16)}
    with dataset(path, mode="w") as ds:
        for k, a in arrays.items():
            ds[k] = a
    open(os.path.join(path, "notes.txt"), "w").write("x")
    open(os.path.join(path, "123.npy"), "w").write("x")                  # not an 8-digit name: not a trajectory
    ro = dataset(path)
    assert list(ro.keys()) == sorted(arrays) and len(ro) == len(arrays)
    assert not [f for f in os.listdir(path) if f.endswith(".part")]     # no temporary files left behind
    for k, v in ro.items():
        a = arrays[k]
        assert v.dtype == a.dtype and v.shape == a.shape and np.array_equal(v, a)
        if a.dtype.names is None and a.size:
            mm = ro.get(k, mmap=True)
            assert isinstance(mm, np.memmap) and not mm.flags.writeable and np.array_equal(mm, a)
            assert mm.flags.f_contiguous == a.flags.f_contiguous or a.ndim < 2
    assert ro.get(11, mmap=True).shape == (0, 3)
    with pytest.raises(IndexError):
        ro.get(4)
    # append mode adds to an existing store and keeps its provenance file
    with dataset(path, mode="a") as ap:
        ap[4] = np.ones((2, 2))
    assert list(dataset(path).keys()) == sorted(list(arrays) + [4])
    with pytest.raises(NotImplementedError):
        dataset(path, fmt="hdf5")
    with pytest.raises(ValueError):
        dataset(path, mode="x")


def test_bench_cpu_baseline_runs_on_the_host():
    """bench.py's cpu_baseline leg (the oracle on the host cores, the reference's libdistance for the clustering legs when
    oracle/_ref is present): structure of what it reports, on a tiny sample."""
    import bench
    rs = np.random.RandomState(0)
    X = [(np.cumsum(rs.randn(1500, 32), 0) * 0.01 + rs.randn(1500, 32)).astype(np.float32) for _ in range(5)]
    out, oracle_model, used = bench.cpu_baseline(X, 10, 4, 12, budget_s=0.5)
    assert out["unit"] == "frames/s" and out["value"] > 0 and out["cores"] == os.cpu_count()
    assert out["kind"] == "port" and set(out["kinds"]) == {"tica", "clustering"}
    assert out["tica_fit_best_frames_per_s"] >= max(out["tica_fit_frames_per_s"], out["tica_fit_1thread_frames_per_s"]) * 0.999
    assert out["threadpool_info"] and all(p["num_threads"] == 1 for p in out["threadpool_info_limited"] if p["user_api"] == "blas")
    assert len(used) >= 1 and oracle_model.n_sequences_ == len(used)


def test_adjacent_view_only_joins_what_lies_back_to_back():
    """`_lib.adjacent_view` (one launch for `transform` / `predict` over trajectories that are views of one allocation):
    joins consecutive C-contiguous slices of one base array, and nothing else."""
    from msmbuilder_amd._lib import adjacent_view
    X = np.arange(60, dtype=np.float32).reshape(12, 5)
    v = adjacent_view([X[:3], X[3:7], X[7:]])
    assert v is not None and v.shape == (12, 5) and np.array_equal(v, X) and not v.flags.writeable
    assert adjacent_view([X[:3], X[3:3], X[3:]]).shape == (12, 5)          # an empty trajectory in between
    assert adjacent_view([X[:3], X[4:7]]) is None                          # a gap
    assert adjacent_view([X[3:7], X[:3]]) is None                          # out of order
    assert adjacent_view([X[:3].copy(), X[3:7].copy()]) is None            # separate allocations
    assert adjacent_view([X]) is None and adjacent_view([]) is None
    assert adjacent_view([X[:3], X[3:7].astype(np.float64)]) is None       # dtypes differ
    assert adjacent_view([X[:3], X[3:7, :4]]) is None                      # widths differ / not contiguous
    assert adjacent_view([X[:3], [[1.0] * 5]]) is None                     # not arrays

