"""Parity at BASELINE.json's full (single-GPU) sizes, through the C ABI: either against the
oracle where it finishes in seconds, or through size-independent properties / an independent
float64 contraction of the same data computed with torch on the device."""
import ctypes as C
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _lagged(m, C):
    """raw lagged moment, or its symmetric part when the handle keeps only that (fp32 sum/difference kernel)"""
    return 0.5 * (C + C.T) if m._lagged_symmetrised else C


def _accumulators(m):
    m._pull()
    return m._outer_0_to_T_lagged, m._outer_gram_sum, m._sum_0_to_TminusTau, m._sum_tau_to_T


def test_config5_bf16_mfma_vs_fp32(gpu, monkeypatch):
    """configs[4] at single-GPU scale: F = 2048, bf16-MFMA covariance against the fp32 path
    (stated tolerance: eigenvalues rtol 1e-3 for bf16, 1e-5 for the two-term split)."""
    from msmbuilder_amd import tICA
    g = torch.Generator(device="cuda").manual_seed(9)
    N, F, T = 200_000, 2048, 10_000
    z = torch.randn(N, 12, generator=g, device="cuda")
    for s in range(N // T):
        z[s * T:(s + 1) * T] = z[s * T:(s + 1) * T].cumsum(0) * 0.02
    X = (z @ torch.randn(12, F, generator=g, device="cuda") + torch.randn(N, F, generator=g, device="cuda")
         + torch.linspace(-1, 1, F, device="cuda")).float().contiguous()
    seqs = list(X.view(N // T, T, F).unbind(0))
    ev = {}
    for mode in ("f32", "bf16x2", "bf16"):
        monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev[mode] = tICA(n_components=8, lag_time=100).fit(seqs).eigenvalues_
    np.testing.assert_allclose(ev["bf16"], ev["f32"], rtol=1e-3)
    np.testing.assert_allclose(ev["bf16x2"], ev["f32"], rtol=1e-5)
    assert 0 < ev["f32"][-1] and ev["f32"][0] < 1


@pytest.mark.parametrize("mode,rtol", [("f32", 2e-6), ("f64", 1e-12), ("bf16x2", 1e-5)])
def test_config2_tica_1M_x_128_vs_fp64_contraction(gpu, monkeypatch, mode, rtol):
    """configs[1]: 1M x 128 fp32, lag 100, one trajectory and the same data as 100 trajectories."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    g = torch.Generator(device="cuda").manual_seed(7)
    N, F, lag = 1_000_000, 128, 100
    z = torch.randn(N, 8, generator=g, device="cuda").cumsum(0) * 0.01
    X = (z @ torch.randn(8, F, generator=g, device="cuda") + torch.randn(N, F, generator=g, device="cuda")
         + torch.linspace(-2, 2, F, device="cuda")).float().contiguous()
    Xd = X.double()
    for n_seq in (1, 100):
        T = N // n_seq
        seqs = list(X.view(n_seq, T, F).unbind(0))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=5, lag_time=lag).fit(seqs)
        Cm, Gm, s0, st = _accumulators(m)
        V = Xd.view(n_seq, T, F)
        A, B = V[:, :-lag], V[:, lag:]
        Cr = torch.einsum("sti,stj->ij", A, B).cpu().numpy()
        Gr = (torch.einsum("sti,stj->ij", A, A) + torch.einsum("sti,stj->ij", B, B)).cpu().numpy()
        scale = np.abs(Gr).max()
        np.testing.assert_allclose(Cm, _lagged(m, Cr), rtol=0, atol=rtol * scale)
        np.testing.assert_allclose(Gm, Gr, rtol=0, atol=rtol * scale)
        np.testing.assert_allclose(s0, A.sum((0, 1)).cpu().numpy(), rtol=1e-11)
        np.testing.assert_allclose(st, B.sum((0, 1)).cpu().numpy(), rtol=1e-11)
        assert m.n_observations_ == N and m.n_sequences_ == n_seq
        assert np.array_equal(Gm, Gm.T)
        assert np.all(np.diff(m.eigenvalues_) <= 0) and m.eigenvalues_[0] < 1.0 + 1e-9


@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_huge_lag_offsets_beyond_4GiB(gpu, monkeypatch, mode):
    """lag * row pitch > 4 GiB: the lag must live in a 64-bit base pointer, not in 32-bit lane offsets."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", mode)
    g = torch.Generator(device="cuda").manual_seed(5)
    N, F, lag = 2_300_000, 512, 2_200_000           # 2.2M * 2 KiB = 4.5 GB
    X = torch.randn(N, F, generator=g, device="cuda") + 0.25
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=2, lag_time=lag).fit([X])
    Cm, Gm, s0, st = _accumulators(m)
    A, B = X[:-lag].double(), X[lag:].double()
    Cr = (A.T @ B).cpu().numpy()
    Gr = (A.T @ A + B.T @ B).cpu().numpy()
    tol = (2e-6 if mode == "f32" else 1e-12) * np.abs(Gr).max()   # fp32 partials of <= 8192 all-positive terms on the diagonal
    np.testing.assert_allclose(Cm, _lagged(m, Cr), rtol=0, atol=tol)
    np.testing.assert_allclose(Gm, Gr, rtol=0, atol=tol)
    np.testing.assert_allclose(s0, A.sum(0).cpu().numpy(), rtol=1e-11)
    np.testing.assert_allclose(st, B.sum(0).cpu().numpy(), rtol=1e-11)


def test_tica_additivity_and_import_export(gpu, monkeypatch):
    """sum of per-shard exports == one fit over everything (the multi-GPU exchange, single process);
    import(export(x)) is the identity; shift invariance of the covariance."""
    from msmbuilder_amd import tICA
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    rs = np.random.RandomState(0)
    seqs = [rs.randn(int(n), 40).astype(np.float32) + 1.5 for n in rs.randint(60, 4000, size=31)]
    full = tICA(n_components=4, lag_time=50).fit(seqs)
    parts = [tICA(n_components=4, lag_time=50).fit(seqs[i::4]) for i in range(4)]
    for k in range(4):
        ref = _accumulators(full)[k]
        np.testing.assert_allclose(sum(_accumulators(p)[k] for p in parts), ref, rtol=1e-12, atol=1e-9 * np.abs(ref).max())
    # merge the shards through the C ABI's packed import/export
    L = gpu.lib()
    n = int(L.msm_tica_packed_size(parts[0]._handle))
    total = np.zeros(n)
    for p in parts:
        buf = np.empty(n)
        gpu.check(L.msm_tica_export_packed(p._handle, C.c_void_p(buf.ctypes.data), 0))
        total += buf
    merged = parts[0]
    gpu.check(L.msm_tica_import_packed(merged._handle, C.c_void_p(total.ctypes.data), 0))
    merged.n_observations_, merged.n_sequences_ = int(total[-2]), int(total[-1])
    merged._host_stale = merged._is_dirty = True
    np.testing.assert_allclose(merged.eigenvalues_, full.eigenvalues_, rtol=1e-11)
    assert merged.n_observations_ == full.n_observations_
    # covariance / offset correlation are invariant under a constant shift of the data
    shifted = tICA(n_components=4, lag_time=50).fit([s + np.float32(8.0) for s in seqs])
    np.testing.assert_allclose(shifted.eigenvalues_, full.eigenvalues_, rtol=1e-6)
    np.testing.assert_allclose(shifted.means_ - 8.0, full.means_, atol=1e-6)


def test_config3_kcenters_280k_bit_exact(gpu):
    """configs[2] shape: 280,000 x 10 (tICA space), K = 200: bit-exact against the C oracle."""
    from msmbuilder_amd import KCenters
    from oracle.libdistance_oracle import Oracle
    o = Oracle()
    rs = np.random.RandomState(3)
    for dt in (np.float64, np.float32):
        Y = (rs.randn(280_000, 10) * np.linspace(3, 0.3, 10)).astype(dt)
        seqs = [Y[i * 10_000:(i + 1) * 10_000] for i in range(28)]
        m = KCenters(n_clusters=200, random_state=0).fit(seqs)
        ids, labels, dist = o.kcenters_fit(Y, 200, "euclidean", m.cluster_ids_[0])
        assert m.cluster_ids_ == list(ids)
        assert np.array_equal(np.concatenate(m.labels_), labels)
        assert np.array_equal(np.concatenate(m.distances_), dist)
        assert m.inertia_ == np.sum(dist)
        pred = np.concatenate(m.predict(seqs))
        assert np.array_equal(pred, o.assign_nearest(Y, np.ascontiguousarray(Y[ids]), "euclidean")[0])
        # size-independent properties: every centre labels itself at distance 0; distances_ are the
        # distances to the assigned centre; they never exceed the covering radius
        assert np.array_equal(labels[ids], np.arange(200)) and np.all(dist[ids] == 0)
        assert np.array_equal(pred, labels)


def test_config3_stress_280k_x_171_without_tica_bit_exact(gpu):
    """SURVEY 8(d)'s C3 stress variant: KCenters(200).fit and assign_nearest on the RAW contact features, 280,000 x 171
    float32 (28 trajectories x 10,000), no tICA in front -- the wide-row kernels (171 is odd: no 16-byte aligned rows, the
    scalar-staged pair kernel) at full size.  Bit for bit against the C oracle and, where it is built, against the
    reference's own headers compiled from where they lie (oracle/_ref; assign.hpp:50-91)."""
    from msmbuilder_amd import KCenters, libdistance
    from oracle.libdistance_oracle import Oracle, Ref
    o = Oracle()
    rs = np.random.RandomState(171)
    n, F, K = 280_000, 171, 200
    # contact-distance-like features: positive, column means 0.4 - 2.5 nm, slow drift along each trajectory + noise
    base = np.linspace(0.4, 2.5, F).astype(np.float32)
    Y = np.empty((n, F), dtype=np.float32)
    for t in range(28):
        drift = np.cumsum(rs.randn(10_000, 8).astype(np.float32) * 0.01, axis=0) @ rs.randn(8, F).astype(np.float32)
        Y[t * 10_000:(t + 1) * 10_000] = np.abs(base + 0.3 * drift + 0.05 * rs.randn(10_000, F).astype(np.float32))
    Y[123_456] = Y[7]                                      # a duplicate row: an exact tie for assign_nearest's strict <
    seqs = [Y[i * 10_000:(i + 1) * 10_000] for i in range(28)]
    m = KCenters(n_clusters=K, random_state=0).fit(seqs)
    ids, labels, dist = o.kcenters_fit(Y, K, "euclidean", m.cluster_ids_[0])
    assert m.cluster_ids_ == list(ids)
    assert np.array_equal(np.concatenate(m.labels_), labels)
    assert np.array_equal(np.concatenate(m.distances_), dist)
    assert m.inertia_ == np.sum(dist)
    assert m.cluster_centers_.dtype == np.float32 and np.array_equal(m.cluster_centers_, Y[ids])
    centres = np.ascontiguousarray(Y[ids])
    got, inertia = libdistance.assign_nearest(Y, centres, "euclidean")
    want = o.assign_nearest(Y, centres, "euclidean")
    assert np.array_equal(got, want[0]) and abs(inertia - want[1]) <= 1e-13 * want[1]
    assert np.array_equal(np.concatenate(m.predict(seqs)), want[0])
    if Ref.available():
        ref = Ref().assign_nearest(Y, centres, "euclidean")
        assert np.array_equal(got, ref[0]) and abs(inertia - ref[1]) <= 1e-13 * ref[1]
    assert got[123_456] == got[7] and np.array_equal(got[ids], np.arange(K))


@pytest.mark.parametrize("metric", ["euclidean", "cityblock", "chebyshev"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kcenters_clustered_rows_beyond_2_18(gpu, metric, dtype):
    """>= 2^18 rows of CLUSTERED data (where the per-row pruning test and, for float64 euclidean rows, the screened and
    batched passes skip most rows) against the C oracle, bit for bit -- with duplicate rows (zero distances, argmax ties
    decided by the row index) and a row count that is not a multiple of any tile."""
    from msmbuilder_amd import KCenters
    from oracle.libdistance_oracle import Oracle
    o = Oracle()
    rs = np.random.RandomState(11)
    n, f, k = 300_123, 10, 72
    blobs = rs.randn(40, f) * 6.0
    Y = (blobs[rs.randint(0, 40, n)] + rs.randn(n, f) * 0.4).astype(dtype)
    Y[1000:1040] = Y[7]
    Y[250_000:250_005] = Y[123_456]
    m = KCenters(n_clusters=k, metric=metric, random_state=5).fit([Y[:100_000], Y[100_000:]])
    ids, labels, dist = o.kcenters_fit(Y, k, metric, m.cluster_ids_[0])
    assert m.cluster_ids_ == list(ids)
    assert np.array_equal(np.concatenate(m.labels_), labels)
    assert np.array_equal(np.concatenate(m.distances_), dist)
    assert m.inertia_ == np.sum(dist)


@pytest.mark.parametrize("case", ["plain", "offset", "beyond_float32", "nan_row", "shell", "anisotropic", "lattice", "tiny", "huge"])
@pytest.mark.parametrize("m", [3, 10, 16])
def test_kcenters_float32_screened_passes(gpu, case, m):
    """float64 rows, euclidean: after 4 plain passes the fit continues with SCREENED passes (a bfloat16 copy of the rows,
    centred on the first centre, + distances rounded up to float32 decide which rows cannot change; the others are
    re-evaluated exactly).  Bit for bit against the C oracle, with duplicate rows (float32-image ties in the argmax), a large
    common offset (what the centring is for), a row outside the float32 range (the screen must switch itself off) and a NaN
    row (never assigned, distance stays inf)."""
    from msmbuilder_amd import KCenters
    from oracle.libdistance_oracle import Oracle
    o = Oracle()
    rs = np.random.RandomState(m)
    n, k = 100_003, 64
    Y = rs.randn(n, m)
    Y[2000:2030] = Y[11]
    Y[90_000:90_004] = Y[54_321]
    if case == "offset":
        Y += 3.0e6
    elif case == "beyond_float32":
        Y[777] = 1.0e39
    elif case == "nan_row":
        Y[4242, 0] = np.nan
    elif case == "shell":  # every row at the same distance from the copy's origin: the rounding margin is tight for all of them
        Y = Y / np.linalg.norm(Y, axis=1, keepdims=True) * 7.0 + 0.01 * rs.randn(n, m)
        Y[2000:2030] = Y[11]
    elif case == "anisotropic":  # configs[2]-like scales; what caught a margin of 2^-9 ||x~|| (bfloat16's unit roundoff is 2^-8)
        Y *= np.linspace(3.0, 0.3, m)
    elif case == "lattice":  # half-integer lattice: masses of exactly equal distances, duplicates and argmax ties
        Y = np.round(Y * 2.0) / 2.0
    elif case == "tiny":  # everything far below float32's (and bfloat16's) normal range scale of interest
        Y *= 1e-20
    elif case == "huge":
        Y *= 1e18
    m_ = KCenters(n_clusters=k, random_state=2).fit([Y[:40_000], Y[40_000:]])
    ids, labels, dist = o.kcenters_fit(Y, k, "euclidean", m_.cluster_ids_[0])
    assert m_.cluster_ids_ == list(ids)
    assert np.array_equal(np.concatenate(m_.labels_), labels)
    assert np.array_equal(np.concatenate(m_.distances_), dist)


def test_assign_nearest_1M_device_resident_properties(gpu):
    from msmbuilder_amd import libdistance as ld
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(1_000_000, 10, generator=g, device="cuda", dtype=torch.float64)
    Y = X[::5000].contiguous()                       # 200 centres that are data points
    lab, inertia = ld.assign_nearest(X, Y, "euclidean")
    d = ld.cdist(X[:50_000], Y, "euclidean")
    assert torch.equal(lab[:50_000], d.argmin(1))
    assert torch.equal(lab[::5000], torch.arange(200, device="cuda"))
    dmin = ld.dist(X, Y[0], "euclidean")
    assert float(dmin[0]) == 0.0
    idx = torch.arange(0, 1_000_000, 7, device="cuda")
    lab_i, _ = ld.assign_nearest(X, Y, "euclidean", idx)
    assert torch.equal(lab_i, lab[::7])
    # cityblock against a float64 torch evaluation (different summation order -> tolerance, labels may tie)
    labc, _ = ld.assign_nearest(X[:50_000].contiguous(), Y, "cityblock")
    ref = (X[:50_000, None, :] - Y[None]).abs().sum(-1)     # (torch.cdist(p=1) returns garbage here on ROCm)
    pick = ref.gather(1, labc.view(-1, 1)).squeeze(1)
    assert torch.all(pick <= ref.min(1).values * (1 + 1e-12))


def test_config4_kmeans_label_1M_x_512_k1000(gpu):
    from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
    g = torch.Generator(device="cuda").manual_seed(2)
    Cn = torch.randn(1000, 512, generator=g, device="cuda") * 2
    X = Cn[torch.randint(0, 1000, (1_000_000,), generator=g, device="cuda")] + torch.randn(1_000_000, 512, generator=g, device="cuda")
    lab, inertia = label_inertia(X, Cn.cpu().numpy())
    d2 = torch.cdist(X[:100_000].double(), Cn.double()) ** 2
    ref = d2.argmin(1)
    mism = (lab[:100_000].long() != ref).nonzero().flatten()
    picked = d2[mism, lab[:100_000].long()[mism]]
    assert torch.all(picked <= d2[mism].min(1).values * (1 + 1e-4))      # only fp32 near-ties may differ
    assert len(mism) < 100
    ref_inertia = ((X.double() - Cn.double()[lab.long()]) ** 2).sum().item()
    assert abs(inertia - ref_inertia) <= 1e-6 * ref_inertia


def test_c_abi_direct_without_torch(gpu):
    """The ABI on its own device buffers: ld > F, check_finite = 0 with the sticky flag, batch table."""
    L = gpu.lib()
    rs = np.random.RandomState(5)
    F, ld, lag = 24, 32, 3
    host = rs.randn(500, ld).astype(np.float32)
    dptr = C.c_void_p()
    gpu.check(L.msm_malloc(C.byref(dptr), host.nbytes))
    gpu.check(L.msm_memcpy_h2d(dptr, C.c_void_p(host.ctypes.data), host.nbytes))
    h = C.c_void_p()
    gpu.check(L.msm_tica_create(C.byref(h), F, lag, gpu.TICA_F64))
    # two trajectories inside one padded device array + one too-short one
    ptrs = (C.c_void_p * 3)(dptr.value, dptr.value + 300 * ld * 4, dptr.value + 10 * ld * 4)
    rows = (C.c_int64 * 3)(300, 200, 3)
    skipped = C.c_int64(0)
    gpu.check(L.msm_tica_accumulate_batch(h, ptrs, rows, 3, 4, ld, 1, 0, C.byref(skipped)))
    assert skipped.value == 1
    flag = C.c_int(-1)
    gpu.check(L.msm_tica_nonfinite(h, C.byref(flag)))
    assert flag.value == 0
    Cm, Gm, s0, st = np.empty((F, F)), np.empty((F, F)), np.empty(F), np.empty(F)
    nobs, nseq = C.c_int64(), C.c_int64()
    gpu.check(L.msm_tica_export(h, Cm.ctypes.data, Gm.ctypes.data, s0.ctypes.data, st.ctypes.data,
                                C.byref(nobs), C.byref(nseq)))
    assert (nobs.value, nseq.value) == (500, 2)
    X = host[:, :F].astype(np.float64)
    ref = sum(a[:-lag].T @ a[lag:] for a in (X[:300], X[300:]))
    np.testing.assert_allclose(Cm, ref, rtol=1e-12, atol=1e-10)
    # non-finite input with check_finite = 1 is rejected and leaves the state alone
    bad = host.copy()
    bad[7, 3] = np.inf
    gpu.check(L.msm_memcpy_h2d(dptr, C.c_void_p(bad.ctypes.data), bad.nbytes))
    rc = L.msm_tica_accumulate(h, dptr, 4, 300, ld, 1, 1, None)
    assert rc == gpu.MSM_ERR_NONFINITE and b"NaN" in L.msm_last_error()
    C2 = np.empty((F, F))
    gpu.check(L.msm_tica_export(h, C2.ctypes.data, None, None, None, None, None))
    np.testing.assert_array_equal(C2, Cm)
    # bad arguments
    assert L.msm_tica_accumulate(h, dptr, 2, 300, ld, 1, 1, None) == gpu.MSM_ERR_INVALID
    assert L.msm_tica_accumulate(h, dptr, 4, 300, F - 1, 1, 1, None) == gpu.MSM_ERR_INVALID
    assert L.msm_tica_create(C.byref(C.c_void_p()), 0, 1, 0) == gpu.MSM_ERR_INVALID
    out = np.zeros(4)
    assert L.msm_dist_f32(dptr, dptr, b"manhattan", 4, 2, None, 0, C.c_void_p(out.ctypes.data), 0) == gpu.MSM_ERR_METRIC
    gpu.check(L.msm_tica_destroy(h))
    gpu.check(L.msm_free(dptr))


def test_host_trajectories_staged_in_overlapped_groups(gpu):
    """Pageable numpy trajectories are staged in groups of 512 MiB through the two halves of a device buffer, group g + 1
    copied while group g is accumulated (tica.hip, tica_accumulate_any).  1.45 GB here = three groups, with trajectories of
    unequal length, one shorter than the lag (skipped, tica.py:404-409 semantics) and one larger than a quarter of a group:
    the accumulators must be the device-resident fit's (same kernels, different launch boundaries: 1e-5 of their scale), and
    a NaN in the LAST group must raise like array2d (utils/validation.py:68-74) does."""
    from msmbuilder_amd import tICA
    rng = np.random.default_rng(5)
    F, lag = 512, 50
    lens = [60_000] * 4 + [30] + [45_000] * 6 + [150_000] + [20_000] * 5
    host = []
    for n in lens:
        z = np.cumsum(rng.standard_normal((n, 6), dtype=np.float32), axis=0) * 0.02
        host.append((z @ rng.standard_normal((6, F), dtype=np.float32) + rng.standard_normal((n, F), dtype=np.float32)).astype(np.float32))
    assert sum(x.nbytes for x in host) > 2.5 * (512 << 20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mh = tICA(n_components=4, lag_time=lag).fit(host)
        md = tICA(n_components=4, lag_time=lag).fit([torch.from_numpy(x).cuda() for x in host])
    assert mh.n_sequences_ == md.n_sequences_ == len(lens) - 1
    assert mh.n_observations_ == md.n_observations_
    for a, b in zip(_accumulators(mh), _accumulators(md)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * np.abs(b).max())   # covariances to 1e-5 of their scale (each launch has its own shift row)
    np.testing.assert_allclose(mh.eigenvalues_, md.eigenvalues_, rtol=1e-6)
    bad = [x for x in host]
    bad[-2] = bad[-2].copy()
    bad[-2][777, 13] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ValueError):
            tICA(n_components=4, lag_time=lag).fit(bad)
        # the library is usable afterwards, and a fit of the clean data gives the same numbers again
        m2 = tICA(n_components=4, lag_time=lag).fit(host)
    np.testing.assert_array_equal(m2.eigenvalues_, mh.eigenvalues_)


def test_transform_of_host_trajectories_in_overlapped_groups(gpu):
    """tICA.transform on a list of numpy trajectories goes down in ONE call (msm_tica_project_host_list: groups of 512 MiB
    through two device buffers, the copy of group g + 1 beside the projection of group g, one copy back): every row must be
    what the per-trajectory path (partial_transform: tica.py:356-377) returns, bit for bit -- ragged lengths, an empty
    trajectory, three groups, float64 rows, a width that is no multiple of 4, and a NaN raises like array2d."""
    from msmbuilder_amd import tICA
    rng = np.random.default_rng(11)
    for F, dtype, lens in ((512, np.float32, [60_000, 1, 45_000, 0, 150_000, 20_000, 33_333, 90_000, 70_000, 41_000, 52_000, 66_000]),
                           (171, np.float32, [5_000, 12_345, 7]), (64, np.float64, [30_000, 2_000])):
        host = [rng.standard_normal((n, F)).astype(dtype) + np.linspace(-1, 1, F, dtype=dtype) for n in lens]
        if F == 512:
            assert sum(x.nbytes for x in host) > 2.2 * (512 << 20)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = tICA(n_components=7, lag_time=3, kinetic_mapping=(F == 171)).fit([x for x in host if len(x) > 3])
            Y = m.transform(host)
            assert len(Y) == len(host)
            for x, y in zip(host, Y):
                assert y.shape == (len(x), 7) and y.dtype == np.float64
                if len(x):
                    np.testing.assert_array_equal(y, m.partial_transform(x))
            bad = list(host)
            bad[-1] = bad[-1].copy()
            bad[-1][-1, 5] = np.inf
            with pytest.raises(ValueError):
                m.transform(bad)
            np.testing.assert_array_equal(m.transform(host)[0], Y[0])     # usable afterwards


def test_fit_transform_of_host_trajectories_uploads_once(gpu):
    """tICA.fit_transform on numpy trajectories stages the list on the device once (msm_upload_list) and runs fit and
    transform there: the result must be exactly what fit + transform of the same rows as device tensors give, numpy arrays
    per trajectory like the reference's mixin (base.py fit_transform), close to the two-pass host route, and a NaN raises."""
    from msmbuilder_amd import tICA
    rng = np.random.default_rng(3)
    F, lag = 256, 7
    lens = [20_000, 3, 15_001, 40_000, 9_999]
    host = []
    for n in lens:
        z = np.cumsum(rng.standard_normal((n, 5)), axis=0) * 0.05
        host.append((np.tanh(z) @ rng.standard_normal((5, F)) + 0.5 * rng.standard_normal((n, F))).astype(np.float32))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m1 = tICA(n_components=4, lag_time=lag)
        Y1 = m1.fit_transform(host)
        dev = [torch.from_numpy(x).cuda() for x in host]
        m2 = tICA(n_components=4, lag_time=lag).fit(dev)
        Y2 = [y.cpu().numpy() for y in m2.transform(dev)]
        m3 = tICA(n_components=4, lag_time=lag).fit(host)
        Y3 = m3.transform(host)
    assert m1.n_sequences_ == m2.n_sequences_ == len(lens) - 1
    assert all(isinstance(y, np.ndarray) and y.dtype == np.float64 and y.shape == (n, 4) for y, n in zip(Y1, lens))
    np.testing.assert_array_equal(m1.eigenvalues_, m2.eigenvalues_)
    for a, b, c in zip(Y1, Y2, Y3):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, c, rtol=0, atol=2e-4 * max(1.0, np.abs(c).max()))
    bad = list(host)
    bad[0] = bad[0].copy()
    bad[0][11, 2] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ValueError):
            tICA(n_components=4, lag_time=lag).fit_transform(bad)
