"""Round 6: k-centers passes screened on the FEATURE-major byte copy (csrc/distance_wscreen_dev.h) --
float32 rows of any length, float64 rows of more than 16 features.  Centre ids, labels_ and distances_ must be the reference
scan's bit for bit (the C oracle = kcenters.py:79-102 over libdistance.dist, pinned against the compiled reference), with the
adversarial inputs of tests/test_gpu_fullsize.py::test_kcenters_float32_screened_passes: duplicate rows (float32-image ties in
the argmax), a large common offset, values beyond the float32 range and a NaN (the screen must switch itself off / never
assign), rows on a shell around the copy's origin (the margin is tight for every row), anisotropic scales, a lattice (masses of
exactly equal distances), tiny and huge scales; and the same fit with the screen switched off (MSM_KC_WSCREEN=0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rs, n, m, case, dtype):
    Y = rs.randn(n, m)
    Y[2000:2030] = Y[11]
    Y[n - 5000:n - 4996] = Y[n // 2]
    if case == "offset":
        Y += 3.0e3 if dtype == np.float32 else 3.0e6
    elif case == "beyond_float32" and dtype == np.float64:
        Y[777] = 1.0e39
    elif case == "nan_row":
        Y[4242, m // 2] = np.nan
    elif case == "shell":
        Y = Y / np.linalg.norm(Y, axis=1, keepdims=True) * 7.0 + 0.01 * rs.randn(n, m)
        Y[2000:2030] = Y[11]
    elif case == "anisotropic":
        Y *= np.linspace(3.0, 0.3, m)
    elif case == "lattice":
        Y = np.round(Y * 2.0) / 2.0
    elif case == "tiny":
        Y *= 1e-20
    elif case == "huge":
        Y *= 1e18
    return np.ascontiguousarray(Y.astype(dtype))


@pytest.mark.parametrize("case", ["plain", "offset", "beyond_float32", "nan_row", "shell", "anisotropic", "lattice", "tiny", "huge"])
@pytest.mark.parametrize("dtype,m", [(np.float32, 3), (np.float32, 10), (np.float32, 33), (np.float32, 171), (np.float64, 17), (np.float64, 40)])
def test_wide_screened_passes_bit_exact(gpu, case, dtype, m):
    from msmbuilder_amd import KCenters
    from oracle.libdistance_oracle import Oracle
    o = Oracle()
    rs = np.random.RandomState(m + len(case))
    n, k = 70_001, 40
    Y = _case(rs, n, m, case, dtype)
    m_ = KCenters(n_clusters=k, random_state=2).fit([Y[:30_000], Y[30_000:]])
    ids, labels, dist = o.kcenters_fit(Y, k, "euclidean", m_.cluster_ids_[0])
    assert m_.cluster_ids_ == list(ids)
    assert np.array_equal(np.concatenate(m_.labels_), labels)
    assert np.array_equal(np.concatenate(m_.distances_), dist)


@pytest.mark.parametrize("dtype,n,m,k", [(np.float32, 280_000, 171, 60), (np.float32, 200_000, 64, 100), (np.float64, 150_000, 24, 80)])
def test_wide_screen_equals_plain_passes_device_rows(gpu, monkeypatch, dtype, n, m, k):
    """A/B on device-resident rows at sizes the oracle would take minutes for: identical ids / labels / distances / inertia
    with the screen off, and the screen really ran (msm_kcenters_last_stats counts its passes)."""
    import ctypes as C
    import torch
    from msmbuilder_amd import KCenters, _lib
    g = torch.Generator(device="cuda").manual_seed(m)
    hubs = torch.randn(12, m, generator=g, device="cuda") * 2.0
    X = (hubs[torch.randint(0, 12, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")).to(
        torch.float32 if dtype == np.float32 else torch.float64).contiguous()
    out = {}
    monkeypatch.setenv("MSM_KC_WBATCH", "0")        # one centre per screened pass (the batched passes: the test below)
    for sw in ("1", "0"):
        monkeypatch.setenv("MSM_KC_WSCREEN", sw)
        kc = KCenters(n_clusters=k, random_state=1).fit([X])
        st = (C.c_int64 * 5)()
        _lib.check(_lib.lib().msm_kcenters_last_stats(st))
        out[sw] = (list(kc.cluster_ids_), kc.labels_[0].cpu().numpy(), kc.distances_[0].cpu().numpy(), kc.inertia_, list(st))
    a, b = out["1"], out["0"]
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    assert a[4][2] == k - 4 and b[4][2] == 0        # screened passes: all but the four plain ones / none


@pytest.mark.parametrize("dtype,n,m,k,data", [(np.float32, 280_000, 171, 60, "hubs"), (np.float32, 200_000, 64, 100, "hubs"),
                                              (np.float64, 150_000, 24, 80, "hubs"), (np.float32, 120_000, 512, 40, "hubs"),
                                              (np.float64, 100_000, 100, 50, "blob"), (np.float32, 90_000, 33, 300, "ties"),
                                              (np.float32, 70_000, 300, 30, "nan"), (np.float32, 70_000, 40, 12, "const"),
                                              (np.float64, 66_000, 300, 24, "hubs"), (np.float32, 66_000, 2048, 12, "hubs")])
def test_wide_batched_passes_equal_plain_passes(gpu, monkeypatch, dtype, n, m, k, data):
    """Several centres per screened pass of wide rows (distance_wbatch_dev.h: threshold lists, the selector replays the
    algorithm on the listed rows, the pass applies the batch in order): centre ids / labels / distances / inertia of the
    one-centre-per-pass loop, bit for bit, in fewer passes.  16 centres per pass (rows of <= 256 float32 features) and 8
    (512 features); one blob (the list is always full), duplicated rows (exact ties: the lowest row wins), a NaN row."""
    import ctypes as C
    import torch
    from msmbuilder_amd import KCenters, _lib
    g = torch.Generator(device="cuda").manual_seed(m + k)
    td = torch.float32 if dtype == np.float32 else torch.float64
    if data == "blob":
        X = torch.randn(n, m, generator=g, device="cuda").to(td)
    else:
        hubs = torch.randn(12, m, generator=g, device="cuda") * 2.0
        X = (hubs[torch.randint(0, 12, (n,), generator=g, device="cuda")] + torch.randn(n, m, generator=g, device="cuda")).to(td)
    if data == "const":
        X[:] = X[0]                             # every distance is 0: no usable threshold, one centre per pass
    if data == "ties":
        X[n // 2:] = X[: n - n // 2]            # every row twice: every maximum is tied
    if data == "nan":
        X[12345, 7] = float("nan")
    X = X.contiguous()
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("MSM_KC_WBATCH", sw)
        monkeypatch.setenv("MSM_KC_WSCREEN", sw)
        kc = KCenters(n_clusters=k, random_state=1).fit([X])
        st = (C.c_int64 * 5)()
        _lib.check(_lib.lib().msm_kcenters_last_stats(st))
        out[sw] = (list(kc.cluster_ids_), kc.labels_[0].cpu().numpy(), kc.distances_[0].cpu().numpy(), kc.inertia_, list(st))
    a, b = out["1"], out["0"]
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    assert a[3] == b[3] or (np.isnan(a[3]) and np.isnan(b[3]))
    assert 0 < a[4][2] <= k - 4 and b[4][2] == 0    # batched passes: fewer than centres ...
    assert data in ("nan", "const") or a[4][2] < (k - 4) // 2 or m == 2048   # (2,048 features: rows too long for the batches' LDS: one centre per pass)   # ... (a NaN distance makes every list unusable: one centre per pass, still exact)
