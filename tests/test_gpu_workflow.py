"""The reference's canonical pipeline (tests/workflows/basic.sh: RobustScaler -> tICA(n_components=4,
shrinkage=0, kinetic_mapping, lag_time=2) -> KCenters(metric=cityblock) -> transition counts) end to end
on the device, fed from a dir-npy dataset, every stage checked against its CPU oracle."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle.libdistance_oracle import Oracle
    return Oracle()


def _features(seed, n_traj=6, F=15):
    rs = np.random.RandomState(seed)
    k = 3
    M = rs.randn(k, F)
    a = np.exp(-1.0 / np.array([60.0, 20.0, 7.0]))
    scale = rs.uniform(0.05, 30.0, F)
    out = []
    for n in rs.randint(1500, 4000, size=n_traj):
        z = np.zeros((n, k))
        e = rs.randn(n, k)
        for t in range(1, n):
            z[t] = a * z[t - 1] + np.sqrt(1 - a * a) * e[t]
        out.append(((z.dot(M) + 0.4 * rs.randn(n, F)) * scale + 3.0).astype(np.float32))
    return out


def test_basic_workflow(gpu, oracle, tmp_path, monkeypatch):
    from sklearn.preprocessing import RobustScaler as RefScaler
    from msmbuilder_amd import tICA, KCenters
    from msmbuilder_amd.dataset import dataset
    from msmbuilder_amd.msm import _transition_counts
    from msmbuilder_amd.preprocessing import RobustScaler
    from oracle.tica_oracle import TicaOracle
    from oracle.transition_oracle import transition_counts
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    feats = _features(0)
    ds = dataset(str(tmp_path / "atom_pairs"), mode="w", fmt="dir-npy")
    for i, x in enumerate(feats):
        ds[i] = x
    ds = dataset(str(tmp_path / "atom_pairs"))
    warnings.simplefilter("ignore")

    # msmb RobustScaler -i atom_pairs/ -t scaled_atom_pairs
    scaler = ds.fit_with(RobustScaler())
    ref_scaler = RefScaler().fit(np.concatenate(feats))
    assert np.array_equal(scaler.center_, ref_scaler.center_) and np.array_equal(scaler.scale_, ref_scaler.scale_)
    scaled = ds.transform_with(scaler, str(tmp_path / "scaled_atom_pairs"))
    for x, y in zip(feats, scaled):
        assert np.array_equal(y, ref_scaler.transform(x))

    # msmb tICA --n_components 4 --shrinkage 0 --kinetic_mapping --lag_time 2
    kw = dict(n_components=4, shrinkage=0, kinetic_mapping=True, lag_time=2)
    tica = tICA(**kw).fit(scaled.device_sequences())          # streamed from the dir-npy files into HBM
    otica = TicaOracle(**kw).fit(list(scaled))
    np.testing.assert_allclose(tica.eigenvalues_, otica.eigenvalues_, rtol=1e-10)
    np.testing.assert_allclose(tica.timescales_, otica.timescales_, rtol=1e-8)
    tics = scaled.transform_with(tica, str(tmp_path / "atom_pairs_tica"))
    otics = otica.transform(list(scaled))
    for y, yo in zip(tics, otics):
        s = np.sign(np.sum(y * yo, axis=0))
        np.testing.assert_allclose(y * s, yo, rtol=1e-6, atol=1e-9)

    # msmb KCenters --metric cityblock   (default n_clusters = 8), on the projection the device produced
    seqs = list(tics)
    kc = KCenters(metric="cityblock", random_state=0).fit(seqs)
    X = np.concatenate(seqs)
    ids, labels, dist = oracle.kcenters_fit(X, 8, "cityblock", kc.cluster_ids_[0])
    assert kc.cluster_ids_ == list(ids)
    assert np.array_equal(np.concatenate(kc.labels_), labels)
    assert np.array_equal(np.concatenate(kc.distances_), dist)

    # msmb MarkovStateModel: the counting step, on device-resident labels
    import torch
    dev_labels = [torch.from_numpy(np.ascontiguousarray(l)).cuda() for l in kc.labels_]
    counts, mapping = _transition_counts(dev_labels, lag_time=1)
    ocounts, omapping = transition_counts([np.asarray(l) for l in kc.labels_], lag_time=1)
    assert np.array_equal(counts, ocounts) and list(mapping.items()) == list(omapping.items())
    assert counts.sum() == sum(len(l) - 1 for l in kc.labels_)


@pytest.mark.parametrize("where", ["device", "host"])
def test_transform_and_predict_over_adjacent_views_equal_the_per_trajectory_calls(gpu, where):
    """Trajectories that are views of ONE allocation (`X.view(n, T, F).unbind(0)`, slices of a joined numpy array) are
    projected / labelled in one launch (`_lib.adjacent_view`); the per-trajectory results must be the ones separate
    allocations get, ragged lengths and an empty trajectory included."""
    import torch
    from msmbuilder_amd import tICA, KCenters, MiniBatchKMeans
    rs = np.random.RandomState(5)
    lens = [700, 1300, 0, 64, 2001]
    X = (rs.randn(sum(lens), 24) + rs.randn(24)).astype(np.float32)
    big = torch.from_numpy(X).cuda() if where == "device" else X
    bounds = np.concatenate(([0], np.cumsum(lens)))
    views = [big[bounds[i]:bounds[i + 1]] for i in range(len(lens))]
    copies = [v.clone() if where == "device" else v.copy() for v in views]
    fit_on = [v for v in views if len(v)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tica = tICA(n_components=3, lag_time=5).fit(fit_on)
        ya, yb = tica.transform(views), tica.transform(copies)
    as_np = lambda t: t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
    assert [len(y) for y in ya] == lens
    for a, b in zip(ya, yb):
        assert a.shape == b.shape and np.array_equal(as_np(a), as_np(b))
    for est in (KCenters(n_clusters=12, random_state=0), MiniBatchKMeans(n_clusters=12, random_state=0, n_init=1)):
        Yv = [y for y in ya]
        est.fit([y for y in Yv if len(y)])
        la, lb = est.predict(Yv), est.predict([y.clone() if hasattr(y, "clone") else y.copy() for y in Yv])
        assert [len(l) for l in la] == lens
        for a, b in zip(la, lb):
            assert np.array_equal(as_np(a), as_np(b))


@pytest.mark.parametrize("dtype,F,k", [("float32", 64, 3), ("float64", 30, 20), ("bfloat16", 40, 5), ("float32", 36, 17),
                                       ("float32", 30, 4)])   # (30 float32 features: rows are no whole 16-byte vectors -- one by one)
def test_transform_of_separately_allocated_device_trajectories_is_one_batched_launch(gpu, dtype, F, k):
    """`tICA.transform` on a list of separately allocated device trajectories goes through `msm_tica_project_batch` (a table
    of 256-row tiles, one launch per 16 components); every trajectory's result must equal `partial_transform` of it alone
    -- lengths that are not multiples of the tile, one row, an empty trajectory, more than 16 components, bfloat16 rows."""
    import torch
    from msmbuilder_amd import tICA
    rs = np.random.RandomState(F + k)
    tdt = getattr(torch, dtype)
    lens = [257, 1, 0, 1024, 300, 255, 2049]
    seqs = [torch.from_numpy((rs.randn(n, F) + 0.5).astype(np.float32)).cuda().to(tdt) for n in lens]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = tICA(n_components=k, lag_time=3).fit([s.float() for s in seqs if len(s) > 10])
        ys = m.transform(seqs)
        one = [m.partial_transform(s) for s in seqs]
    assert [tuple(y.shape) for y in ys] == [(n, k) for n in lens]
    for a, b in zip(ys, one):
        assert a.dtype == torch.float64 and np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    bad = [s.clone() for s in seqs]
    bad[4][7, 3] = float("nan")
    with pytest.raises(ValueError, match="NaN"):
        m.transform(bad)

