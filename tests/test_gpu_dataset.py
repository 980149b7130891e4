"""dir-npy datasets (dataset.py:290-331) through the native device loader: the bytes in HBM are the
file payloads, and estimators fitted on the streamed view equal the ones fitted on np.load'ed arrays."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(tmp_path, arrays):
    from msmbuilder_amd.dataset import dataset
    ds = dataset(str(tmp_path / "ds"), mode="w", fmt="dir-npy")
    for i, a in enumerate(arrays):
        ds[i] = a
    return dataset(str(tmp_path / "ds"), mode="r")


def test_container_protocol_and_device_stream(gpu, tmp_path):
    rs = np.random.RandomState(0)
    arrays = [rs.randn(1000, 12).astype(np.float32), rs.randn(1, 12).astype(np.float32),
              rs.randn(70000, 12).astype(np.float32),          # 3.4 MB: many 64 KiB buffer turns
              rs.randn(333, 12).astype(np.float32)]
    ds = _make(tmp_path, arrays)
    assert list(ds.keys()) == [0, 1, 2, 3] and len(ds) == 4
    assert sorted(os.listdir(ds.path)) == ["%08d.npy" % i for i in range(4)] + ["PROVENANCE.txt"]
    for a, b in zip(arrays, ds):
        assert np.array_equal(a, b)
    assert np.array_equal(ds.get(2, mmap=True)[5], arrays[2][5])
    with pytest.raises(IndexError):
        ds.get(17)
    with pytest.raises(IOError):
        ds.set(9, arrays[0])
    view = ds.device_sequences(prefetch=3, buffer_bytes=64 << 10)
    assert len(view) == 4
    for _ in range(2):                                          # re-iterable
        got = list(view)
        assert all(t.is_cuda for t in got)
        for a, t in zip(arrays, got):
            assert tuple(t.shape) == a.shape and np.array_equal(t.cpu().numpy(), a)
    assert np.array_equal(view[2].cpu().numpy(), arrays[2])


def test_other_dtypes_and_errors(gpu, tmp_path):
    rs = np.random.RandomState(1)
    arrays = [rs.randn(50, 3), rs.randint(0, 9, 77).astype(np.int64), rs.randint(0, 9, (5, 2)).astype(np.int32)]
    ds = _make(tmp_path, arrays)
    for a, t in zip(arrays, ds.device_sequences()):
        assert str(t.dtype).endswith(str(a.dtype)) and np.array_equal(t.cpu().numpy(), a)
    np.save(os.path.join(ds.path, "%08d.npy" % 3), np.asfortranarray(rs.randn(4, 5)))
    with pytest.raises(ValueError, match="Fortran"):
        list(ds.device_sequences())
    os.remove(os.path.join(ds.path, "%08d.npy" % 3))
    with open(os.path.join(ds.path, "%08d.npy" % 3), "wb") as f:
        f.write(b"not an npy file at all")
    with pytest.raises(ValueError, match="not a .npy"):
        list(ds.device_sequences())


def test_estimators_on_streamed_dataset(gpu, tmp_path, monkeypatch):
    from msmbuilder_amd import tICA, KCenters
    monkeypatch.setenv("MSMBUILDER_AMD_TICA_MODE", "f64")
    rs = np.random.RandomState(2)
    arrays = [np.cumsum(rs.randn(n, 8), axis=0).astype(np.float32) * 0.05 + rs.randn(n, 8).astype(np.float32)
              for n in (4000, 2500, 3, 6000, 1200)]
    ds = _make(tmp_path, arrays)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = tICA(n_components=3, lag_time=4).fit(arrays)
        m = ds.fit_with(tICA(n_components=3, lag_time=4))              # host arrays through np.load
        md = tICA(n_components=3, lag_time=4).fit(ds.device_sequences())  # streamed into HBM
    assert md.n_observations_ == ref.n_observations_ == m.n_observations_
    np.testing.assert_allclose(m.eigenvalues_, ref.eigenvalues_, rtol=1e-12)
    np.testing.assert_allclose(md.eigenvalues_, ref.eigenvalues_, rtol=1e-12)
    out = ds.transform_with(md, str(tmp_path / "tics"))
    for a, y in zip(arrays, out):
        assert y.shape == (len(a), 3)
        np.testing.assert_allclose(y, ref.partial_transform(a), rtol=1e-9, atol=1e-11)
    kc = KCenters(n_clusters=5, random_state=0).fit(list(out.device_sequences()))
    kr = KCenters(n_clusters=5, random_state=0).fit(list(out))
    assert kc.cluster_ids_ == kr.cluster_ids_
