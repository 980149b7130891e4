"""Inputs of golden fixtures that are regenerated from seeds (by tests/golden/make_golden.py when it captures the expected
outputs, and by the tests that compare against them) instead of being stored."""
import numpy as np


def mbkm_f64_data(which):
    """The float64 inputs of mbkm_f64_golden.npz, regenerated from seeds by the generator and by the tests alike (the arrays
    themselves would be megabytes): 'small' = 20,000 x 16 around 25 centres; 'proj' = 100,000 x 10 with tICA-like column
    scales (what msmbuilder hands MiniBatchKMeans: the float64 output of tICA.transform, tica.py:329-352)."""
    if which == "small":
        rs = np.random.RandomState(101)
        cent = rs.randn(25, 16) * 4.0
        X = cent[rs.randint(0, 25, 20000)] + rs.randn(20000, 16)
        init = X[rs.choice(20000, 25, replace=False)].copy()
        return np.ascontiguousarray(X), init
    rs = np.random.RandomState(202)
    hubs = rs.randn(40, 10) * np.linspace(3.0, 0.4, 10)
    X = hubs[rs.randint(0, 40, 100000)] + rs.randn(100000, 10) * np.linspace(0.8, 0.3, 10)
    return np.ascontiguousarray(X), None
