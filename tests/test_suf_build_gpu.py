"""NeRegSuf(X, y) on the device (RegressionModel.cpp:309-328): X'X by the f64-MFMA syrk in its
XCD-aware tile order (suf_kernel.hip: supertiles of the lower block triangle dealt to the
eight L2s; the supertile's edge depends on how many 64-column tiles there are), X'y, y'y
and the sums -- against numpy at shapes that exercise every edge (1, 2, 4, 8), tile counts
that are no multiple of it, row counts that are no multiple of a panel, and the split over
row slices.  Tolerance: f64 sums in another order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p", [(77, 5), (1000, 64), (333, 65), (5000, 130), (2000, 520),
                                 (1500, 1100), (900, 2100), (600, 4200)])
def test_sufficient_statistics_match_numpy(n, p):
    import boom_amd
    rng = np.random.Generator(np.random.PCG64(n + p))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    y = X[:, : min(p, 4)] @ np.arange(1.0, min(p, 4) + 1.0) + rng.standard_normal(n)
    eng = boom_amd.Engine(2, seed=1)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    xtx = X.T @ X
    scale = np.sqrt(np.outer(np.diag(xtx), np.diag(xtx)))
    assert np.max(np.abs(s["xtx"] - xtx) / scale) < 1e-13
    assert np.array_equal(s["xtx"], s["xtx"].T)          # mirrored on store: exactly symmetric
    assert np.max(np.abs(s["xty"] - X.T @ y)) < 1e-10 * np.abs(X.T @ y).max()
    assert abs(s["yty"] - y @ y) < 1e-12 * (y @ y)
    assert s["n"] == n
    assert abs(s["ybar"] - y.mean()) < 1e-12 * max(1.0, abs(y.mean()))
    assert np.max(np.abs(s["xbar"] - X.mean(0))) < 1e-12
    # the same call again: bitwise the same statistics (fixed summation order)
    eng.build_suf_from_xy(X, y)
    s2 = eng.get_suf()
    assert np.array_equal(s["xtx"], s2["xtx"]) and np.array_equal(s["xty"], s2["xty"])
