"""CPU tier: the product's host-side restatement of scikit-learn's KDTree tie order (csrc/knn_r1.cpp through the C ABI,
mimrl_knn_r1_host) against scikit-learn itself -- the third-party dependency the reference calls in prod_knn_sample
(Model.py:82-86) -- on discrete (MOSI-style, 0.2-step) and continuous 1-column label banks."""
import numpy as np
import pytest

from mimrl_amd import _lib

NearestNeighbors = pytest.importorskip("sklearn.neighbors").NearestNeighbors


@pytest.mark.parametrize("N,m", [(40, 4), (163, 16), (1284, 64), (1284, 16), (700, 33), (16326, 128)])
@pytest.mark.parametrize("k", [2, 3, 5])
@pytest.mark.parametrize("discrete", [True, False])
def test_host_knn_matches_sklearn_tie_order(N, m, k, discrete):
    for seed in range(3):
        g = np.random.default_rng(seed * 100 + N + k)
        z = g.uniform(-3, 3, size=N).astype(np.float32)
        if discrete:
            z = (np.round(z * 5) / 5).astype(np.float32)
        anchors = g.choice(N, size=m, replace=False)
        keep = np.ones(N, bool)
        keep[anchors] = False
        cand = np.nonzero(keep)[0]
        if k >= len(cand) // 2:
            assert _lib.knn_r1_host(z, anchors, k) is None      # scikit-learn's brute-force regime: not restated, caller falls back
            continue
        nn = NearestNeighbors(n_neighbors=k, radius=1.0, metric="euclidean").fit(z[cand].reshape(-1, 1))   # Model.py:82-85
        ref = cand[nn.kneighbors(z[anchors].reshape(-1, 1), return_distance=False)]
        got = _lib.knn_r1_host(z, anchors, k)
        np.testing.assert_array_equal(got, ref)
        if discrete:   # the point of the exercise: ties are everywhere, and the lower-index rule would differ
            d = np.abs(z[cand][None, :] - z[anchors][:, None])
            assert N < 100 or (np.sort(d, 1)[:, k - 1] == np.sort(d, 1)[:, k]).mean() > 0.5
