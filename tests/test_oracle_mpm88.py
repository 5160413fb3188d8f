"""Config C1: the 2D 88-line demo (mls-mpm88.cpp:16-69) — CPU oracle only ("plumbing, no GPU")."""
import numpy as np


def _seed(n_per=1000, seed=88):
    rng = np.random.default_rng(seed)
    xs = []
    for c in ((0.55, 0.45), (0.45, 0.65), (0.55, 0.85)):  # mls-mpm88.cpp:76-77
        xs.append((rng.random((n_per, 2)) * 2 - 1) * 0.08 + np.array(c))
    x = np.concatenate(xs).astype(np.float32)
    n = len(x)
    return x, np.zeros((n, 2), np.float32), np.tile(np.eye(2, dtype=np.float32).reshape(1, 4), (n, 1)), \
        np.zeros((n, 4), np.float32), np.ones(n, np.float32)


def test_mpm88_first_step_is_free_fall(orc):
    x, v, F, C, Jp = _seed()
    x0 = x.copy()
    grid = orc.mpm88_advance(80, 1e-4, x, v, F, C, Jp)
    # undeformed, at rest: stress = 0, every node gets v = (0, -200 dt)
    assert np.allclose(v[:, 0], 0, atol=1e-6) and np.allclose(v[:, 1], -200 * 1e-4, atol=1e-6)
    assert np.allclose(x, x0 + 1e-4 * v, atol=1e-7)
    assert np.allclose(F.reshape(-1, 2, 2), np.eye(2), atol=1e-5) and np.allclose(Jp, 1, atol=1e-5)


def test_mpm88_runs_and_stays_bounded(orc):
    x, v, F, C, Jp = _seed(300)
    for _ in range(1500):
        orc.mpm88_advance(80, 1e-4, x, v, F, C, Jp)
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(v)) and np.all(np.isfinite(F))
    assert x.min() > 0.03 and x.max() < 0.97                        # box walls at 0.05 / 0.95
    assert Jp.min() >= 0.6 - 1e-6 and Jp.max() <= 20 + 1e-6           # mls-mpm88.cpp:65
    s = np.linalg.svd(F.reshape(-1, 2, 2), compute_uv=False)
    assert s.min() >= 0.975 - 1e-4 and s.max() <= 1.0075 + 1e-4       # mls-mpm88.cpp:62-63
    assert x[:, 1].mean() < 0.62                                      # it fell
