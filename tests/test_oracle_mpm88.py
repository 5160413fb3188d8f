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


# ------------------------------------------------------------------------------------------ pinned to the reference
# tests/golden/ref_mpm88.npz is OUTPUT OF /root/reference/mls-mpm88.cpp's own advance() compiled in place
# (tests/golden/make_mpm88_golden.py, oracle/ref_mpm88_driver.cpp).  Tolerances, fp32: the restated oracle sums the 9
# scatter terms of a node in particle order like the reference but forms the stress in a different association ->
# 1e-5 relative on one step; 40 steps 1e-4 on positions.
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mpm88.npz")


def _case(g, name):
    return [g["%s_in_%s" % (name, k)].copy() for k in "xvFCJ"], [g["%s_out_%s" % (name, k)] for k in "xvFCJ"]


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-30))


def test_restated_advance_matches_the_reference_file_on_one_step(orc):
    g = np.load(GOLD)
    for name, plastic in (("stir_plastic", True), ("stir_elastic", False)):
        s, want = _case(g, name)
        grid = orc.mpm88_advance(80, 1e-4, *s, plastic=plastic)
        assert _rel(grid.reshape(81, 81, 3), g[name + "_grid"]) <= 1e-5
        assert np.abs(s[0] - want[0]).max() <= 1.2e-7
        for got, ref in zip(s[1:], want[1:]):
            assert _rel(got, ref) <= 1e-5, name


def test_restated_advance_tracks_the_reference_file_over_forty_steps(orc):
    g = np.load(GOLD)
    x = g["fall_in_x"].copy()
    n = len(x)
    v = np.zeros((n, 2), np.float32); F = np.tile(np.eye(2, dtype=np.float32).reshape(1, 4), (n, 1))
    C = np.zeros((n, 4), np.float32); Jp = np.ones(n, np.float32)
    for _ in range(40):
        orc.mpm88_advance(80, 1e-4, x, v, F, C, Jp)
    assert np.abs(x - g["fall_out_x"]).max() <= 2e-6
    assert _rel(v, g["fall_out_v"]) <= 1e-4 and _rel(F, g["fall_out_F"]) <= 1e-5 and _rel(Jp, g["fall_out_J"]) <= 1e-5


def test_live_reference_file_reproduces_its_fixture():
    from oracle import refmpm as ref
    if not ref.available() or not ref.mpm88_available():
        import pytest
        pytest.skip("oracle/_ref/libmpm_ref.so (with mls-mpm88.cpp) not built")
    g = np.load(GOLD)
    s, want = _case(g, "stir_plastic")
    grid = ref.mpm88_advance(*s, steps=1, plastic=True)
    assert np.array_equal(grid, g["stir_plastic_grid"])
    for got, w in zip(s, want):
        assert np.array_equal(got, w)
