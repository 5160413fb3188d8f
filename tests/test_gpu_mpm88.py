"""BASELINE configs[0] on the device: the 2D dense-grid demo (mls-mpm88.cpp:16-69) through the C ABI
(mpmhip_mpm88_*), against the CPU oracle's restatement of the same function on identical seeded particles.

Tolerances (fp32; the oracle does its 2x2 polar / SVD in double, the device in float, and the P2G scatter uses
float atomics in arbitrary order): one step 1e-5 relative; 200 steps 1e-3 on positions' spread (the clamp makes the
map contractive, but the trajectories are not compared particle by particle beyond a few dozen steps)."""
import numpy as np
import pytest

from tests.common import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def _seed(n_per, seed=88, stir=0.0):
    rng = np.random.default_rng(seed)
    xs = [(rng.random((n_per, 2)) * 2 - 1) * 0.08 + np.array(c) for c in ((0.55, 0.45), (0.45, 0.65), (0.55, 0.85))]
    x = np.concatenate(xs).astype(np.float32)
    n = len(x)
    v = (rng.normal(0, stir, (n, 2))).astype(np.float32)
    F = (np.tile(np.eye(2).reshape(1, 4), (n, 1)) + rng.normal(0, 0.02 if stir else 0.0, (n, 4))).astype(np.float32)
    Cm = rng.normal(0, stir, (n, 4)).astype(np.float32)
    Jp = (1.0 + rng.normal(0, 0.02 if stir else 0.0, n)).astype(np.float32)
    return x, v, F, Cm, Jp


@pytest.mark.parametrize("plastic", [True, False])
def test_one_step_matches_oracle_on_a_stirred_state(tm, orc, plastic):
    x, v, F, Cm, Jp = _seed(2667, stir=1.0)  # ~8 k particles, the BASELINE wording
    sim = tm.MPM88(plastic=plastic)
    sim.add_particles(x, v, F, Cm, Jp)
    sim.advance(1)
    gx, gv, gF, gC, gJ = sim.particles()
    ggrid = sim.grid()
    rx, rv, rF, rC, rJ = x.copy(), v.copy(), F.copy(), Cm.copy(), Jp.copy()
    rgrid = orc.mpm88_advance(80, 1e-4, rx, rv, rF, rC, rJ, plastic=plastic)
    assert rel_l2(ggrid, rgrid) <= 1e-5
    assert np.abs(gx - rx).max() <= 1e-7
    assert rel_l2(gv, rv) <= 1e-5 and rel_l2(gC, rC) <= 1e-5
    assert rel_l2(gF, rF) <= 1e-5 and rel_l2(gJ, rJ) <= 1e-5
    sim.close()


def test_first_step_from_rest_is_free_fall(tm):
    """undeformed, at rest: zero stress, every node gets v = (0, -200 dt)  (same property as tests/test_oracle_mpm88.py)"""
    sim = tm.MPM88()
    for c in ((0.55, 0.45), (0.45, 0.65), (0.55, 0.85)):
        sim.add_object(c)  # mls-mpm88.cpp:76-77
    assert sim.num_particles() == 3000
    x0 = sim.particles()[0]
    sim.advance(1)
    x, v, F, Cm, Jp = sim.particles()
    assert np.allclose(v[:, 0], 0, atol=1e-6) and np.allclose(v[:, 1], -200 * 1e-4, atol=1e-6)
    assert np.allclose(x, x0 + 1e-4 * v, atol=1e-7)
    assert np.allclose(F.reshape(-1, 2, 2), np.eye(2), atol=1e-5) and np.allclose(Jp, 1, atol=1e-5)
    sim.close()


def test_forty_steps_track_the_oracle_and_1500_stay_bounded(tm, orc):
    x, v, F, Cm, Jp = _seed(500)
    sim = tm.MPM88()
    sim.add_particles(x)
    rx, rv, rF, rC, rJ = x.copy(), v.copy(), F.copy(), Cm.copy(), Jp.copy()
    sim.advance(40)
    for _ in range(40):
        orc.mpm88_advance(80, 1e-4, rx, rv, rF, rC, rJ)
    gx, gv, gF, gC, gJ = sim.particles()
    assert np.abs(gx - rx).max() <= 1e-5 and rel_l2(gv, rv) <= 1e-3 and rel_l2(gF, rF) <= 1e-4
    sim.advance(1460)
    gx, gv, gF, gC, gJ = sim.particles()
    assert np.all(np.isfinite(gx)) and np.all(np.isfinite(gv)) and np.all(np.isfinite(gF))
    assert gx.min() > 0.03 and gx.max() < 0.97                            # box walls at 0.05 / 0.95
    assert gJ.min() >= 0.6 - 1e-6 and gJ.max() <= 20 + 1e-6               # mls-mpm88.cpp:65
    s = np.linalg.svd(gF.reshape(-1, 2, 2), compute_uv=False)
    assert s.min() >= 0.975 - 1e-4 and s.max() <= 1.0075 + 1e-4           # :62-63
    assert gx[:, 1].mean() < 0.62                                         # it fell
    sim.close()


def test_empty_and_growing_particle_sets(tm):
    sim = tm.MPM88(n=40, dt=2e-4)
    sim.advance(3)  # no particles: only the grid kernels run
    assert sim.num_particles() == 0 and not sim.grid().any()
    sim.add_object((0.5, 0.5), count=10)
    sim.add_object((0.4, 0.6), count=700)  # forces the arrays to grow
    sim.advance(5)
    x, v, F, Cm, Jp = sim.particles()
    assert len(x) == 710 and np.all(np.isfinite(x)) and (v[:, 1] < 0).all()
    sim.close()


# ------------------------------------------------------------------------------------------ pinned to the reference
# tests/golden/ref_mpm88.npz = output of /root/reference/mls-mpm88.cpp's own advance() compiled in place
# (tests/golden/make_mpm88_golden.py).  Tolerances (fp32, float atomics in arbitrary order for the scatter, 2x2 polar / SVD
# in float on the device against the shim's double): one step 1e-5 relative, x 1.2e-7 absolute; 40 steps x 1e-5.
import os

GOLD88 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mpm88.npz")


@pytest.mark.parametrize("name,plastic", [("stir_plastic", True), ("stir_elastic", False)])
def test_one_step_matches_the_reference_file(tm, name, plastic):
    g = np.load(GOLD88)
    sim = tm.MPM88(plastic=plastic)
    sim.add_particles(*[g["%s_in_%s" % (name, k)] for k in "xvFCJ"])
    sim.advance(1)
    got = sim.particles()
    want = [g["%s_out_%s" % (name, k)] for k in "xvFCJ"]
    assert rel_l2(sim.grid().reshape(81, 81, 3), g[name + "_grid"]) <= 1e-5
    assert np.abs(got[0] - want[0]).max() <= 1.2e-7
    for a, b in zip(got[1:], want[1:]):
        assert rel_l2(a, b) <= 1e-5
    sim.close()


def test_forty_steps_track_the_reference_file(tm):
    g = np.load(GOLD88)
    sim = tm.MPM88()
    sim.add_particles(g["fall_in_x"])
    sim.advance(40)
    gx, gv, gF, gC, gJ = sim.particles()
    assert np.abs(gx - g["fall_out_x"]).max() <= 1e-5
    assert rel_l2(gv, g["fall_out_v"]) <= 1e-3 and rel_l2(gF, g["fall_out_F"]) <= 1e-4 and rel_l2(gJ, g["fall_out_J"]) <= 1e-4
    sim.close()


def test_live_reference_file_next_to_the_device_on_a_fresh_scene(tm):
    """the compiled mls-mpm88.cpp itself (oracle/_ref/libmpm_ref.so travels to the GPU box) on a scene no fixture holds"""
    from oracle import refmpm as ref
    if not ref.available() or not ref.mpm88_available():
        pytest.skip("oracle/_ref/libmpm_ref.so (with mls-mpm88.cpp) not built")
    x, v, F, Cm, Jp = _seed(1500, seed=4242, stir=0.7)
    sim = tm.MPM88()
    sim.add_particles(x, v, F, Cm, Jp)
    sim.advance(3)
    got = sim.particles()
    r = [a.copy() for a in (x, v, F, Cm, Jp)]
    ref.mpm88_advance(*r, steps=3, plastic=True)
    assert np.abs(got[0] - r[0]).max() <= 3e-7
    for a, b in zip(got[1:], r[1:]):
        assert rel_l2(a, b) <= 3e-5
    sim.close()
