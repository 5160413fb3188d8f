"""AsyncMPM, first half, against the reference's own async stepper compiled from src/async/async_mpm.cpp
(oracle/_ref/libmpm_ref.so): per-material get_allowed_dt, the per-block strength / CFL limits and the power-of-two
continuous_dt_limit of AsyncMPM<dim>::update_dt_limits (src/async/async_mpm.cpp:90-164), and the `limit` attribute of
the frame output (src/async/async_visualize.cpp:17-26)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.common import lattice_cube, make_state

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
MATS = ["jelly", "snow", "sand", "water", "linear", "elastic", "von_mises", "visco"]


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


@pytest.mark.parametrize("mat", MATS)
def test_allowed_dt_matches_the_reference_per_material(tm, mat):
    """mpmhip_debug_allowed_dt (the expression mpmhip_async_update_dt_limits reduces per block) against
    MPMParticle::get_allowed_dt of every registered type (fixture tests/golden/ref_materials.npz)"""
    g = np.load(os.path.join(HERE, "golden", "ref_materials.npz"))
    gp, t = np.ascontiguousarray(g[mat + "_gp"], np.float32), int(g[mat + "_type"])
    F, aux, v = (np.ascontiguousarray(g[mat + k], np.float32) for k in ("_F", "_aux", "_v"))
    n = len(F)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(32,) * 3))
    sim.add_particles(dict(type="jelly", positions=lattice_cube(32, 10, 12, 1 / 32)))
    sim._ensure_ctx()
    fp = C.POINTER(C.c_float)
    out = np.zeros(n, np.float32)
    sim._check(sim._L.mpmhip_debug_allowed_dt(sim._ctx, t, gp.ctypes.data_as(fp), n, F.ctypes.data_as(fp), aux.ctypes.data_as(fp),
                                              v.ctypes.data_as(fp), C.c_float(float(g["dx"])), out.ctypes.data_as(fp)))
    want = g[mat + "_allowed_dt"]
    ok = np.isfinite(want)
    assert ok.mean() > 0.9
    assert np.allclose(out[ok], want[ok], rtol=2e-5, atol=0)
    if mat in ("linear", "jelly"):
        assert np.all(want == 0)
    sim.close()


def _two_stiffness_scene():
    res = 32
    dx = 1.0 / res
    xa = lattice_cube(res, 9, 15, dx, jitter=0.2, seed=1)
    xb = lattice_cube(res, 15, 21, dx, jitter=0.2, seed=2)
    sa = make_state(xa, "elastic", dx, perturb_F=0.01, vel_scale=0.5)  # soft: E = 5e3
    sb = make_state(xb, "sand", dx, perturb_F=0.01, vel_scale=0.5)     # stiff
    return res, dx, sa, sb


def test_block_dt_limits_match_the_reference_async_stepper_on_a_two_stiffness_scene(tm, tmp_path):
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)
    res, dx, sa, sb = _two_stiffness_scene()
    r = ref.AsyncSim(res, dx, unit_delta_t=1e-6, max_units=8192)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, keep_apic_b=True))
    for s, mat in ((sa, "elastic"), (sb, "sand")):
        r.add_particles(mat, s.gparams[0][0], s.gparams[0][1], s.x, s.v, s.F, s.B, s.aux)
        sim.add_particles(dict(type=mat, positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
    r.update_dt_limits()
    want, mm_ref = r.blocks()
    sim.enable_async(unit_delta_t=1e-6, max_units=8192)
    sim.update_dt_limits()
    got, mm = sim.async_blocks()
    key = lambda b: [tuple(c) for c in b["coord"]]  # noqa: E731
    assert sorted(key(got)) == sorted(key(want))
    o_g = np.lexsort(got["coord"].T[::-1]); o_w = np.lexsort(want["coord"].T[::-1])
    for k in ("count", "continuous"):
        assert np.array_equal(got[k][o_g], want[k][o_w]), k
    assert len(np.unique(want["continuous"])) >= 2  # the scene really has two step sizes
    assert np.abs(got["strength"][o_g] - want["strength"][o_w]).max() <= 1  # int(...) of fp32 expressions: one unit
    assert np.allclose(got["cfl"][o_g], want["cfl"][o_w], rtol=1e-5, atol=1)
    assert mm == tuple(int(v) for v in mm_ref)
    # frame output: every particle's limit attribute = its block's (continuous, strength, cfl)
    from tests.bgeo_reader import parse
    f = parse(sim.bgeo_bytes(verbose=False))["data"]
    d = r.download()
    lim = np.asarray(f["limit"]).reshape(-1, 3)
    assert np.array_equal(np.asarray(f["index"]).reshape(-1), d["id"])
    assert np.array_equal(lim[:, 0], d["limits"][:, 0])
    assert np.abs(lim[:, 1] - d["limits"][:, 1]).max() <= 1 and np.allclose(lim[:, 2], d["limits"][:, 2], rtol=1e-5, atol=1)
    sim.close(); r.close()


def test_async_stepping_matches_the_reference_async_stepper(tm):
    """create_simulation3('async_mpm').step(dt): blocks advancing with their own power-of-two multiples of unit_delta_t
    (AsyncMPM<dim>::step / advance, src/async/async_mpm.cpp:255-421) against the reference's own stepper on the
    two-stiffness scene: same pools (every container of every block, at its block's time), same number of particle
    updates, particle states to fp32 tolerance"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)
    res, dx, sa, sb = _two_stiffness_scene()
    kw = dict(unit_delta_t=2e-6, max_units=1024)  # sand blocks step with 256 units, the soft elastic ones with 1024
    r = ref.AsyncSim(res, dx, shapes=[(0, 0, 0, 1, 0, -0.2)], friction=0.4, **kw)
    sim = tm.create_simulation3("async_mpm").initialize(dict(res=(res,) * 3, delta_x=dx, **kw))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    for s, mat in ((sa, "elastic"), (sb, "sand")):
        r.add_particles(mat, s.gparams[0][0], s.gparams[0][1], s.x, s.v, s.F, s.B, s.aux)
        sim.add_particles(dict(type=mat, positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
    for _ in range(2):
        r.step(2.5e-3)
        sim.step(2.5e-3)
    assert sim.current_t_int == r.time_int()
    assert sim.update_counter == r.update_counter()
    a, b = sim.get_pool_particles(), r.download()
    assert np.array_equal(a["id"], b["id"])
    assert len(np.unique(b["limits"][:, 0])) >= 2, "the scene must step with at least two block step sizes"
    assert np.array_equal(a["continuous"], b["limits"][:, 0]) and np.array_equal(a["particle_t"], b["limits"][:, 3])
    from tests.common import rel_l2
    assert np.abs(a["x"] - b["x"]).max() <= 1e-6
    # the stiff blocks have taken 12 substeps, their frozen neighbours were re-advanced as often: rounding differences have
    # had a dozen stiff steps to grow (the synchronous multi-step tests allow 1e-3 after 5)
    assert rel_l2(a["v"], b["v"]) <= 5e-4 and rel_l2(a["F"], b["F"]) <= 1e-4
    sim.close(); r.close()
