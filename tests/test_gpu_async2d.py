"""AsyncMPM<2> on the device (create_simulation2('async_mpm'); include/mpmhip.h "AsyncMPM<2>", csrc/async2d_api.h, csrc/k_async2d.h)
against the REFERENCE's own 2D asynchronous stepper (TC_IMPLEMENTATION(Simulation2D, AsyncMPM2D, "async_mpm"),
src/async/async_mpm.cpp:423-427, compiled in place into oracle/_ref/libmpm_ref.so)."""
import ctypes as C

import numpy as np
import pytest

from tests.common import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def _two_stiffness_scene_2d(tm, res=128):
    """a soft elastic square next to a stiff sand square (4 particles per cell, stirred): two block step sizes"""
    from tests.golden.make_golden import mpm2d_state
    dx = 1.0 / res
    vol = dx * dx / 4
    xa, va, Fa, Ba = mpm2d_state(res, lo=(30, 40), cells=24, seed=3)
    xb, vb, Fb, Bb = mpm2d_state(res, lo=(54, 40), cells=24, seed=4)
    gpa, _ = tm.group_params("elastic", 400 * vol, vol)
    gpb, _ = tm.group_params("sand", 400 * vol, vol)
    return res, dx, [("elastic", gpa, xa, 0.3 * va, Fa, Ba), ("sand", gpb, xb, 0.3 * vb, Fb, Bb)]


def _pair(tm, res, dx, groups, kw, floor=-0.2, friction=0.4):
    from oracle import refmpm as ref
    r = ref.AsyncSim(res, dx, dim=2, shapes=[(0, 0, 0, 1, 0, floor)], friction=friction, **kw)
    sim = tm.create_simulation2("async_mpm").initialize(dict(res=(res, res), delta_x=dx, **kw))
    sim.set_levelset(tm.mpm.LevelSet(friction=friction).add_plane((0, 1, 0), d=floor))
    for mat, gp, x, v, F, B in groups:
        r.add_particles(mat, gp[0], gp[1], x, v, F, B, None)
        sim.add_particles(dict(type=mat, positions=x, velocities=v, F=F, B=B, params=gp))
    return r, sim


@pytest.mark.parametrize("left_boundary", [False, True])
def test_async_stepping_2d_matches_the_reference_async_stepper(tm, left_boundary):
    """blocks (8 x 16 nodes) advancing with their own power-of-two multiples of unit_delta_t (AsyncMPM<2>::step / advance,
    src/async/async_mpm.cpp:255-421): same pools (every container of every block, at its block's time), same clocks, same number
    of particle updates, particle states to fp32 tolerance.  left_boundary (:43-53, 155-163): the blocks in x <= 0.2 follow the
    SMALLEST step in use."""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)
    res, dx, groups = _two_stiffness_scene_2d(tm)
    kw = dict(unit_delta_t=2e-6, max_units=1024)
    if left_boundary:
        kw["left_boundary"] = True
    r, sim = _pair(tm, res, dx, groups, kw)
    for _ in range(2):
        r.step(2.5e-3)
        sim.step(2.5e-3)
    assert sim.current_t_int == r.time_int()
    assert sim.update_counter == r.update_counter()
    assert (sim.min_delta_t_int, sim.max_delta_t_int) != (1, 1)
    a, b = sim.get_pool_particles(), r.download()
    assert np.array_equal(a["id"], b["id"])
    assert len(np.unique(b["limits"][:, 0])) >= 2, "the scene must step with at least two block step sizes"
    assert np.array_equal(a["continuous"], b["limits"][:, 0]) and np.array_equal(a["particle_t"], b["limits"][:, 3])
    assert np.abs(a["x"] - b["x"]).max() <= 1e-6
    assert rel_l2(a["v"], b["v"]) <= 5e-4 and rel_l2(a["F"], b["F"]) <= 1e-4 and rel_l2(a["B"], b["B"]) <= 2e-3
    # the non-empty blocks' limits, against the reference's table
    want, mm = r.blocks()
    tab = sim.block_table()
    blk = (want["coord"][:, 0] >> 3) * tab["nb"][1] + (want["coord"][:, 1] >> 4)
    assert np.array_equal(tab["count"][blk], want["count"]) and np.array_equal(tab["continuous"][blk], want["continuous"])
    assert (sim.min_delta_t_int, sim.max_delta_t_int) == tuple(int(v) for v in mm)
    sim.close(); r.close()


def test_async_2d_through_the_c_abi_frames_additions_and_growth(tm):
    """(i) mpmhip2d_step IS the asynchronous step once the stepper is resident (the reference's virtual Simulation::step) and
    mpmhip2d_substep is refused; (ii) downloads and counts cover ALL pool containers (AsyncMPM::visualize's particle list), not
    the last working set; (iii) particles added between steps join their pools, the arrays grow past the initial capacity;
    (iv) rigid bodies are refused; (v) many advances squeeze freed containers out of the store without changing the state —
    the same sequence on the reference's stepper: the same pool containers after 16 steps, INCLUDING the two particles the
    reference's filing rule drops on the way (a result that lands in a block that is neither stepping nor due for a backup is
    not filed: src/async/async_mpm.cpp:345-372)"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)
    res, dx, groups = _two_stiffness_scene_2d(tm)
    kw = dict(unit_delta_t=2e-6, max_units=1024)
    r = ref.AsyncSim(res, dx, dim=2, shapes=[(0, 0, 0, 1, 0, -0.2)], friction=0.4, **kw)
    sim = tm.create_simulation2("async_mpm").initialize(dict(res=(res, res), delta_x=dx, **kw))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    mat, gp, x, v, F, B = groups[0]
    sim.add_particles(dict(type=mat, positions=x, velocities=v, F=F, B=B, params=gp))  # 2 304 particles: past the 1 024 the ctx started with
    r.add_particles(mat, gp[0], gp[1], x, v, F, B, None)
    L, ctx = sim._L, sim._ctx
    assert L.mpmhip2d_substep(ctx) < 0 and b"asynchronous" in L.mpmhip2d_last_error(ctx)
    for _ in range(3):
        assert L.mpmhip2d_step(ctx, C.c_float(2e-3)) == 0
        r.step(2e-3)
    st = sim._state()
    assert st[0] == r.time_int() and st[7] == 3
    n_pool = sim.get_num_pool_particles()
    assert n_pool == r.num_particles() >= len(x) and sim.get_num_particles() == n_pool
    p = sim.get_particles()
    assert len(p["id"]) == n_pool and set(np.unique(p["id"])) == set(range(len(x)))
    t0 = sim.get_current_time()
    mat, gp, xb, vb, Fb, Bb = groups[1]
    sim.add_particles(dict(type=mat, positions=xb, velocities=vb, F=Fb, B=Bb, params=gp))
    r.add_particles(mat, gp[0], gp[1], xb, vb, Fb, Bb, None)
    with pytest.raises(tm.MPMError):
        sim.add_particles(dict(type="rigid"))
    sim.step(2e-3); r.step(2e-3)
    assert sim.get_current_time() > t0
    assert set(np.unique(sim.get_particles()["id"])) == set(range(len(x) + len(xb)))
    before = sim._state()
    for _ in range(12):
        sim.step(2e-3); r.step(2e-3)
    after = sim._state()
    assert after[6] > before[6] or after[5] < 4 * after[4], "freed containers pile up: the store was never compacted"
    assert after[0] == r.time_int() and after[1] == r.update_counter()
    a, b = sim.get_pool_particles(), r.download()
    assert np.array_equal(a["id"], b["id"])  # every container of every pool, duplicates and the dropped particles included
    assert np.array_equal(a["particle_t"], b["limits"][:, 3])
    assert np.isfinite(a["F"]).all() and np.abs(a["x"] - b["x"]).max() <= 2e-5
    sim.close(); r.close()


def test_async_2d_refuses_materials_without_a_sound_speed_bound(tm):
    """linear / jelly particles return get_allowed_dt = 0: the reference's stepper stops (src/async/async_mpm.cpp:118-125)"""
    res = 64
    dx = 1.0 / res
    sim = tm.create_simulation2("async_mpm").initialize(dict(res=(res, res), delta_x=dx))
    sim.add_particles(dict(type="jelly", square=(20, 30)))
    with pytest.raises(tm.MPMError, match="allowed time step"):
        sim.step(1e-3)
    sim.close()
