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


@pytest.mark.parametrize("left_boundary", [False, True])
def test_async_stepping_matches_the_reference_async_stepper(tm, left_boundary):
    """create_simulation3('async_mpm').step(dt): blocks advancing with their own power-of-two multiples of unit_delta_t
    (AsyncMPM<dim>::step / advance, src/async/async_mpm.cpp:255-421) against the reference's own stepper on the
    two-stiffness scene: same pools (every container of every block, at its block's time), same number of particle
    updates, particle states to fp32 tolerance.  left_boundary (src/async/async_mpm.cpp:43-53, 155-163): the (empty) blocks in
    x <= 0.2 follow the SMALLEST step in use instead of the largest, so the soft blocks next to them are re-advanced with
    the stiff level (more particle updates: the counter must agree with the reference's)"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)
    res, dx, sa, sb = _two_stiffness_scene()
    kw = dict(unit_delta_t=2e-6, max_units=1024)  # sand blocks step with 256 units, the soft elastic ones with 1024
    if left_boundary:
        kw["left_boundary"] = True
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


# ------------------------------------------------------------------------------------------ the resident stepper
def _async_pair(tm, res, dx, states, kw, shapes, friction=0.4):
    from oracle import refmpm as ref
    r = ref.AsyncSim(res, dx, shapes=shapes, friction=friction, **kw)
    sim = tm.create_simulation3("async_mpm").initialize(dict(res=(res,) * 3, delta_x=dx, **kw))
    ls = tm.mpm.LevelSet(friction=friction)
    for s in shapes:  # (type 0 = plane, inside_out, normal, d) as oracle/refmpm.py takes them
        ls.add_plane(s[2:5], d=s[5])
    sim.set_levelset(ls)
    for s, mat in states:
        r.add_particles(mat, s.gparams[0][0], s.gparams[0][1], s.x, s.v, s.F, s.B, s.aux)
        sim.add_particles(dict(type=mat, positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
    return r, sim


def test_stepping_moves_no_particle_data_across_the_host_boundary_and_frames_hold_every_pool(tm, tmp_path):
    """(i) AsyncMPM::step on the device-resident pools: the ctx's host<->device particle byte counter does not move while
    stepping (the block tables and four counters are all the host sees); (ii) frame output, downloads and counts cover ALL
    pool containers (AsyncMPM::visualize, src/async/async_visualize.cpp:86-96), not the last working set; (iii) particles added
    between steps join their pools; (iv) many advances squeeze freed containers out of the store (compaction) without
    changing the state."""
    from tests.bgeo_reader import parse
    res, dx, sa, sb = _two_stiffness_scene()
    kw = dict(unit_delta_t=2e-6, max_units=1024)
    sim = tm.create_simulation3("async_mpm").initialize(dict(res=(res,) * 3, delta_x=dx, frame_directory=str(tmp_path), **kw))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    sim.add_particles(dict(type="elastic", positions=sa.x, velocities=sa.v, F=sa.F, B=sa.B, aux=sa.aux, params=sa.gparams[0]))
    before = sim.host_particle_bytes()
    assert before > 0  # the upload
    for _ in range(3):
        sim.step(2e-3)
    assert sim.host_particle_bytes() == before
    n_pool = sim.get_num_pool_particles()
    assert n_pool >= len(sa.x)
    frame = parse(open(sim.visualize(), "rb").read())["data"]
    assert len(np.asarray(frame["index"]).reshape(-1)) == n_pool == sim.get_num_particles()
    p = sim.get_particles()
    assert len(p["id"]) == n_pool and set(np.unique(p["id"])) == set(range(len(sa.x)))
    lim = np.asarray(frame["limit"]).reshape(n_pool, -1)
    assert lim[:, 0].min() >= 1 and (lim[:, 0] & (lim[:, 0] - 1)).max() == 0  # powers of two: the pools' block limits
    t0 = sim.get_current_time()
    sim.add_particles(dict(type="sand", positions=sb.x, velocities=sb.v, F=sb.F, B=sb.B, aux=sb.aux, params=sb.gparams[0]))
    assert sim.get_num_pool_particles() == n_pool + len(sb.x)  # (the view of the pools was not pooled a second time)
    before = sim.host_particle_bytes()
    for _ in range(6):
        sim.step(2e-3)
    assert sim.host_particle_bytes() == before
    st = sim._state()
    assert st[6] >= 1, "the store was never compacted: %r" % (st,)
    assert sim.get_current_time() > t0 + 0.0119
    q = sim.get_pool_particles()
    assert set(np.unique(q["id"])) == set(range(len(sa.x) + len(sb.x)))
    assert np.isfinite(q["x"]).all() and np.isfinite(q["v"]).all() and np.abs(q["v"]).max() < 20
    tab = sim.block_table()
    # (`count` = pool sizes at the last update_dt_limits, duplicates of an id included: close to, not equal to, the current number)
    assert abs(int(tab["count"].sum()) - len(q["id"])) <= 64 and len(np.unique(tab["continuous"][tab["count"] > 0])) >= 2
    sim.close()


def test_a_million_particles_in_two_stiffnesses_match_the_live_reference_async_stepper(tm):
    """128^3 grid, 2 x 0.5 M particles (soft Hencky-elastic next to stiff sand), AsyncMPM::step on the device-resident pools
    next to the reference's own stepper on the same scene: block times, update counter, pool membership, states"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(min(16, os.cpu_count() or 1))
    res = 128
    dx = 1.0 / res
    xa = lattice_cube(res, 30, 70, dx, jitter=0.2, seed=11)[::1]
    xa = xa[xa[:, 0] < 50 * dx]   # 20 x 40 x 40 cells x 8 = 256 k ... two slabs side by side
    xb = lattice_cube(res, 30, 70, dx, jitter=0.2, seed=12)
    xb = xb[xb[:, 0] >= 50 * dx]
    sa = make_state(xa, "elastic", dx, perturb_F=0.01, vel_scale=0.5, seed=13)
    sb = make_state(xb, "sand", dx, perturb_F=0.01, vel_scale=0.5, seed=14)
    assert len(xa) + len(xb) >= 500_000
    kw = dict(unit_delta_t=2e-6, max_units=256)
    r, sim = _async_pair(tm, res, dx, ((sa, "elastic"), (sb, "sand")), kw, shapes=[(0, 0, 0, 1, 0, -0.2)])
    before = sim.host_particle_bytes()
    r.step(6e-4)
    sim.step(6e-4)
    assert sim.host_particle_bytes() == before
    assert sim.current_t_int == r.time_int()
    assert sim.update_counter == r.update_counter()
    a, b = sim.get_pool_particles(), r.download()
    assert np.array_equal(a["id"], b["id"])
    assert len(np.unique(b["limits"][:, 0])) >= 2, "the scene must step with at least two block step sizes"
    assert np.array_equal(a["continuous"], b["limits"][:, 0]) and np.array_equal(a["particle_t"], b["limits"][:, 3])
    from tests.common import rel_l2
    assert np.abs(a["x"] - b["x"]).max() <= 1e-6
    assert rel_l2(a["v"], b["v"]) <= 5e-4 and rel_l2(a["F"], b["F"]) <= 1e-4
    sim.close(); r.close()


def test_async_snapshot_restart_continues_the_run(tm, tmp_path):
    """every pool and backup container, the block table and the clocks travel in the snapshot (the reference serialises the
    same, src/async/async_mpm.h:120-172): a fresh simulation that loads it continues like the one that wrote it (up to the
    summation order inside a cell: working sets are gathered in order of arrival)"""
    res, dx, sa, sb = _two_stiffness_scene()
    kw = dict(unit_delta_t=2e-6, max_units=1024)

    def scene(add):
        sim = tm.create_simulation3("async_mpm").initialize(dict(res=(res,) * 3, delta_x=dx, **kw))
        sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
        if add:
            for s, mat in ((sa, "elastic"), (sb, "sand")):
                sim.add_particles(dict(type=mat, positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
        return sim
    a = scene(True)
    a.step(2.5e-3)
    path = str(tmp_path / "async.snap")
    a.save_snapshot(path)
    a.step(2.5e-3)
    b = scene(False)
    b.load_snapshot(path)
    assert b.current_t_int > 0 and b.get_num_pool_particles() >= len(sa.x) + len(sb.x)
    b.step(2.5e-3)
    assert a.current_t_int == b.current_t_int and a.update_counter == b.update_counter
    pa, pb = a.get_pool_particles(), b.get_pool_particles()
    assert np.array_equal(pa["id"], pb["id"]) and np.array_equal(pa["particle_t"], pb["particle_t"])
    assert np.array_equal(pa["continuous"], pb["continuous"])
    from tests.common import rel_l2
    assert np.abs(pa["x"] - pb["x"]).max() <= 2e-6 and rel_l2(pa["v"], pb["v"]) <= 2e-4 and rel_l2(pa["F"], pb["F"]) <= 2e-5
    a.close(); b.close()


def test_pooling_more_batches_than_the_ctx_holds_grows_the_records_before_the_gather(tm):
    """a C-ABI caller may pool several batches of up to `cap` particles each (mpmhip_async_pool_particles empties the
    slots, mpmhip_add_particles only checks slots + n <= cap): the working set of an advance can then exceed the ctx's
    record arrays — they are grown BEFORE k_async_gather writes into them (async_api.h: async_advance)"""
    res, dx, sa, sb = _two_stiffness_scene()
    total = len(sa.x) + len(sb.x)
    sim = tm.create_simulation3("async_mpm").initialize(dict(res=(res,) * 3, delta_x=dx, unit_delta_t=2e-6, max_units=1024,
                                                             max_particles=max(len(sa.x), len(sb.x)) + 64))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    for s, mat in ((sa, "elastic"), (sb, "sand")):
        sim.add_particles(dict(type=mat, positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
        sim._n_added = 0  # what a C caller does: it never tells anybody how many particles it has pooled so far
    assert int(sim._L.mpmhip_capacity(sim._ctx)) < total
    sim.step(2.5e-3)
    assert int(sim._L.mpmhip_capacity(sim._ctx)) >= total  # the stiff level's working set is the whole scene at t = 0
    p = sim.get_pool_particles()
    assert len(np.unique(p["id"])) == total and np.isfinite(p["x"]).all()
    sim.close()


def test_a_snapshot_with_a_container_array_out_of_range_is_refused_before_anything_is_loaded(tm, tmp_path):
    res, dx, sa, sb = _two_stiffness_scene()
    kw = dict(res=(res,) * 3, delta_x=dx, unit_delta_t=2e-6, max_units=1024)
    a = tm.create_simulation3("async_mpm").initialize(dict(kw))
    a.add_particles(dict(type="elastic", positions=sa.x, velocities=sa.v, F=sa.F, B=sa.B, aux=sa.aux, params=sa.gparams[0]))
    a.step(1e-3)
    path = str(tmp_path / "ok.snap")
    a.save_snapshot(path)
    raw = np.fromfile(path, np.uint8)
    n_cont = a.get_num_pool_particles()
    hdr = 8 + (8 + 4 + 4 + 24 + 8 + 7 * 8 + 8 + 8) + 80 * 1 + 6 * 8 * int(np.prod(a.nb))  # frame word, SnapAsync, one group, six block arrays
    for what, word in (("tag", 0x7FFFFFF0), ("id", -5)):
        bad = raw.copy()
        if what == "tag":
            bad[hdr:hdr + 4] = np.array([word], np.uint32).view(np.uint8)
        else:  # ids follow the tags: find the array through the blob's own container count
            containers = (len(raw) - hdr) // (4 + 4 + 128)
            bad[hdr + 4 * containers:hdr + 4 * containers + 4] = np.array([word], np.int32).view(np.uint8)
        p2 = str(tmp_path / ("bad_%s.snap" % what))
        bad.tofile(p2)
        b = tm.create_simulation3("async_mpm").initialize(dict(kw))
        with pytest.raises(tm.mpm.MPMError, match="snapshot container"):
            b.load_snapshot(p2)
        assert b.get_num_pool_particles() == 0  # nothing of the blob got in
        b.load_snapshot(path)                   # ... and the ctx is still usable
        assert b.get_num_pool_particles() == n_cont
        b.close()
    a.close()
