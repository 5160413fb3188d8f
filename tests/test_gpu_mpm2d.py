"""MPM<2> on the device (mpmhip2d_*, taichi_mpm_amd.Simulation2D) against the REFERENCE's 2D simulation
(create_simulation2('mpm'): the generic rasterize / resample of src/transfer.cpp:193-278,585-687 with the dim = 2 particle
types of src/particles.cpp) — committed fixture tests/golden/ref_mpm2d.npz and the live library at a larger size — and the
`optimized=False` switch of the 3D simulation against the reference's generic 3D path (gen_* arrays of substep_*.npz)."""
import json
import os

import numpy as np
import pytest

from tests.common import rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
MATS = ["jelly", "snow", "sand", "water", "linear", "elastic", "von_mises", "visco"]


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def _levelset(tm, rows, friction):
    ls = tm.mpm.LevelSet(friction=friction)
    for r in rows:
        t, io, p = int(r[0]), bool(r[1]), list(r[2:]) + [0.0] * (8 - len(r))
        if t == 0:
            ls.add_plane(p[0:3], d=p[3])
        elif t == 1:
            ls.add_sphere(p[0:3], p[3], io)
        else:
            ls.add_cuboid(p[0:3], (p[3], p[4], 1.0), io)
    return ls


def _cases():
    g = np.load(os.path.join(HERE, "golden", "ref_mpm2d.npz"))
    cases = json.loads(str(g["cases"]))
    return g, cases, [(c, m) for c in cases for m in MATS if "%s_%s" % (c, m) in g]


@pytest.mark.parametrize("case,mat", _cases()[2])
def test_mpm2d_matches_the_reference_fixture(tm, case, mat):
    g, cases, _ = _cases()
    c = cases[case]
    res, dx, dt = int(g["res"]), float(g["dx"]), float(g["dt"])
    sim = tm.create_simulation2("mpm").initialize(dict(res=(res, res), delta_x=dx, base_delta_t=dt, **c["cfg"]))
    if c.get("shapes1"):
        sim.set_levelset(tm.mpm.DynamicLevelSet().initialize(0.0, c["t1"], _levelset(tm, c["shapes"], c["friction"]),
                                                             _levelset(tm, c["shapes1"], c["friction"])))
    else:
        sim.set_levelset(_levelset(tm, c["shapes"], c["friction"]))
    aux0 = {"snow": 1.0, "water": 1.0, "visco": 1000.0}.get(mat, 0.0)
    sim.add_particles(dict(type=mat, positions=g["x"], velocities=g["v"], F=g["F"], B=g["B"], aux=np.full(len(g["x"]), aux0, np.float32),
                           params=g["gp_" + mat]))
    for _ in range(3):
        sim.substep()
    got = sim.get_particles()
    want = g["%s_%s" % (case, mat)]
    assert np.array_equal(got["id"], g["%s_%s_ids" % (case, mat)])
    assert np.abs(got["x"] - want[:, 0:2]).max() <= 5e-7
    assert rel_l2(got["v"], want[:, 2:4]) <= 5e-5
    if mat != "water":
        assert rel_l2(got["F"], want[:, 4:8]) <= 1e-4
    assert rel_l2(got["B"], want[:, 8:12]) <= 2e-4
    assert np.abs(got["aux"] - want[:, 12]).max() <= 5e-5 * max(1.0, np.abs(want[:, 12]).max())
    sim.close()


def test_mpm2d_against_the_live_reference_at_scale(tm):
    """256^2 grid, 100 x 100 cells x 4 = 40 000 sand particles dropping onto a floor: 200 substeps on both sides, compared
    statistically (trajectories diverge), then one further substep from the downloaded state compared tightly"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)  # the generic P2G of the reference is racy with more threads
    from tests.golden.make_golden import mpm2d_state
    res = 256
    dx, dt = 1.0 / res, 1e-4
    x, v, F, B = mpm2d_state(res, lo=(78, 40), cells=100, seed=9)
    vol = dx * dx / 4
    gp, _ = tm.group_params("sand", 400.0 * vol, vol)
    ls = tm.mpm.LevelSet(friction=0.5).add_plane((0, 1, 0), d=-0.12)
    sim = tm.create_simulation2("mpm").initialize(dict(res=(res, res), delta_x=dx, base_delta_t=dt))
    sim.set_levelset(ls)
    sim.add_particles(dict(type="sand", positions=x, velocities=v, F=F, B=B, params=gp))
    sim.run_substeps(200)
    st = sim.get_particles()
    assert len(st["x"]) == len(x) and np.isfinite(st["F"]).all()
    r = ref.Sim(res, dx, dt, dim=2, shapes=[(0, 0, 0, 1, 0, -0.12)], friction=0.5)
    r.add_particles("sand", gp[0], gp[1], st["x"], st["v"], st["F"], st["B"], st["aux"])
    r.substep(1)
    sim.substep()
    a, b = sim.get_particles(), r.download()
    sim.close(); r.close()
    assert len(a["x"]) == len(b["x"])
    assert np.abs(a["x"] - b["x"]).max() <= 2e-7
    assert rel_l2(a["v"], b["v"]) <= 5e-5 and rel_l2(a["F"], b["F"]) <= 1e-4 and rel_l2(a["B"], b["B"]) <= 2e-4


@pytest.mark.parametrize("mat", ["jelly", "sand", "snow"])
def test_optimized_false_is_the_reference_generic_3d_path(tm, mat):
    """config optimized=False (src/mpm.cpp:508-515,546-552: rasterize / resample instead of the *_optimized pair): against the
    reference's generic path on the substep fixture (gen_* arrays)"""
    g = np.load(os.path.join(HERE, "golden", "substep_%s.npz" % mat))
    sim = tm.create_simulation3("mpm").initialize(dict(res=(int(g["res"]),) * 3, delta_x=float(g["dx"]), base_delta_t=float(g["dt"]),
                                                       optimized=False, keep_apic_b=True))
    ls = tm.mpm.LevelSet(friction=float(g["friction"]))
    for p in g["planes"]:
        ls.add_plane(p[:3], d=float(p[3]))
    sim.set_levelset(ls)
    sim.add_particles(dict(type=mat, positions=g["in_x"], velocities=g["in_v"], F=g["in_F"], B=g["in_B"], aux=g["in_aux"],
                           params=g["gparams"][0]))
    sim.substep()
    got = sim.get_particles()
    assert np.abs(got["x"] - g["gen_x"]).max() <= 2e-7
    assert rel_l2(got["v"], g["gen_v"]) <= 2e-5 and rel_l2(got["F"], g["gen_F"]) <= 1e-4 and rel_l2(got["B"], g["gen_B"]) <= 1e-4
    sim.close()


def test_generic_path_clamps_positions_into_the_domain(tm):
    """the one arithmetic difference of the generic path: p.pos clamped into [0, res - eps] after the advection
    (src/transfer.cpp:668-670); the optimised path does not clamp"""
    res, dx, dt = 32, 1.0 / 32, 1e-3
    x = np.array([[0.5, 0.5, 0.97]], np.float32)
    v = np.array([[0.0, 0.0, 200.0]], np.float32)  # would leave the unit cube in one substep
    out = {}
    for optimized in (True, False):
        sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=dt, gravity=(0, 0, 0),
                                                           clean_boundary=False, optimized=optimized, keep_apic_b=True))
        sim.add_particles(dict(type="jelly", positions=np.array([[0.5, 0.5, 0.5]], np.float32)))  # keeps the ctx non-empty
        sim._ensure_ctx(extra=4)
        gp, mat = tm.group_params("jelly", 400 * dx ** 3 / 8, dx ** 3 / 8)
        import ctypes as C
        gi = sim._check(sim._L.mpmhip_add_group(sim._ctx, mat, gp.ctypes.data_as(C.POINTER(C.c_float))))
        sim._groups.append((mat, gp))
        sim._upload_new(gi, x, v, None, None, None)  # bypasses the python-side near-boundary filter on purpose
        sim.substep()
        import ctypes as C2  # noqa: F401
        rec = np.zeros((8, 3), np.float32)
        n = sim._L.mpmhip_download(sim._ctx, 0, rec.ctypes.data_as(C.c_void_p), 8)
        out[optimized] = rec[:max(n, 0)]
        sim.close()
    # generic: the particle was clamped to (res - eps) dx before the deletion test saw it
    assert len(out[True]) >= 1
    zs = sorted(float(p[2]) for p in out[False])
    assert all(z <= 1.0 for z in zs)


def test_dirichlet_boundary_2d_matches_the_live_reference(tm):
    """MPM<2>::apply_dirichlet_boundary_conditions (src/mpm.cpp:374-399, config dirichlet_boundary_radius + the left / right
    distances and velocities): a jelly bar clamped between a wall at rest on the left and a wall pulling to the right"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(1)
    from tests.golden.make_golden import mpm2d_state
    res = 128
    dx, dt = 1.0 / res, 1e-4
    x, v, F, B = mpm2d_state(res, lo=(10, 50), cells=100, seed=3)
    keep = (x[:, 1] < 0.55)
    x, v, F, B = x[keep], v[keep], F[keep], B[keep]
    vol = dx * dx / 4
    gp, _ = tm.group_params("jelly", 400.0 * vol, vol)
    keys = dict(dirichlet_boundary_radius=0.15, dirichlet_distance_right=0.2, dirichlet_boundary_left=0.0, dirichlet_boundary_right=1.5)
    sim = tm.create_simulation2("mpm").initialize(dict(res=(res, res), delta_x=dx, base_delta_t=dt, gravity=(0, 0), **keys))
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, F=F, B=B, params=gp))
    r = ref.Sim(res, dx, dt, dim=2, gravity=(0, 0), **keys)
    r.add_particles("jelly", gp[0], gp[1], x, v, F, B, np.zeros(len(x), np.float32))
    for _ in range(5):
        sim.substep()
    r.substep(5)
    a, b = sim.get_particles(), r.download()
    sim.close(); r.close()
    assert len(a["x"]) == len(b["x"]) == len(x)
    left, right = x[:, 0] < 0.12, x[:, 0] > 0.83   # particles whose whole stencil lies in a clamped strip
    assert left.sum() > 100 and right.sum() > 100
    assert np.abs(a["v"][left]).max() < 1e-6 and np.abs(a["v"][right, 0] - 1.5).max() < 1e-5   # the walls really act
    assert np.abs(a["x"] - b["x"]).max() <= 2e-7
    assert rel_l2(a["v"], b["v"]) <= 5e-5 and rel_l2(a["F"], b["F"]) <= 1e-4


def test_dirichlet_boundary_3d_matches_the_live_reference(tm):
    """MPM<3>::apply_dirichlet_boundary_conditions (src/mpm.cpp:401-412): with dirichlet_boundary_radius > 0 every grid node
    above y = 0.525 is held at rest (the 3D form ignores the radius) — a stirred jelly block reaching across that plane"""
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    ref.set_threads(8)
    from tests.common import lattice_cube, make_state
    res = 64
    dx, dt = 1.0 / res, 1e-4
    s = make_state(lattice_cube(res, 24, 40, dx, jitter=0.2, seed=5), "jelly", dx, perturb_F=0.02, seed=6, vel_scale=2.0)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=dt, dirichlet_boundary_radius=0.1))
    sim.add_particles(dict(type="jelly", positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
    r = ref.Sim(res, dx, dt, dirichlet_boundary_radius=0.1)
    r.add_particles("jelly", s.gparams[0][0], s.gparams[0][1], s.x, s.v, s.F, s.B, s.aux)
    for _ in range(4):
        sim.substep()
    r.substep(4)
    a, b = sim.get_particles(), r.download()
    sim.close(); r.close()
    top = s.x[:, 1] > 0.525 + 2.5 * dx   # whole stencil above the plane: grid velocity zero
    assert top.sum() > 1000 and np.abs(a["v"][top]).max() < 1e-6
    assert (np.abs(a["v"][s.x[:, 1] < 0.45]).max(axis=1) > 0.1).any()
    assert np.abs(a["x"] - b["x"]).max() <= 2e-7
    assert rel_l2(a["v"], b["v"]) <= 2e-5 and rel_l2(a["F"], b["F"]) <= 2e-5


def test_2d_snapshot_restart_continues_the_run_with_a_rigid_body(tm, tmp_path):
    """general_action save / load (src/mpm.cpp:940-960) of the 2D simulation: particles, groups, clocks and the rigid body's record
    come out of the blob; the scene (level set, configuration, the body's outline) is set up again before the load.  The restarted
    run continues like the uninterrupted one (to the summation order of the 2D scatter's float atomics)."""
    import tests.cpic_scenes as cs
    from taichi_mpm_amd.mpm import MPMError
    x, v = cs.block2()
    body = dict(cs.BODIES2["box"])

    def scene(with_particles):
        sim = tm.create_simulation2("mpm").initialize(dict(res=(cs.RES2,) * 2, delta_x=cs.DX2, base_delta_t=cs.DT, gravity=(0, -10),
                                                           max_particles=len(x) + 16, penalty=1e3))
        sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
        sim.add_particles(dict(type="rigid", **body))
        if with_particles:
            sim.add_particles(dict(type="sand", positions=x, velocities=v))
        return sim
    a = scene(True)
    a.run_substeps(10)
    path = str(tmp_path / "snap2d.bin")
    assert a.general_action(dict(action="save", file_name=path)) == ""
    a.run_substeps(10)
    want, wb = a.get_particles(), a.get_rigid_state(1)
    b = scene(False)  # the body of the scene, no particles: they come out of the blob
    assert b.general_action(dict(action="load", file_name=path)) == ""
    assert np.isclose(b.get_current_time(), 10 * cs.DT, rtol=1e-5) and b.get_num_particles() == len(want["id"])
    b.run_substeps(10)
    got, gb = b.get_particles(), b.get_rigid_state(1)
    assert np.array_equal(got["id"], want["id"]) and np.array_equal(got["gid"], want["gid"])
    assert np.abs(got["x"] - want["x"]).max() <= 2e-6 and rel_l2(got["v"], want["v"]) <= 1e-4 and rel_l2(got["F"], want["F"]) <= 1e-5
    np.testing.assert_allclose(gb[:6], wb[:6], atol=1e-5)
    c = tm.create_simulation2("mpm").initialize(dict(res=(cs.RES2,) * 2, delta_x=cs.DX2, base_delta_t=cs.DT))  # a scene WITHOUT the body
    with pytest.raises(MPMError, match="rigid bodies"):
        c.general_action(dict(action="load", file_name=path))
    raw = np.fromfile(path, np.uint8)
    raw[16 + 200] ^= 0xFF
    bad = str(tmp_path / "bad.bin")
    raw[:len(raw) - 7].tofile(bad)  # truncated
    with pytest.raises(MPMError, match="size mismatch"):
        scene(False).general_action(dict(action="load", file_name=bad))
    for s in (a, b, c):
        s.close()


def test_2d_async_snapshot_restart_continues_the_run(tm, tmp_path):
    """the asynchronous stepper's snapshot carries every pool and backup container, the block table and the clocks (the reference
    serialises the same, src/async/async_mpm.h:120-172): a restart continues the run — same clocks, same containers in every pool"""
    from tests.test_gpu_async2d import _two_stiffness_scene_2d
    res, dx, groups = _two_stiffness_scene_2d(tm)
    kw = dict(unit_delta_t=2e-6, max_units=1024)

    def scene(with_particles):
        sim = tm.create_simulation2("async_mpm").initialize(dict(res=(res, res), delta_x=dx, **kw))
        sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
        if with_particles:
            for mat, gp, x, v, F, B in groups:
                sim.add_particles(dict(type=mat, positions=x, velocities=v, F=F, B=B, params=gp))
        return sim
    a = scene(True)
    for _ in range(2):
        a.step(2e-3)
    path = str(tmp_path / "async2d.bin")
    a.save_snapshot(path)
    for _ in range(2):
        a.step(2e-3)
    b = scene(False)
    b.load_snapshot(path)
    assert b.current_t_int > 0 and b.get_num_pool_particles() >= sum(len(g[2]) for g in groups)
    for _ in range(2):
        b.step(2e-3)
    assert b.current_t_int == a.current_t_int and b.update_counter == a.update_counter
    p, q = a.get_pool_particles(), b.get_pool_particles()
    assert np.array_equal(p["id"], q["id"]) and np.array_equal(p["block"], q["block"]) and np.array_equal(p["particle_t"], q["particle_t"])
    assert np.abs(p["x"] - q["x"]).max() <= 2e-6 and rel_l2(p["v"], q["v"]) <= 2e-4
    sync = tm.create_simulation2("mpm").initialize(dict(res=(res, res), delta_x=dx))
    with pytest.raises(tm.MPMError, match="asynchronous"):
        sync.load_snapshot(path)
    for s in (a, b, sync):
        s.close()
