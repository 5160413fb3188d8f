"""CPIC rigid coupling on the device against the reference's own code (oracle/_ref/libmpm_ref.so: src/rigid_transfer.cpp,
src/mpm_rigid_body.cpp and the rigid branches of src/transfer.cpp compiled in place; the rigid body itself is the shim's,
oracle/taichi_shim/taichi/dynamics/rigid_body_shim.h — see DESIGN.md §2 for what that does and does not pin)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

RES, DX, DT = 32, 1.0 / 32, 1e-4


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


@pytest.fixture(scope="module")
def refmpm():
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    refmpm.set_threads(1)  # the reference's impulse sums are order-dependent: one thread keeps them reproducible
    return refmpm


def plate(half=0.2):
    """a square plate in the x-z plane: two triangles"""
    h = half
    return np.array([[[-h, 0, -h], [h, 0, -h], [h, 0, h]], [[-h, 0, -h], [h, 0, h], [-h, 0, h]]], np.float32)


def box(hx=0.1, hy=0.06, hz=0.12):
    """a closed box, outward-facing triangles"""
    c = np.array([[x, y, z] for x in (-hx, hx) for y in (-hy, hy) for z in (-hz, hz)], np.float32)
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    return np.array([[c[a], c[b], c[d]] for a, b, d, e in q] + [[c[a], c[d], c[e]] for a, b, d, e in q], np.float32)


def block_of_particles(lo=10, hi=22, seed=0):
    rng = np.random.default_rng(seed)
    g = np.arange(lo, hi) + 0.25
    X = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    X = np.concatenate([X, X + 0.5])
    X = X + rng.uniform(-0.2, 0.2, X.shape)
    v = rng.normal(0, 0.3, X.shape)
    return (X * DX).astype(np.float32), v.astype(np.float32)


BODIES = {
    "plate": dict(mesh=plate(), codimensional=True, density=40.0, friction=0.3, initial_position=(0.5, 0.5, 0.5),
                  initial_rotation=(20.0, 0.0, 10.0)),
    "box": dict(mesh=box(), codimensional=False, density=400.0, friction0=0.2, friction1=-1.0, initial_position=(0.5, 0.52, 0.5),
                initial_rotation=(0.0, 30.0, 15.0), initial_velocity=(0.3, -0.5, 0.1), initial_angular_velocity=(0.0, 2.0, 1.0)),
}


def make_pair(tm, refmpm, body, material="jelly", **cfg):
    x, v = block_of_particles()
    vol = DX ** 3 / 8
    mass = vol * 400.0
    gp, _ = orc.group_params(material, mass, vol)
    ref = refmpm.Sim(RES, DX, DT, gravity=(0, -10, 0), **cfg)
    b = dict(BODIES[body])
    mesh = b.pop("mesh")
    rid_ref = ref.add_rigid(mesh, **b)
    ref.add_particles(material, mass, vol, x, v)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16, **cfg))
    rid = int(sim.add_particles(dict(type="rigid", mesh=mesh, **b)))
    sim.add_particles(dict(type=material, positions=x, velocities=v, params=gp))
    assert rid == rid_ref
    return ref, sim, rid


@pytest.mark.parametrize("body", sorted(BODIES))
def test_rigid_body_creation_matches_the_reference(tm, refmpm, body):
    ref, sim, rid = make_pair(tm, refmpm, body)
    a, b = ref.rigid_state(rid), sim.get_rigid_state(rid)
    for k in ("position", "rotation", "velocity", "angular_velocity"):
        np.testing.assert_allclose(b[k], a[k], rtol=0, atol=1e-6, err_msg=k)
    assert abs(b["mass"] - a["mass"]) <= 1e-5 * a["mass"]
    np.testing.assert_allclose(b["inertia"], a["inertia"], rtol=1e-5, atol=1e-7 * np.abs(a["inertia"]).max())
    sa, sb = ref.rigid_samples(rid), sim.get_rigid_samples(rid)
    assert len(sb["pos"]) == len(sa["pos"]) > 50
    np.testing.assert_allclose(sb["pos"], sa["pos"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(sb["offset"], sa["offset"], rtol=0, atol=2e-7)


def by_id(sim):
    p = sim.get_particles(sort_by_id=False)
    o = np.argsort(p["id"], kind="stable")
    return p, o


@pytest.mark.parametrize("body", sorted(BODIES))
def test_colored_distance_field_and_particle_colours_match_the_reference(tm, refmpm, body):
    """rasterize_rigid_boundary + gather_cdf (src/rigid_transfer.cpp): the grid's colour tags / body ids / distances and
    the particles' colour words, boundary distances and normals"""
    ref, sim, rid = make_pair(tm, refmpm, body)
    ref.sort(); ref.rasterize_rigid_boundary()
    sim.sort_particles_and_populate_grid(); sim.rasterize_rigid_boundary()
    st_r, d_r = ref.download_cdf()
    st_h, d_h = sim.download_cdf()
    tagged = (st_r & 0xFFFFFF) != 0
    assert tagged.sum() > 200
    # a node a triangle edge passes within rounding distance of may flip in or out: allow a handful
    differ = st_r != st_h
    assert differ.sum() <= 3, differ.sum()
    same = ~differ & tagged
    np.testing.assert_allclose(d_h[same], d_r[same], rtol=0, atol=2e-7)
    ref.gather_cdf(); sim.gather_cdf()
    pr = ref.particle_cdf()
    ph, o = by_id(sim)
    bh = sim.download_boundary()
    ro = np.argsort(ref.download(by_id=False)["id"], kind="stable")
    st_ref, st_hip = pr["states"][ro], ph["states"][o].astype(np.uint32)
    assert (st_ref != 0).sum() > 500
    bad = st_ref != st_hip
    assert bad.sum() <= 3, bad.sum()
    ok = ~bad
    np.testing.assert_array_equal(bh["near"][o][ok], pr["near"][ro][ok])
    near = ok & (pr["near"][ro] != 0)
    assert near.sum() > 300
    np.testing.assert_allclose(bh["distance"][o][near], pr["distance"][ro][near], rtol=0, atol=2e-6 * DX + 1e-8)
    np.testing.assert_allclose(bh["normal"][o][near], pr["normal"][ro][near], rtol=0, atol=2e-4)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def compare_run(ref, sim, rid, n, tol_v=2e-4, tol_body=2e-4):
    ref.substep(n)
    sim.run_substeps(n)
    r = ref.download(by_id=True)
    h = sim.get_particles(sort_by_id=True)
    assert len(h["x"]) == len(r["x"])
    assert np.abs(h["x"] - r["x"]).max() <= 5e-6, np.abs(h["x"] - r["x"]).max()
    assert rel_l2(h["v"], r["v"]) <= tol_v, rel_l2(h["v"], r["v"])
    assert rel_l2(h["F"], r["F"]) <= 1e-4, rel_l2(h["F"], r["F"])
    a, b = ref.rigid_state(rid), sim.get_rigid_state(rid)
    np.testing.assert_allclose(b["position"], a["position"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(b["rotation"], a["rotation"], rtol=0, atol=2e-6)
    scale_v = max(np.abs(a["velocity"]).max(), 1e-3)
    np.testing.assert_allclose(b["velocity"], a["velocity"], rtol=0, atol=tol_body * scale_v)
    scale_w = max(np.abs(a["angular_velocity"]).max(), 1e-3)
    np.testing.assert_allclose(b["angular_velocity"], a["angular_velocity"], rtol=0, atol=tol_body * scale_w)
    return r, h


@pytest.mark.parametrize("body,material", [("plate", "jelly"), ("box", "jelly"), ("plate", "sand"), ("box", "water")])
def test_substeps_with_a_free_rigid_body_match_the_reference(tm, refmpm, body, material):
    """whole substeps (src/mpm.cpp:452-575 with has_rigid_body()): sort, rasterize_rigid_boundary, gather_cdf, P2G and G2P
    with the colour test and the impulses handed to the body, advect_rigid_bodies — a free body falling through / onto
    the material under gravity, with penalty forces on"""
    ref, sim, rid = make_pair(tm, refmpm, body, material, penalty=1e3)
    r, h = compare_run(ref, sim, rid, 5)
    # the body has received impulses from the material: its velocity is not just gravity's
    free_fall = np.array(BODIES[body].get("initial_velocity", (0, 0, 0)), np.float64) + np.array([0, -10.0, 0]) * 5 * DT
    assert np.abs(sim.get_rigid_state(rid)["velocity"] - free_fall).max() > 1e-6
    assert (h["states"] != 0).sum() > 500


def test_substeps_with_a_scripted_rigid_body_match_the_reference(tm, refmpm):
    """a plate on a scripted path (translation + rotation about two axes) cutting through the block: the body follows its
    script (infinite mass and inertia), the material sees its surface velocity"""
    from oracle.refmpm import rigid_script
    p0, vel, amp, omega = (0.5, 0.55, 0.5), (0.2, -1.0, 0.0), (0.0, 0.0, 0.02), 40.0
    e0, rate = (10.0, 0.0, 5.0), (0.0, 90.0, 30.0)
    x, v = block_of_particles()
    vol = DX ** 3 / 8
    mass = vol * 400.0
    gp, _ = orc.group_params("jelly", mass, vol)
    ref = refmpm.Sim(RES, DX, DT, gravity=(0, -10, 0))
    rid = ref.add_rigid(plate(), script=rigid_script(p0, vel, amp, omega, e0, rate), codimensional=True, friction=0.4)
    ref.add_particles("jelly", mass, vol, x, v)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16))
    f32 = np.float32

    def pos(t):
        t = f32(t)
        return [f32(p0[k]) + f32(vel[k]) * t + f32(amp[k]) * f32(np.sin(f32(omega) * t)) for k in range(3)]

    def rot(t):
        t = f32(t)
        return [f32(e0[k]) + f32(rate[k]) * t for k in range(3)]
    assert int(sim.add_particles(dict(type="rigid", mesh=plate(), codimensional=True, friction=0.4, scripted_position=pos,
                                      scripted_rotation=rot))) == rid
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=gp))
    compare_run(ref, sim, rid, 8)
    assert abs(sim.get_rigid_state(rid)["position"][1] - (0.55 - 8 * DT)) < 1e-6
