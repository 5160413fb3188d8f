"""CPIC rigid coupling on the device against the REFERENCE's own code: tests/golden/ref_cpic.npz holds output of
src/rigid_transfer.cpp, src/mpm_rigid_body.cpp and the rigid branches of src/transfer.cpp compiled in place
(oracle/_ref/libmpm_ref.so; generator tests/golden/make_golden.py cpic).  The rigid BODY under that code is the shim's
(oracle/taichi_shim/taichi/dynamics/rigid_body_shim.h: the taichi core's is not in the reference tree) — what an impulse
does to a body and how a script becomes a velocity are therefore conventions shared by shim and device, not reference
facts; colours, distances, projections and the impulses themselves are the reference's arithmetic.

Tolerances: grid colour words and particle colour words bit-exact except where a triangle edge or the zero level of the
fitted distance passes within rounding of a node / particle (a handful of entries allowed to flip); distances 2e-7 (grid),
5e-7 (particles, least-squares fit), normals 2e-4; after whole substeps x abs 5e-6, v rel-L2 2e-4, F 1e-4, body velocities
2e-4 relative to the scale of the single impulse terms."""
import os

import numpy as np
import pytest

from tests import cpic_scenes as cs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ref_cpic.npz"))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def by_id(sim):
    p = sim.get_particles(sort_by_id=False)
    return p, np.argsort(p["id"], kind="stable")


CASE_IDS = [c[0] for c in cs.CASES]


@pytest.mark.parametrize("case", cs.CASES, ids=CASE_IDS)
def test_rigid_body_creation_matches_the_reference(tm, gold, case):
    """mass, inertia, initial pose and the boundary particles of MPM::add_rigid_particle (src/mpm_rigid_body.cpp:130-252)"""
    name, body, material, n, cfg = case
    sim, rid = cs.build_device(tm, body, material, **cfg)
    st = sim.get_rigid_state(rid)
    a, b = gold[name + "_body0"], cs.rigid_vector(st)
    np.testing.assert_allclose(b[:13], a[:13], rtol=0, atol=1e-6)
    assert abs(b[13] - a[13]) <= 1e-5 * a[13]
    np.testing.assert_allclose(st["inertia"], gold[name + "_inertia"], rtol=1e-5, atol=1e-7 * np.abs(gold[name + "_inertia"]).max())
    s = sim.get_rigid_samples(rid)
    assert len(s["pos"]) == len(gold[name + "_samples"]) > 50
    np.testing.assert_allclose(s["pos"], gold[name + "_samples"], rtol=0, atol=2e-7)


@pytest.mark.parametrize("case", cs.CASES, ids=CASE_IDS)
def test_colored_distance_field_and_particle_colours_match_the_reference(tm, gold, case):
    """rasterize_rigid_boundary + gather_cdf (src/rigid_transfer.cpp)"""
    name, body, material, n, cfg = case
    sim, rid = cs.build_device(tm, body, material, **cfg)
    sim.sort_particles_and_populate_grid(); sim.rasterize_rigid_boundary()
    st_h, d_h = sim.download_cdf()
    st_r, d_r = np.zeros(st_h.size, np.uint32), np.zeros(st_h.size, np.float32)
    st_r[gold[name + "_cdf_idx"]], d_r[gold[name + "_cdf_idx"]] = gold[name + "_cdf_states"], gold[name + "_cdf_dist"]
    st_h, d_h = st_h.reshape(-1), d_h.reshape(-1)
    assert ((st_r & 0xFFFFFF) != 0).sum() > 200
    differ = st_r != st_h
    assert differ.sum() <= 3, differ.sum()
    same = ~differ & (st_r != 0)
    np.testing.assert_allclose(d_h[same], d_r[same], rtol=0, atol=2e-7)
    sim.gather_cdf()
    ph, o = by_id(sim)
    bh = sim.download_boundary()
    st_ref, st_hip = gold[name + "_p_states"], ph["states"][o].astype(np.uint32)
    assert (st_ref != 0).sum() > 500
    bad = st_ref != st_hip
    assert bad.sum() <= 3, bad.sum()
    near_ref = gold[name + "_p_near"].astype(np.int32)
    flips = (bh["near"][o] != near_ref) & ~bad
    assert flips.sum() <= 2, flips.sum()  # (|det| of the fit within rounding of the guard)
    near = ~bad & ~flips & (near_ref != 0)
    assert near.sum() > 300
    np.testing.assert_allclose(bh["distance"][o][near], gold[name + "_p_dist"][near], rtol=0, atol=5e-7)
    np.testing.assert_allclose(bh["normal"][o][near], gold[name + "_p_normal"][near], rtol=0, atol=2e-4)


def check_run(name, gold, h, body_vec):
    assert len(h["x"]) == len(gold[name + "_x"])
    assert np.abs(h["x"] - gold[name + "_x"]).max() <= 5e-6, np.abs(h["x"] - gold[name + "_x"]).max()
    assert rel_l2(h["v"], gold[name + "_v"]) <= 2e-4, rel_l2(h["v"], gold[name + "_v"])
    assert rel_l2(h["F"], gold[name + "_F"]) <= 1e-4, rel_l2(h["F"], gold[name + "_F"])
    assert (h["states"].astype(np.uint32) != gold[name + "_states"]).sum() <= 5
    a = gold[name + "_body"]
    np.testing.assert_allclose(body_vec[0:7], a[0:7], rtol=0, atol=2e-6)  # position, rotation
    np.testing.assert_allclose(body_vec[7:10], a[7:10], rtol=0, atol=2e-4 * max(np.abs(a[7:10]).max(), 1e-2))
    # (the angular velocity is a cancelling sum of impulse torques: compared on the scale of the single terms)
    np.testing.assert_allclose(body_vec[10:13], a[10:13], rtol=0, atol=2e-4 * max(np.abs(a[10:13]).max(), 1e-2))


@pytest.mark.parametrize("case", cs.CASES, ids=CASE_IDS)
def test_substeps_with_a_rigid_body_match_the_reference(tm, gold, case):
    """whole substeps (src/mpm.cpp:452-575 with has_rigid_body()): sort, rasterize_rigid_boundary, gather_cdf, P2G and G2P with
    the colour test and the impulses handed to the body, advect_rigid_bodies.  Free bodies fall through / onto the block
    under gravity with the penalty force on; the scripted plate follows its path (infinite mass and inertia) while the
    material sees its surface velocity."""
    name, body, material, n, cfg = case
    sim, rid = cs.build_device(tm, body, material, **cfg)
    sim.run_substeps(n)
    h = sim.get_particles(sort_by_id=True)
    vec = cs.rigid_vector(sim.get_rigid_state(rid))
    check_run(name, gold, h, vec)
    assert (h["states"] != 0).sum() > 500
    if body == "scripted":
        assert abs(vec[1] - (cs.SCRIPT["p0"][1] + cs.SCRIPT["vel"][1] * n * cs.DT)) < 1e-6
    else:  # the body has received impulses from the material: its velocity is not just gravity's
        free = np.array(cs.BODIES[body].get("initial_velocity", (0, 0, 0)), np.float64) + np.array([0, -10.0, 0]) * n * cs.DT
        assert np.abs(vec[7:10] - free).max() > 1e-6


def test_live_reference_agrees_with_the_device_on_a_longer_run(tm):
    """the compiled reference next to the device for 40 substeps of the box pushing into sand (no fixture: live)"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    refmpm.set_threads(1)  # (impulse sums are order-dependent: one thread keeps the reference reproducible)
    ref, rid = cs.build_reference(refmpm, "box", "sand", penalty=1e3)
    sim, rid2 = cs.build_device(tm, "box", "sand", penalty=1e3)
    assert rid == rid2
    ref.substep(40)
    sim.run_substeps(40)
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    assert len(r["x"]) == len(h["x"])
    np.testing.assert_array_equal(h["id"], r["id"])  # creation ids count the boundary particles, as the reference's allocator does
    assert np.abs(h["x"] - r["x"]).max() <= 5e-5
    assert rel_l2(h["v"], r["v"]) <= 2e-3
    a, b = cs.rigid_vector(ref.rigid_state(rid)), cs.rigid_vector(sim.get_rigid_state(rid))
    np.testing.assert_allclose(b[0:7], a[0:7], rtol=0, atol=2e-5)
    np.testing.assert_allclose(b[7:10], a[7:10], rtol=0, atol=2e-3 * np.abs(a[7:10]).max())


def read_bgeo_rows(path, verbose=False):
    """point rows of a Houdini .bgeo v5 file as Partio writes it: (n, W) big-endian words behind the attribute table"""
    raw = open(path, "rb").read()
    assert raw[:4] == b"Bgeo" and raw[4:5] == b"V"
    import struct
    n = struct.unpack(">I", raw[9:13])[0]
    n_attr = struct.unpack(">I", raw[25:29])[0]
    off = 41
    width = 4
    for _ in range(n_attr):
        ln = struct.unpack(">H", raw[off:off + 2])[0]
        off += 2 + ln
        cnt = struct.unpack(">H", raw[off:off + 2])[0]
        off += 2 + 4 + 4 * cnt
        width += cnt
    rows = np.frombuffer(raw, dtype=">u4", count=n * width, offset=off).reshape(n, width)
    return rows


def test_bgeo_frame_lists_the_boundary_particles_like_the_reference(tm, tmp_path):
    """write_partio (src/visualize.cpp:17-100) lists every particle of MPM::particles by creation id: material particles
    (type 0) and the boundary particles of rigid bodies (type 1, at their anchor points, moving with the body)"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    refmpm.set_threads(1)
    ref, rid = cs.build_reference(refmpm, "box", "jelly", penalty=1e3)
    sim, _ = cs.build_device(tm, "box", "jelly", penalty=1e3)
    ref.substep(3)
    sim.run_substeps(3)
    fr, fh = str(tmp_path / "ref.bgeo"), str(tmp_path / "hip.bgeo")
    ref.write_bgeo(fr)
    sim.write_partio(fh)
    a, b = read_bgeo_rows(fr), read_bgeo_rows(fh)
    assert a.shape == b.shape and a.shape[1] == 12
    np.testing.assert_array_equal(a[:, 4:9], b[:, 4:9])  # type, index, limit: bit-identical rows in the same order
    assert (a[:, 4] == 1).sum() == len(sim.get_rigid_samples(rid)["pos"]) > 50
    fa, fb = a.astype(">u4").view(">f4"), b.astype(">u4").view(">f4")
    assert np.abs(fa[:, 0:3] - fb[:, 0:3]).max() <= 5e-6
    assert np.abs(fa[:, 9:12] - fb[:, 9:12]).max() <= 2e-4 * np.abs(fa[:, 9:12]).max()


def paddle(r=0.18, h=0.12):
    """two crossed rectangular blades (a paddle wheel), four triangles"""
    a = np.array([[[-r, -h, 0], [r, -h, 0], [r, h, 0]], [[-r, -h, 0], [r, h, 0], [-r, h, 0]]], np.float32)
    b = a[:, :, [2, 1, 0]].copy()
    return np.concatenate([a, b])


def test_rotating_paddle_in_a_million_particles_matches_the_live_reference(tm):
    """BASELINE configs[1] size (128^3 grid, 50^3 cells x 8 = 1 M jelly particles) with a scripted paddle wheel turning
    inside the block (the scene type of scripts/mls-cpic/sand_paddles.py): whole substeps against the compiled reference"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    from tests.common import lattice_cube
    import os as _os
    refmpm.set_threads(min(32, _os.cpu_count() or 1))  # a scripted body takes no impulses: the run is thread-safe
    res, dx, dt, n = 128, 1.0 / 128, 1e-4, 4
    x = lattice_cube(res, 39, 89, dx, jitter=0.15, seed=3)
    rng = np.random.default_rng(4)
    v = rng.normal(0, 0.2, x.shape).astype(np.float32)
    vol = dx ** 3 / 8
    mass = vol * 400.0
    gp, _ = orc.group_params("jelly", mass, vol)
    centre, rate = (0.5, 0.5, 0.5), (0.0, 0.0, 720.0)
    ref = refmpm.Sim(res, dx, dt, gravity=(0, -10, 0))
    rid = ref.add_rigid(paddle(), script=refmpm.rigid_script(centre, (0, 0, 0), (0, 0, 0), 0.0, (0, 0, 0), rate), codimensional=True,
                        friction=-1.0)
    ref.add_particles("jelly", mass, vol, x, v)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=dt, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16))
    f32 = np.float32
    assert int(sim.add_particles(dict(type="rigid", mesh=paddle(), codimensional=True, friction=-1.0,
                                      scripted_position=lambda t: centre,
                                      scripted_rotation=lambda t: [f32(rate[k]) * f32(t) for k in range(3)]))) == rid
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=gp))
    assert len(x) == 1_000_000
    ref.substep(n)
    sim.run_substeps(n)
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    assert len(r["x"]) == len(h["x"]) == len(x)
    assert np.abs(h["x"] - r["x"]).max() <= 2e-6
    assert rel_l2(h["v"], r["v"]) <= 2e-4
    assert rel_l2(h["F"], r["F"]) <= 1e-4
    o = np.argsort(ref.download(by_id=False)["id"], kind="stable")
    st = ref.particle_cdf()["states"][o]
    assert (st != 0).sum() > 50_000
    assert (st != h["states"].astype(np.uint32)).sum() <= 20  # of a million: particles on the zero level of a blade
    a, b = cs.rigid_vector(ref.rigid_state(rid)), cs.rigid_vector(sim.get_rigid_state(rid))
    np.testing.assert_allclose(b[0:13], a[0:13], rtol=0, atol=2e-5)


# ---------------------------------------------------------------------------------------------- 2D (MPM<2>)
@pytest.fixture(scope="module")
def gold2():
    return np.load(os.path.join(HERE, "golden", "ref_cpic2d.npz"))


CASE2_IDS = [c[0] for c in cs.CASES2]


@pytest.mark.parametrize("case", cs.CASES2, ids=CASE2_IDS)
def test_2d_colored_distance_field_and_colours_match_the_reference(tm, gold2, case):
    """MPM<2>: bodies of segments (creation, boundary particles), rasterize_rigid_boundary + gather_cdf in their dim = 2 form"""
    name, body, material, n, cfg = case
    sim, rid = cs.build_device2(tm, body, material, **cfg)
    st0 = sim.get_rigid_state(rid)
    np.testing.assert_allclose(st0[:6], gold2[name + "_body0"][:6], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st0[6:], gold2[name + "_body0"][6:], rtol=1e-5)
    s = sim.get_rigid_samples(rid)
    assert len(s) == len(gold2[name + "_samples"]) >= 10
    np.testing.assert_allclose(s, gold2[name + "_samples"], rtol=0, atol=2e-7)
    sim.cdf_phase()
    st_h, d_h = sim.download_cdf()
    st_r, d_r = np.zeros(st_h.size, np.uint32), np.zeros(st_h.size, np.float32)
    st_r[gold2[name + "_cdf_idx"]], d_r[gold2[name + "_cdf_idx"]] = gold2[name + "_cdf_states"], gold2[name + "_cdf_dist"]
    st_h, d_h = st_h.reshape(-1), d_h.reshape(-1)
    differ = st_r != st_h
    assert differ.sum() <= 2, differ.sum()
    same = ~differ & (st_r != 0)
    assert same.sum() > 50
    np.testing.assert_allclose(d_h[same], d_r[same], rtol=0, atol=2e-7)
    c = sim.download_colours()
    o = np.argsort(sim.get_particles(sort_by_id=False)["id"], kind="stable")
    bad = c["states"][o] != gold2[name + "_p_states"]
    assert bad.sum() <= 2, bad.sum()
    near_ref = gold2[name + "_p_near"].astype(np.int32)
    flips = (c["near"][o] != near_ref) & ~bad
    assert flips.sum() <= 2
    near = ~bad & ~flips & (near_ref != 0)
    assert near.sum() > 50
    np.testing.assert_allclose(c["distance"][o][near], gold2[name + "_p_dist"][near], rtol=0, atol=1e-6)
    np.testing.assert_allclose(c["normal"][o][near], gold2[name + "_p_normal"][near], rtol=0, atol=5e-4)


@pytest.mark.parametrize("case", cs.CASES2, ids=CASE2_IDS)
def test_2d_substeps_with_a_rigid_body_match_the_reference(tm, gold2, case):
    name, body, material, n, cfg = case
    sim, rid = cs.build_device2(tm, body, material, **cfg)
    sim.run_substeps(n)
    h = sim.get_particles(sort_by_id=True)
    assert len(h["x"]) == len(gold2[name + "_x"])
    assert np.abs(h["x"] - gold2[name + "_x"]).max() <= 5e-6
    assert rel_l2(h["v"], gold2[name + "_v"]) <= 2e-4
    assert rel_l2(h["F"], gold2[name + "_F"]) <= 1e-4
    o = np.argsort(sim.get_particles(sort_by_id=False)["id"], kind="stable")
    assert (sim.download_colours()["states"][o] != gold2[name + "_states"]).sum() <= 3
    a, b = gold2[name + "_body"], sim.get_rigid_state(rid)
    np.testing.assert_allclose(b[0:3], a[0:3], rtol=0, atol=2e-6)
    np.testing.assert_allclose(b[3:5], a[3:5], rtol=0, atol=2e-4 * max(np.abs(a[3:5]).max(), 1e-2))
    np.testing.assert_allclose(b[5], a[5], rtol=0, atol=2e-4 * max(abs(a[5]), 1e-1))


# ---------------------------------------------------------------------------------------------- errors and limits
def test_rigid_api_refuses_what_it_cannot_do(tm):
    from taichi_mpm_amd.mpm import MPMError
    sim = tm.create_simulation3("mpm").initialize(dict(res=(32,) * 3, delta_x=1 / 32, base_delta_t=1e-4, max_particles=4096))
    with pytest.raises(MPMError, match="codimensional"):
        sim.add_particles(dict(type="rigid", mesh=cs.plate(), initial_position=(0.5, 0.5, 0.5)))
    with pytest.raises(MPMError, match="initial_position"):
        sim.add_particles(dict(type="rigid", mesh=cs.plate(), codimensional=True))
    with pytest.raises(MPMError, match="cannot coexist"):
        sim.add_particles(dict(type="rigid", mesh=cs.plate(), codimensional=True, initial_position=(0.5, 0.5, 0.5), friction=0.1, friction0=0.2,
                               friction1=0.2))
    rid = int(sim.add_particles(dict(type="rigid", mesh=cs.plate(), codimensional=True, initial_position=(0.5, 0.5, 0.5))))
    assert rid == 1
    x, v = cs.block_of_particles()
    sim.add_particles(dict(type="jelly", positions=x[:1000], velocities=v[:1000]))
    sim.run_substeps(2)
    # (growing past max_particles with bodies present works in place: test_a_scene_with_a_body_grows_without_max_particles)
    for k in range(10):  # 11 bodies fit the 24 colour bits (2 per body, body 0 = background)
        sim.add_particles(dict(type="rigid", mesh=cs.plate(0.05), codimensional=True, initial_position=(0.3 + 0.03 * k, 0.3, 0.5)))
    with pytest.raises(MPMError, match="rigid bodies"):
        sim.add_particles(dict(type="rigid", mesh=cs.plate(0.05), codimensional=True, initial_position=(0.7, 0.7, 0.5)))
    sim.run_substeps(2)  # eleven bodies at once still step
    assert sim.get_num_particles() == 1000


def test_a_body_that_touches_nothing_just_falls(tm):
    """no material near the body: gravity only, the colored distance field stays confined to the body's pages"""
    sim = tm.create_simulation3("mpm").initialize(dict(res=(64,) * 3, delta_x=1 / 64, base_delta_t=1e-4, max_particles=4096))
    rid = int(sim.add_particles(dict(type="rigid", mesh=cs.box(), codimensional=False, initial_position=(0.5, 0.7, 0.5),
                                     initial_angular_velocity=(0.0, 3.0, 0.0))))
    x = (np.stack(np.meshgrid(*[np.arange(20, 26) + 0.5] * 3, indexing="ij"), -1).reshape(-1, 3) / 64).astype(np.float32)
    sim.add_particles(dict(type="jelly", positions=x))
    sim.run_substeps(50)
    st = sim.get_rigid_state(rid)
    np.testing.assert_allclose(st["velocity"], (0.0, -10.0 * 50e-4, 0.0), atol=1e-6)
    np.testing.assert_allclose(st["angular_velocity"], (0.0, 3.0, 0.0), atol=1e-6)
    assert abs(st["position"][1] - (0.7 - 0.5 * 10 * (50e-4) ** 2 * (49 / 50))) < 2e-6  # explicit Euler, position first: sum_{k<n} k dt^2 g
    assert (sim.get_particles()["states"] == 0).all()


def test_two_bodies_at_once_match_the_live_reference(tm):
    """two bodies = two pairs of colour bits: a free box (body 1) and a scripted plate (body 2) in the same block of sand"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    refmpm.set_threads(1)
    x, v = cs.block_of_particles()
    gp = orc.group_params("sand", cs.MASS, cs.VOL)[0]
    s = cs.SCRIPT
    box_cfg = dict(cs.BODIES["box"])
    box_mesh = box_cfg.pop("mesh")
    box_cfg["initial_position"] = (0.42, 0.47, 0.46)
    ref = refmpm.Sim(cs.RES, cs.DX, cs.DT, gravity=(0, -10, 0), penalty=1e3)
    r1 = ref.add_rigid(box_mesh, **box_cfg)
    r2 = ref.add_rigid(cs.plate(0.12), script=refmpm.rigid_script((0.58, 0.56, 0.55), s["vel"], s["amp"], s["omega"], s["e0"], s["rate"]),
                       codimensional=True, friction=0.4)
    ref.add_particles("sand", cs.MASS, cs.VOL, x, v)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16, penalty=1e3))
    f32 = np.float32
    p0 = (0.58, 0.56, 0.55)
    assert int(sim.add_particles(dict(type="rigid", mesh=box_mesh, **box_cfg))) == r1 == 1
    assert int(sim.add_particles(dict(
        type="rigid", mesh=cs.plate(0.12), codimensional=True, friction=0.4,
        scripted_position=lambda t: [f32(p0[k]) + f32(s["vel"][k]) * f32(t) + f32(s["amp"][k]) * f32(np.sin(f32(s["omega"]) * f32(t))) for k in range(3)],
        scripted_rotation=lambda t: [f32(s["e0"][k]) + f32(s["rate"][k]) * f32(t) for k in range(3)]))) == r2 == 2
    sim.add_particles(dict(type="sand", positions=x, velocities=v, params=gp))
    ref.substep(6)
    sim.run_substeps(6)
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    np.testing.assert_array_equal(h["id"], r["id"])
    assert np.abs(h["x"] - r["x"]).max() <= 5e-6
    assert rel_l2(h["v"], r["v"]) <= 2e-4
    o = np.argsort(ref.download(by_id=False)["id"], kind="stable")
    st = ref.particle_cdf()["states"][o]
    assert ((st & 0xC) != 0).sum() > 300 and ((st & 0x30) != 0).sum() > 300  # both bodies colour particles
    assert (st != h["states"].astype(np.uint32)).sum() <= 5
    for rid in (r1, r2):
        a, b = cs.rigid_vector(ref.rigid_state(rid)), cs.rigid_vector(sim.get_rigid_state(rid))
        np.testing.assert_allclose(b[0:7], a[0:7], rtol=0, atol=2e-6)
        np.testing.assert_allclose(b[7:13], a[7:13], rtol=0, atol=2e-4 * max(np.abs(a[7:13]).max(), 1e-2))


def test_rotation_axis_and_damping_match_the_live_reference(tm):
    """rotation_axis (the angular velocity keeps only its component along a world axis: wheels and fans of the scene scripts),
    linear_damping, angular_damping — a free wheel-like box inside the block of particles"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    refmpm.set_threads(1)
    x, v = cs.block_of_particles()
    gp = orc.group_params("jelly", cs.MASS, cs.VOL)[0]
    body = dict(codimensional=False, density=300.0, friction=0.5, initial_position=(0.51, 0.49, 0.5), initial_rotation=(5.0, 0.0, 12.0),
                initial_velocity=(0.0, 0.2, 0.0), initial_angular_velocity=(1.0, 2.0, 6.0), rotation_axis=(0.0, 0.0, 1.0), linear_damping=3.0,
                angular_damping=2.0)
    ref = refmpm.Sim(cs.RES, cs.DX, cs.DT, gravity=(0, -10, 0))
    rid = ref.add_rigid(cs.box(), **body)
    ref.add_particles("jelly", cs.MASS, cs.VOL, x, v)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16))
    assert int(sim.add_particles(dict(type="rigid", mesh=cs.box(), **body))) == rid
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=gp))
    ref.substep(10)
    sim.run_substeps(10)
    a, b = cs.rigid_vector(ref.rigid_state(rid)), cs.rigid_vector(sim.get_rigid_state(rid))
    assert abs(a[10]) < 1e-6 and abs(a[11]) < 1e-6 and abs(a[12]) > 1.0  # the reference's body turns about z only
    np.testing.assert_allclose(b[0:7], a[0:7], rtol=0, atol=2e-6)
    np.testing.assert_allclose(b[7:13], a[7:13], rtol=0, atol=2e-4 * np.abs(a[7:13]).max())
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    assert np.abs(h["x"] - r["x"]).max() <= 5e-6 and rel_l2(h["v"], r["v"]) <= 2e-4


def test_frames_carry_the_rigid_meshes(tm, tmp_path):
    """visualize() = write_bgeo (src/mpm.h:333-343): %04d.bgeo plus rigid_%03d_%04d.obj per body, the mesh in world space"""
    sim, rid = cs.build_device(tm, "scripted", "jelly")
    sim.frame_directory = str(tmp_path)
    sim.run_substeps(8)
    path = sim.visualize()
    assert path.endswith("0001.bgeo") and os.path.exists(path)
    obj = os.path.join(str(tmp_path), "rigid_001_0001.obj")
    lines = open(obj).read().split("\n")
    v = np.array([[float(w) for w in ln.split()[1:]] for ln in lines if ln.startswith("v ")])
    assert v.shape == (6, 3) and lines[6].startswith("f 1 2 3") and lines[7].startswith("f 4 5 6")
    st = sim.get_rigid_state(rid)
    # the plate's corners lie 0.2 * sqrt(2) from its (moving) centre and in a plane through it
    np.testing.assert_allclose(np.linalg.norm(v - st["position"], axis=1), 0.2 * np.sqrt(2), atol=1e-5)
    n = np.cross(v[1] - v[0], v[2] - v[0])
    assert abs(np.dot(n / np.linalg.norm(n), st["position"] - v[0])) < 1e-6
    # 2D: the .poly file of a bar
    sim2, rid2 = cs.build_device2(tm, "bar", "jelly")
    sim2.run_substeps(2)
    poly = open(sim2.write_rigid_body(rid2, str(tmp_path / "bar"))).read().split("\n")
    assert poly[0] == "POINTS" and poly[3] == "POLYS" and poly[4] == "1: 1 2" and poly[5] == "END"


# ---------------------------------------------------------------------------------------------- joints (src/articulation.cpp)
def _joint_scene(tm, joints, dt=cs.JOINT_DT, **cfg):
    sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=dt, gravity=(0, -10, 0),
                                                       max_particles=1 << 12, **cfg))
    for body in cs.JOINT_BODIES:
        sim.add_particles(dict(type="rigid", **body))
    for b, (v, w) in enumerate(cs.JOINT_VELOCITIES):
        sim.set_rigid_velocity(b + 1, v, w)
    for j in joints:
        assert sim.general_action(dict(action="add_articulation", **j)) == ""
    return sim


def _body_rows(sim, n):
    return np.stack([cs.rigid_vector(sim.get_rigid_state(b)) for b in range(1, n + 1)])


def _assert_bodies(got, want, name, vtol):
    """want: fixture rows of the bodies 1.. (position 3, quaternion 4, velocity 3, angular velocity 3, ...)"""
    for b in range(len(got)):
        g, w = got[b], want[b]
        q = g[3:7] if np.dot(g[3:7], w[3:7]) >= 0 else -g[3:7]
        np.testing.assert_allclose(g[0:3], w[0:3], rtol=0, atol=2e-6, err_msg=name)
        np.testing.assert_allclose(q, w[3:7], rtol=0, atol=2e-6, err_msg=name)
        np.testing.assert_allclose(g[7:13], w[7:13], rtol=0, atol=vtol * max(1.0, float(np.abs(want[:, 7:13]).max())), err_msg=name)


@pytest.fixture(scope="module")
def gold_joints():
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_joints.npz"))
    return g, json.loads(str(g["cases"]))


def test_joints_match_the_reference(tm, gold_joints):
    """every Articulation class of src/articulation.cpp on free bodies: set-up, drift (advect_rigid_bodies only), one
    MPM::articulate, then 20 rounds of advect + articulate — against the reference's compiled joints (ref_joints.npz)"""
    g, cases = gold_joints
    nb = len(cs.JOINT_BODIES)
    for name, joints in cases.items():
        sim = _joint_scene(tm, joints)
        _assert_bodies(_body_rows(sim, nb), g[name + "/setup"][1:], name + " setup", 1e-6)
        for _ in range(cs.JOINT_DRIFT):
            sim.advect_rigid_bodies()
        _assert_bodies(_body_rows(sim, nb), g[name + "/drifted"][1:], name + " drifted", 2e-6)
        sim.articulate()
        _assert_bodies(_body_rows(sim, nb), g[name + "/articulated"][1:], name + " articulated", 2e-5)
        for _ in range(20):
            sim.advect_rigid_bodies()
            sim.articulate()
        _assert_bodies(_body_rows(sim, nb), g[name + "/after20"][1:], name + " after20", 2e-4)
        sim.close()


def test_joints_inside_whole_substeps_match_the_live_reference(tm):
    """a hinged pair of boxes with a motor, one of them pushed through a block of jelly: whole substeps (sort, articulate,
    CDF, transfers with impulses to the bodies, advection) next to the compiled reference"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    refmpm.set_threads(1)
    x, v = cs.block_of_particles()
    gp = orc.group_params("jelly", cs.MASS, cs.VOL)[0]
    bodies = [dict(mesh=cs.box(0.08, 0.05, 0.07), codimensional=False, density=400.0, friction=0.3, initial_position=(0.43, 0.52, 0.47),
                   initial_rotation=(5.0, 10.0, -20.0), initial_velocity=(0.2, -0.3, 0.1), initial_angular_velocity=(0.5, 1.0, -2.0)),
              dict(mesh=cs.box(0.05, 0.07, 0.05), codimensional=False, density=300.0, friction=0.3, initial_position=(0.60, 0.58, 0.52),
                   initial_rotation=(0.0, 30.0, 15.0), initial_velocity=(-0.1, 0.0, 0.2), initial_angular_velocity=(1.5, -0.5, 0.3))]
    joints = [dict(type="motor", obj0=1, obj1=2, axis=(0.0, 0.0, 1.0), offset0=(0.08, 0.03, 0.0), power=0.05),
              dict(type="distance", obj0=2, obj1=0, offset1=(0.6, 0.9, 0.5), penalty=2e3),
              dict(type="rotation", obj0=1, obj1=2)]
    ref = refmpm.Sim(cs.RES, cs.DX, cs.DT, gravity=(0, -10, 0))
    sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16))
    for b in bodies:
        kw = dict(b)
        ref.add_rigid(kw.pop("mesh"), **kw)
        sim.add_particles(dict(type="rigid", **b))
    for j in joints:
        ref.general_action(action="add_articulation", **j)
        sim.general_action(dict(action="add_articulation", **j))
    ref.add_particles("jelly", cs.MASS, cs.VOL, x, v)
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=gp))
    ref.substep(30)
    sim.run_substeps(30)
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    np.testing.assert_array_equal(h["id"], r["id"])
    assert np.abs(h["x"] - r["x"]).max() <= 5e-6
    assert rel_l2(h["v"], r["v"]) <= 2e-4
    for rid in (1, 2):
        a, b = cs.rigid_vector(ref.rigid_state(rid)), cs.rigid_vector(sim.get_rigid_state(rid))
        np.testing.assert_allclose(b[0:7], a[0:7], rtol=0, atol=2e-6)
        np.testing.assert_allclose(b[7:13], a[7:13], rtol=0, atol=2e-4 * max(np.abs(a[7:13]).max(), 1e-2))


def test_articulation_api_refuses_what_it_cannot_do(tm):
    from taichi_mpm_amd.mpm import MPMError
    sim = _joint_scene(tm, [])
    for bad, msg in ((dict(type="hinge", obj0=1), "unknown articulation type"), (dict(type="rotation", obj0=7), "not a rigid body"),
                     (dict(type="rotation", obj0=1, obj1=9), "not a rigid body"), (dict(type="distance", obj0=1), "offset1"),
                     (dict(type="motor", obj0=1, obj1=2), "axis")):
        with pytest.raises(MPMError, match=msg):
            sim.general_action(dict(action="add_articulation", **bad))
    sim.close()


def test_2d_rotation_joint_matches_the_live_reference(tm):
    """the joint of scripts/mls-cpic/sand_wheel_2D.py:88 — two 2D bodies sharing one angular velocity — over whole
    substeps in a block of jelly, next to the compiled reference (MPM<2>)"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    refmpm.set_threads(1)
    x, v = cs.block2()
    bodies = [dict(mesh=cs.box2(0.07, 0.04), codimensional=False, density=400.0, friction=0.3, initial_position=(0.44, 0.50),
                   initial_rotation=25.0, initial_velocity=(0.2, -0.3), initial_angular_velocity=3.0),
              dict(mesh=cs.bar2(0.09), codimensional=True, density=60.0, friction=0.3, initial_position=(0.585, 0.56),
                   initial_rotation=-40.0, initial_velocity=(-0.1, -0.2), initial_angular_velocity=-1.0)]
    ref = refmpm.Sim(cs.RES2, cs.DX2, cs.DT, dim=2, gravity=(0, -10), penalty=1e3)
    sim = tm.create_simulation2("mpm").initialize(dict(res=(cs.RES2,) * 2, delta_x=cs.DX2, base_delta_t=cs.DT, gravity=(0, -10),
                                                       max_particles=len(x) + 16, penalty=1e3))
    for b in bodies:
        kw = dict(b)
        ref.add_rigid2(kw.pop("mesh"), **kw)
        sim.add_particles(dict(type="rigid", **b))
    ref.general_action(action="add_articulation", type="rotation", obj0=1, obj1=2)
    assert sim.general_action(dict(action="add_articulation", type="rotation", obj0=1, obj1=2)) == ""
    from taichi_mpm_amd.mpm import MPMError
    with pytest.raises(MPMError, match="rotation"):
        sim.general_action(dict(action="add_articulation", type="motor", obj0=1, obj1=2))
    ref.add_particles("jelly", cs.MASS2, cs.VOL2, x, v)
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=orc.group_params("jelly", cs.MASS2, cs.VOL2)[0]))
    ref.substep(12)
    sim.run_substeps(12)
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    assert len(h["x"]) == len(r["x"])
    assert np.abs(h["x"] - r["x"]).max() <= 5e-6
    assert rel_l2(h["v"], r["v"]) <= 2e-4
    states = [(ref.rigid_state2(rid), sim.get_rigid_state(rid)) for rid in (1, 2)]
    for a, b in states:
        np.testing.assert_allclose(b[0:3], a[0:3], rtol=0, atol=2e-6)
        np.testing.assert_allclose(b[3:5], a[3:5], rtol=0, atol=2e-4 * max(np.abs(a[3:5]).max(), 1e-2))
        np.testing.assert_allclose(b[5], a[5], rtol=0, atol=2e-4 * max(abs(a[5]), 1e-1))
    assert abs(states[1][0][5] + 1.0) > 0.5  # the joint really changed the bar's spin (-1 at the start)


def test_snapshot_restart_with_bodies_and_a_joint_continues_the_run(tm, tmp_path):
    """general_action save / load (src/mpm.cpp:940-960) in a scene with rigid bodies: the blob carries the bodies' records
    and the joints; meshes and scripts come from the scene again (it adds the same bodies before it loads)"""
    from taichi_mpm_amd.mpm import MPMError
    x, v = cs.block_of_particles()
    s = cs.SCRIPT
    f32 = np.float32
    p0 = (0.58, 0.56, 0.55)

    def scene(with_particles):
        sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, gravity=(0, -10, 0),
                                                           max_particles=len(x) + 16, penalty=1e3))
        box_cfg = dict(cs.BODIES["box"])
        box_cfg["initial_position"] = (0.42, 0.47, 0.46)
        sim.add_particles(dict(type="rigid", **box_cfg))
        sim.add_particles(dict(
            type="rigid", mesh=cs.plate(0.12), codimensional=True, friction=0.4,
            scripted_position=lambda t: [f32(p0[k]) + f32(s["vel"][k]) * f32(t) + f32(s["amp"][k]) * f32(np.sin(f32(s["omega"]) * f32(t))) for k in range(3)],
            scripted_rotation=lambda t: [f32(s["e0"][k]) + f32(s["rate"][k]) * f32(t) for k in range(3)]))
        if with_particles:
            sim.general_action(dict(action="add_articulation", type="distance", obj0=1, obj1=0, offset1=(0.45, 0.8, 0.5), penalty=2e3))
            sim.add_particles(dict(type="sand", positions=x, velocities=v))
        return sim
    a = scene(True)
    a.run_substeps(8)
    path = str(tmp_path / "cpic_snap.bin")
    assert a.general_action(dict(action="save", file_name=path)) == ""
    a.run_substeps(8)
    want, wb = a.get_particles(), [cs.rigid_vector(a.get_rigid_state(r)) for r in (1, 2)]
    b = scene(False)  # the bodies of the scene, no particles, no joint: both come out of the blob
    assert b.general_action(dict(action="load", file_name=path)) == ""
    assert np.isclose(b.get_current_time(), 8 * cs.DT, rtol=1e-5)
    b.run_substeps(8)
    got = b.get_particles()
    assert np.array_equal(got["id"], want["id"])
    assert np.abs(got["x"] - want["x"]).max() <= 1e-6
    assert rel_l2(got["v"], want["v"]) <= 1e-5
    assert (got["states"] != want["states"]).sum() <= 2
    for r, w in zip((1, 2), wb):
        g = cs.rigid_vector(b.get_rigid_state(r))
        np.testing.assert_allclose(g[0:7], w[0:7], rtol=0, atol=1e-6)
        np.testing.assert_allclose(g[7:13], w[7:13], rtol=0, atol=1e-5 * max(1.0, float(np.abs(w[7:13]).max())))
    c = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, max_particles=len(x) + 16))
    with pytest.raises(MPMError, match="rigid bodies"):  # a scene without the bodies cannot take the blob
        c.general_action(dict(action="load", file_name=path))
    a.close(); b.close(); c.close()


def test_a_scene_with_a_body_grows_without_max_particles(tm):
    """the reference's mls-cpic scenes add the rigid body first and have no max_particles key: the ctx is created small by
    the body and must grow IN PLACE (mpmhip_reserve) when the material arrives and when more is added mid-run — the run
    equals one that had the capacity from the start (bodies, their impulses and the clocks carry over)"""
    xa, va = cs.block_of_particles(10, 16, seed=5)
    xb, vb = cs.block_of_particles(16, 22, seed=6)

    def run(cap):
        cfg = dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, penalty=1e3)
        if cap:
            cfg["max_particles"] = cap
        sim = tm.create_simulation3("mpm").initialize(cfg)
        rid = int(sim.add_particles(dict(type="rigid", **cs.BODIES["plate"])))
        sim.add_particles(dict(type="jelly", positions=xa, velocities=va))  # > the 1024 slots the body-first ctx starts with
        sim.run_substeps(4)
        sim.add_particles(dict(type="sand", positions=xb, velocities=vb))
        sim.run_substeps(4)
        p, o = by_id(sim)
        st = sim.get_rigid_state(rid)
        t = sim.get_current_time()
        sim.close()
        return {k: v[o] for k, v in p.items()}, st, t
    small, st_s, t_s = run(0)
    big, st_b, t_b = run(len(xa) + len(xb) + 64)
    assert t_s == t_b and len(small["id"]) == len(xa) + len(xb)
    assert np.array_equal(small["id"], big["id"]) and np.array_equal(small["gid"], big["gid"])
    assert np.abs(small["x"] - big["x"]).max() <= 2e-7 and rel_l2(small["v"], big["v"]) <= 2e-5 and rel_l2(small["F"], big["F"]) <= 2e-5
    for k in ("position", "velocity", "angular_velocity"):
        np.testing.assert_allclose(st_s[k], st_b[k], rtol=0, atol=2e-6)


def test_a_raising_script_surfaces_and_two_free_bodies_warn(tm):
    """(i) an exception inside a scripted_position callback cannot cross the C frames: it is stashed and re-raised by the
    stepping call, the body keeps its last finite pose; (ii) rigid-rigid collisions are not implemented: a second body that
    could collide with the first warns instead of silently passing through"""
    sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, max_particles=4096))

    def pos(t):
        if t > 2.5 * cs.DT:
            raise ValueError("script broke at t=%g" % t)
        return (0.5, 0.5 - t, 0.5)
    rid = int(sim.add_particles(dict(type="rigid", mesh=cs.plate(0.1), codimensional=True, scripted_position=pos)))
    x = (np.stack(np.meshgrid(*[np.arange(10, 14) + 0.5] * 3, indexing="ij"), -1).reshape(-1, 3) * cs.DX).astype(np.float32)
    sim.add_particles(dict(type="jelly", positions=x))
    sim.run_substeps(2)
    with pytest.raises(tm.mpm.MPMError, match="script broke"):
        sim.run_substeps(3)
    assert np.isfinite(sim.get_rigid_state(rid)["position"]).all() and sim.get_rigid_state(rid)["position"][1] > 0.4
    with pytest.warns(RuntimeWarning, match="rigid-rigid"):
        sim.add_particles(dict(type="rigid", mesh=cs.box(), codimensional=False, density=400.0, initial_position=(0.3, 0.7, 0.3)))
    sim.close()


def test_one_stream_and_two_streams_give_the_same_run(tm, monkeypatch):
    """The colour-aware transfer kernels run on a second stream beside the plain ones (mpmhip.hip: rigid_fork / rigid_join;
    MPMHIP_RIGID_CONCURRENT=0 puts everything on the ctx stream): disjoint blocks and particles, so the two ways must agree up
    to the order of the float atomics that sum the impulses on a free body."""
    from tests.common import lattice_cube
    res, dx = 64, 1.0 / 64
    x = lattice_cube(res, 20, 44, dx, jitter=0.15, seed=5)

    def run(flag):
        monkeypatch.setenv("MPMHIP_RIGID_CONCURRENT", flag)  # (read when the ctx is created)
        sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=1e-4, gravity=(0, -10, 0),
                                                           max_particles=len(x) + 16))
        wheel = int(sim.add_particles(dict(type="rigid", mesh=paddle(0.2, 0.15), codimensional=True, friction=-2,
                                           scripted_position=lambda t: (0.5, 0.5, 0.5), scripted_rotation=lambda t: (0.0, 0.0, 720.0 * t))))
        free = int(sim.add_particles(dict(type="rigid", mesh=cs.box() * 0.5, codimensional=False, friction=0.3, density=40.0,
                                          initial_position=(0.5, 0.72, 0.5))))
        sim.add_particles(dict(type="sand", positions=x))
        sim.run_substeps(40)
        p = sim.get_particles(sort_by_id=True)
        st = sim.get_rigid_state(free)
        sim.close()
        assert wheel != free
        return p, st
    (a, sa), (b, sb) = run("7"), run("0")
    assert (a["states"] != 0).sum() > 1000, "the scene must colour particles"
    assert np.array_equal(a["id"], b["id"])
    assert np.abs(a["x"] - b["x"]).max() <= 2e-6 and rel_l2(a["v"], b["v"]) <= 1e-4 and rel_l2(a["F"], b["F"]) <= 1e-5
    assert (a["states"] != b["states"]).sum() <= 5
    np.testing.assert_allclose(sa["velocity"], sb["velocity"], atol=2e-5)
    np.testing.assert_allclose(sa["position"], sb["position"], atol=1e-6)


@pytest.mark.parametrize("friction,restitution", [(0.0, 0.0), (0.4, 0.5)])
def test_rigid_body_levelset_collision_matches_the_live_reference(tm, friction, restitution):
    """config rigid_body_levelset_collision (src/mpm.cpp:535-538 -> MPM::rigid_body_levelset_collision, src/mpm_rigid_body.cpp:347-387):
    a tilted, spinning free box thrown at a floor plane and a side wall.  Every penetrating boundary particle hands its body an
    impulse at once, in the order of the reference's sorted particle list — which the device keeps for the boundary particles —
    so body state and particles follow the reference through the contact, bounce included."""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    refmpm.set_threads(1)
    x, v = cs.block_of_particles(lo=12, hi=18)
    x = x + np.float32([0.0, 0.20, 0.0])   # a small jelly block above the box (so that material particles exist)
    gp = orc.group_params("jelly", cs.MASS, cs.VOL)[0]
    body = dict(codimensional=False, density=300.0, friction=friction, restitution=restitution, initial_position=(0.5, 0.47, 0.5),
                initial_rotation=(12.0, 20.0, 31.0), initial_velocity=(0.6, -1.5, 0.3), initial_angular_velocity=(2.0, -1.0, 3.0))
    shapes = [(0, 0, 0, 1, 0, -0.3), (0, 0, -1, 0, 0, 0.66)]   # floor y = 0.3, wall x = 0.66 (phi = 0.66 - x)
    keys = dict(rigid_body_levelset_collision=True)
    ref = refmpm.Sim(cs.RES, cs.DX, cs.DT, gravity=(0, -10, 0), shapes=shapes, friction=0.3, **keys)
    rid = ref.add_rigid(cs.box(), **body)
    ref.add_particles("jelly", cs.MASS, cs.VOL, x, v)
    sim = tm.create_simulation3("mpm").initialize(dict(res=(cs.RES,) * 3, delta_x=cs.DX, base_delta_t=cs.DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16, **keys))
    ls = tm.mpm.LevelSet(friction=0.3).add_plane((0, 1, 0), d=-0.3).add_plane((-1, 0, 0), d=0.66)
    sim.set_levelset(ls)
    assert int(sim.add_particles(dict(type="rigid", mesh=cs.box(), **body))) == rid
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=gp))
    vy, hit = [], False
    for k in range(12):
        ref.substep(25)
        sim.run_substeps(25)
        a, b = cs.rigid_vector(ref.rigid_state(rid)), cs.rigid_vector(sim.get_rigid_state(rid))
        vy.append(a[8])
        np.testing.assert_allclose(b[0:7], a[0:7], rtol=0, atol=5e-5, err_msg="pose after %d substeps" % (25 * (k + 1)))
        np.testing.assert_allclose(b[7:13], a[7:13], rtol=0, atol=2e-3 * max(np.abs(a[7:13]).max(), 1.0))
    assert min(vy) < -1.55 and vy[-1] > min(vy) + 0.5, vy   # the box fell, hit the floor and was stopped / thrown back
    free = refmpm.Sim(cs.RES, cs.DX, cs.DT, gravity=(0, -10, 0), shapes=shapes, friction=0.3)  # the same scene WITHOUT the key
    rf = free.add_rigid(cs.box(), **body)
    free.add_particles("jelly", cs.MASS, cs.VOL, x, v)
    free.substep(300)
    assert cs.rigid_vector(free.rigid_state(rf))[8] < a[8] - 0.3     # ... keeps falling through the floor: the key is what stops the box
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    assert np.abs(h["x"] - r["x"]).max() <= 2e-5 and rel_l2(h["v"], r["v"]) <= 1e-3


@pytest.mark.parametrize("friction,restitution", [(0.0, 0.0), (0.4, 0.5)])
def test_2d_rigid_body_levelset_collision_matches_the_live_reference(tm, friction, restitution):
    """config rigid_body_levelset_collision in the 2D simulation (MPM<2>::rigid_body_levelset_collision, src/mpm_rigid_body.cpp:347-387):
    a tilted, spinning free box thrown at a floor line and a side wall; the impulses go in the order of the reference's sorted
    particle list (2D SPGrid key: 8 x 16-node blocks), so the body follows the reference through the contact"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from oracle import oracle as orc
    refmpm.set_threads(1)
    x, v = cs.block2(lo=24, hi=30)
    x = x + np.float32([0.0, 0.22])   # a small jelly block above the box (so that material particles exist)
    gp = orc.group_params("jelly", cs.MASS2, cs.VOL2)[0]
    body = dict(codimensional=False, density=300.0, friction=friction, restitution=restitution, initial_position=(0.5, 0.385),
                initial_rotation=17.0, initial_velocity=(0.6, -1.5), initial_angular_velocity=2.5)
    shapes = [(0, 0, 0, 1, 0, -0.3), (0, 0, -1, 0, 0, 0.605)]   # floor y = 0.3, wall x = 0.605 (phi = 0.605 - x)
    keys = dict(rigid_body_levelset_collision=True)
    ref = refmpm.Sim(cs.RES2, cs.DX2, cs.DT, dim=2, gravity=(0, -10), shapes=shapes, friction=0.3, **keys)
    rid = ref.add_rigid2(cs.box2(), **body)
    ref.add_particles("jelly", cs.MASS2, cs.VOL2, x, v)
    sim = tm.create_simulation2("mpm").initialize(dict(res=(cs.RES2,) * 2, delta_x=cs.DX2, base_delta_t=cs.DT, gravity=(0, -10),
                                                       max_particles=len(x) + 16, **keys))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.3).add_plane((0, 1, 0), d=-0.3).add_plane((-1, 0, 0), d=0.605))
    assert int(sim.add_particles(dict(type="rigid", mesh=cs.box2(), **body))) == rid
    sim.add_particles(dict(type="jelly", positions=x, velocities=v, params=gp))
    vy = []
    for k in range(12):
        ref.substep(25)
        sim.run_substeps(25)
        a, b = ref.rigid_state2(rid), sim.get_rigid_state(rid)
        vy.append(float(a[4]))
        np.testing.assert_allclose(b[0:3], a[0:3], rtol=0, atol=5e-5, err_msg="pose after %d substeps" % (25 * (k + 1)))
        np.testing.assert_allclose(b[3:6], a[3:6], rtol=0, atol=2e-3 * max(np.abs(a[3:6]).max(), 1.0))
    assert min(vy) < -1.55 and vy[-1] > min(vy) + 0.5, vy   # the box fell, hit the floor and was stopped / thrown back
    free = refmpm.Sim(cs.RES2, cs.DX2, cs.DT, dim=2, gravity=(0, -10), shapes=shapes, friction=0.3)  # the same scene WITHOUT the key
    rf = free.add_rigid2(cs.box2(), **body)
    free.add_particles("jelly", cs.MASS2, cs.VOL2, x, v)
    free.substep(300)
    assert free.rigid_state2(rf)[4] < a[4] - 0.3     # ... keeps falling through the floor: the key is what stops the box
    r, h = ref.download(by_id=True), sim.get_particles(sort_by_id=True)
    assert np.abs(h["x"] - r["x"]).max() <= 2e-5 and rel_l2(h["v"], r["v"]) <= 1e-3
