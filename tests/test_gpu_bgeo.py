"""GPU tests of the .bgeo frame encoder (mpmhip_bgeo_size / _encode / mpmhip_write_bgeo; rows assembled on the
device) through the C ABI: byte for byte against the files the REFERENCE's own Partio writer produced for the same
particle state (tests/golden/bgeo_*, made by tests/golden/make_bgeo_golden.py with oracle/_ref/partio_write), against
the numpy restatement (oracle/bgeo.py) on simulated states, and the host-side ordering / error paths."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle import bgeo as obgeo
from taichi_mpm_amd.materials import group_params
from tests import bgeo_reader
from tests.bgeo_state import CASES, E, MASS, MATERIALS, VOL, debug_triple, make_state
from tests.common import lattice_cube
from tests.common import make_state as make_sim_state

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHA = dict((ln.split()[1], ln.split()[0]) for ln in open(os.path.join(GOLD, "bgeo_sha256.txt")))


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def fixture_sim(tm, s, **cfg):
    """the fixture state in a ctx: one group per material, creation ids overwritten with the fixture's"""
    sim = tm.create_simulation3("mpm")
    sim.initialize(dict(res=(32,) * 3, delta_x=1.0 / 32, base_delta_t=1e-4, keep_apic_b=True, max_particles=len(s["id"]) + 64, **cfg))
    ids = []
    eye = np.eye(3, dtype=np.float32).reshape(1, 9)
    for g, mat in enumerate(MATERIALS):
        m = s["gid"] == g
        if not m.any():
            continue
        row, _ = group_params(mat, float(MASS[g]), float(VOL), **(dict(E=E) if mat == "elastic" else {}))
        sim.add_particles(dict(type=mat, positions=s["x"][m], velocities=s["v"][m], B=s["B"][m], aux=s["aux"][m],
                               F=np.repeat(eye, int(m.sum()), 0), params=row))
        ids.append(s["id"][m])
    sim._ensure_ctx()
    if ids:
        sim.upload(tm.mpm.F_ID, np.concatenate(ids).astype(np.int32))
    return sim


def oracle_bytes(p, verbose, mass_of_gid, mat_of_gid):
    dbg = np.array([debug_triple(mat_of_gid[g], a, E) for g, a in zip(p["gid"], p["aux"])], np.float32).reshape(-1, 3)
    return obgeo.encode(p["x"], p["v"], p["id"], verbose, np.asarray(mass_of_gid, np.float32)[p["gid"]], dbg, p["B"])


@pytest.mark.parametrize("name,n,seed", CASES)
@pytest.mark.parametrize("verbose", [False, True])
def test_bgeo_bytes_equal_the_reference_partio_file(tm, name, n, seed, verbose):
    sim = fixture_sim(tm, make_state(n, seed))
    got = sim.bgeo_bytes(verbose)
    tag = "bgeo_%s_%s" % (name, "verbose" if verbose else "plain")
    path = os.path.join(GOLD, tag + ".bgeo")
    if os.path.exists(path):
        assert got == open(path, "rb").read()
    assert hashlib.sha256(got).hexdigest() == SHA[tag]


def _scene(tm, **cfg):
    x = np.concatenate([lattice_cube(32, 9, 14, 1.0 / 32, jitter=0.2, seed=3), lattice_cube(32, 16, 20, 1.0 / 32, jitter=0.2, seed=4)])
    sj = make_sim_state(x[:5 ** 3 * 8], "jelly", 1.0 / 32, seed=5)
    ss = make_sim_state(x[5 ** 3 * 8:], "water", 1.0 / 32, seed=6)
    sim = tm.create_simulation3("mpm")
    sim.initialize(dict(res=(32,) * 3, delta_x=1.0 / 32, base_delta_t=1e-4, **cfg))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.25))
    for st, mat in ((sj, "jelly"), (ss, "water")):
        sim.add_particles(dict(type=mat, positions=st.x, velocities=st.v, F=st.F, B=st.B, aux=st.aux, params=st.gparams[0]))
    return sim, [float(sj.gparams[0][0]), float(ss.gparams[0][0])]


@pytest.mark.parametrize("keep_b", [True, False])
def test_bgeo_of_a_simulated_state_equals_the_oracle_encoding(tm, keep_b):
    """after stepping, the records sit in physically reordered slots (ids not ascending): the plain file must equal the
    oracle's encoding of the downloaded fields bit for bit; the verbose one too when apic_b is stored, and within
    the recovery error of apic_b (include/mpmhip.h: discard_apic_b) when it is folded into the affine matrix"""
    sim, masses = _scene(tm, keep_apic_b=keep_b)
    sim.run_substeps(25)
    p = sim.get_particles(sort_by_id=False)
    assert not np.all(np.diff(p["id"]) > 0)  # the ordering work is real
    mats = ["jelly", "water"]
    assert sim.bgeo_bytes(False) == oracle_bytes(p, False, masses, mats)
    got = sim.bgeo_bytes(True)
    want = oracle_bytes(p, True, masses, mats)
    # the Frobenius sum of arbitrary floats depends on the summation order (the fixtures above use dyadic apic_b to
    # pin the bytes): every other attribute bit for bit, the norm to a few ulp / to the recovery error of apic_b
    a, b = bgeo_reader.parse(got), bgeo_reader.parse(want)
    assert len(got) == len(want) and a["attrs"] == b["attrs"] and np.array_equal(a["position"], b["position"])
    for k in b["data"]:
        if k != "apic_frobenius_norm":
            assert np.array_equal(a["data"][k], b["data"][k]), k
    na, nb = a["data"]["apic_frobenius_norm"], b["data"]["apic_frobenius_norm"]
    if keep_b:
        assert np.abs(na - nb).max() <= 4e-7 * np.abs(nb).max() and np.abs(na / nb - 1).max() <= 1e-5
    else:
        assert np.abs(na - nb).max() <= 2e-3 * np.abs(nb).max()
    d = bgeo_reader.parse(got)
    assert np.array_equal(d["data"]["debug"][:, 1], np.where(p["gid"][np.argsort(p["id"])] == 0, 4.0, 5.0))


def test_bgeo_sparse_ids_take_the_sort_path(tm):
    """ids far apart (max id >> 16 n): the host orders them by sorting instead of the direct table"""
    s = make_state(500, 9)
    s["id"] = (s["id"].astype(np.int64) * 4001 % 2000003).astype(np.int32)
    assert len(np.unique(s["id"])) == 500 and s["id"].max() > 16 * 500 + 1024
    sim = fixture_sim(tm, s)
    p = sim.get_particles(sort_by_id=False)
    assert sim.bgeo_bytes(True) == oracle_bytes(p, True, MASS[[g for g in range(4) if (s["gid"] == g).any()]],
                                                [m for g, m in enumerate(MATERIALS) if (s["gid"] == g).any()])


def test_visualize_writes_numbered_frames_and_errors_are_reported(tm, tmp_path):
    sim, _ = _scene(tm, frame_directory=str(tmp_path / "frames"), verbose_bgeo=True)
    sim.run_substeps(3)
    f1 = sim.visualize()
    sim.run_substeps(2)
    f2 = sim.visualize()
    assert [os.path.basename(f1), os.path.basename(f2)] == ["0001.bgeo", "0002.bgeo"]  # src/mpm.h:334-336
    assert open(f2, "rb").read() == sim.bgeo_bytes(True)
    d = bgeo_reader.parse(open(f1, "rb").read())
    assert d["n"] == sim.get_num_particles() and [a[0] for a in d["attrs"]][-1] == "apic_frobenius_norm"
    with pytest.raises(tm.mpm.MPMError, match="gzip"):
        sim.write_partio(str(tmp_path / "x.bgeo.gz"))
    with pytest.raises(tm.mpm.MPMError, match="cannot open"):
        sim.write_partio(str(tmp_path / "no_such_dir" / "x.bgeo"))
    n, w = C.c_size_t(), C.c_size_t()
    L = sim._L
    assert L.mpmhip_bgeo_size(sim._ctx, 0, C.byref(n)) == 0
    buf = np.empty(n.value, np.uint8)
    rc = L.mpmhip_bgeo_encode(sim._ctx, 0, buf.ctypes.data_as(C.c_void_p), n.value - 1, C.byref(w))
    assert rc == -4 and b"bgeo image needs" in L.mpmhip_last_error(sim._ctx)  # MPMHIP_ECAPACITY
    assert L.mpmhip_bgeo_encode(sim._ctx, 0, buf.ctypes.data_as(C.c_void_p), n.value, C.byref(w)) == 0 and w.value == n.value


def test_bgeo_frame_of_the_benchmark_scene_at_full_size(tm):
    """C2-sized frame (1 M particles, 48 MB): structure, ascending ids and the size formula hold at scale"""
    sim = tm.create_simulation3("mpm")
    sim.initialize(dict(res=(128,) * 3, delta_x=1.0 / 128, base_delta_t=1e-4))
    sim.add_particles(dict(type="jelly", positions=tm.mpm.lattice_cube(39, 89, 1.0 / 128)))
    sim.run_substeps(3)
    b = sim.bgeo_bytes(False)
    d = bgeo_reader.parse(b)
    assert d["n"] == 1000000 and np.array_equal(d["data"]["index"].ravel(), np.arange(1000000))
    p = sim.get_particles(sort_by_id=True)
    assert np.array_equal(d["position"], p["x"]) and np.array_equal(d["data"]["v"], p["v"])


def test_the_frame_loop_resumes_at_the_loaded_frame(tm, tmp_path):
    """scripts/async/async_mpm.py:236-248: `while self.c.frame < self.num_frames`, snapshots named by c.frame — after load() of
    frame 2's snapshot, simulate() runs frames 3 and 4 (not four more) and writes 0004.tcb, and the state equals the
    uninterrupted run's"""
    x = tm.mpm.lattice_cube(10, 18, 1.0 / 32)

    def scene(out):
        m = tm.MPM(res=(32,) * 3, base_delta_t=2e-4, frame_dt=1e-3, num_frames=4, snapshot_interval=2, output_directory=str(out))
        m.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
        return m
    a = scene(tmp_path / "a")
    a.add_particles(type="sand", positions=x)
    a.simulate()
    assert a.c.frame == 4
    assert sorted(os.listdir(tmp_path / "a" / "snapshots")) == ["0002.tcb", "0004.tcb"]
    assert sorted(os.listdir(tmp_path / "a" / "frames")) == ["%04d.bgeo" % k for k in (1, 2, 3, 4)]
    b = scene(tmp_path / "b")
    b.add_particles(type="sand", positions=x)   # (the groups come from the scene, the particles from the snapshot)
    b.load(str(tmp_path / "a" / "snapshots" / "0002.tcb"))
    assert b.c.frame == 2
    b.simulate()
    assert b.c.frame == 4 and os.listdir(tmp_path / "b" / "snapshots") == ["0004.tcb"]
    assert len(os.listdir(tmp_path / "b" / "frames")) == 2
    assert abs(b.get_current_time() - a.get_current_time()) < 1e-6
    pa, pb = a.c.get_particles(), b.c.get_particles()
    assert np.array_equal(pa["id"], pb["id"]) and np.abs(pa["x"] - pb["x"]).max() < 1e-5


@pytest.mark.parametrize("verbose", [False, True])
def test_2d_bgeo_bytes_equal_the_live_reference_partio_writer(tm, tmp_path, verbose):
    """MPM<2>::write_partio (src/visualize.cpp:17-100; z = 0) of the 2D simulation: byte for byte against the file the reference's
    own 2D simulation writes through its Partio for the same particle state (water + elastic: the two types with a non-trivial
    `debug` attribute), then row by row after three substeps on both sides"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    from tests.golden.make_golden import mpm2d_state
    refmpm.set_threads(1)
    res, dx, dt = 64, 1.0 / 64, 1e-4
    vol = dx * dx / 4
    xa, va, Fa, Ba = mpm2d_state(res, lo=(16, 24), cells=12, seed=11)
    xb, vb, Fb, Bb = mpm2d_state(res, lo=(34, 24), cells=12, seed=12)
    Ba = np.round(Ba * 256) / 256  # dyadic apic_b: the Frobenius norm of the verbose file is then independent of the summation order
    Bb = np.round(Bb * 256) / 256
    gpa, gpb = group_params("water", 400 * vol, vol)[0], group_params("elastic", 400 * vol, vol)[0]
    ref = refmpm.Sim(res, dx, dt, dim=2, gravity=(0, -10), shapes=[(0, 0, 0, 1, 0, -0.2)], friction=0.4, verbose_bgeo=verbose)
    sim = tm.create_simulation2("mpm").initialize(dict(res=(res, res), delta_x=dx, base_delta_t=dt, gravity=(0, -10), verbose_bgeo=verbose,
                                                       frame_directory=str(tmp_path / "frames")))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    aux_w = np.full(len(xa), 1.0, np.float32)
    ref.add_particles("water", gpa[0], gpa[1], xa, va, Fa, Ba, aux_w)
    ref.add_particles("elastic", gpb[0], gpb[1], xb, vb, Fb, Bb, None)
    sim.add_particles(dict(type="water", positions=xa, velocities=va, F=Fa, B=Ba, aux=aux_w, params=gpa))
    sim.add_particles(dict(type="elastic", positions=xb, velocities=vb, F=Fb, B=Bb, params=gpb))
    path = str(tmp_path / "ref.bgeo")
    ref.write_bgeo(path)
    want = open(path, "rb").read()
    got = sim.bgeo_bytes()
    assert got == want
    assert open(sim.visualize(), "rb").read() == want and sim.frame_count == 1  # frames start at 0001.bgeo (src/mpm.h:334-336)
    ref.substep(3)
    sim.run_substeps(3)
    ref.write_bgeo(path)
    a, b = bgeo_reader.parse(sim.bgeo_bytes()), bgeo_reader.parse(open(path, "rb").read())
    assert a["attrs"] == b["attrs"] and np.array_equal(a["data"]["index"], b["data"]["index"])
    assert np.abs(a["position"] - b["position"]).max() <= 5e-7 and np.all(a["position"][:, 2] == 0)
    assert np.abs(np.asarray(a["data"]["v"]) - np.asarray(b["data"]["v"])).max() <= 2e-4
    sim.close(); ref.close()


def test_2d_frames_of_a_scene_with_a_rigid_body_and_of_the_async_stepper(tm, tmp_path):
    """(i) the boundary particles of a 2D rigid body are rows of type 1 with ids from the shared creation counter, in ascending-id
    order among the material particles' rows, as in the reference's file; (ii) a frame of the asynchronous stepper lists every
    container of every pool with its block's limits (src/async/async_visualize.cpp:17-26,86-96)"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    import tests.cpic_scenes as cs
    refmpm.set_threads(1)
    x, v = cs.block2()
    body = dict(cs.BODIES2["box"])
    ref = refmpm.Sim(cs.RES2, cs.DX2, cs.DT, dim=2, gravity=(0, -10))
    sim = tm.create_simulation2("mpm").initialize(dict(res=(cs.RES2,) * 2, delta_x=cs.DX2, base_delta_t=cs.DT, gravity=(0, -10),
                                                       max_particles=len(x) + 16, frame_directory=str(tmp_path / "f")))
    ref.add_particles("jelly", cs.MASS2, cs.VOL2, x[:100], v[:100])   # ids 0..99, then the body's boundary particles, then the rest
    sim.add_particles(dict(type="jelly", positions=x[:100], velocities=v[:100]))
    kw = dict(body)
    ref.add_rigid2(kw.pop("mesh"), **kw)
    sim.add_particles(dict(type="rigid", **body))
    ref.add_particles("jelly", cs.MASS2, cs.VOL2, x[100:], v[100:])
    sim.add_particles(dict(type="jelly", positions=x[100:], velocities=v[100:]))
    path = str(tmp_path / "ref.bgeo")
    ref.write_bgeo(path)
    a, b = bgeo_reader.parse(sim.bgeo_bytes()), bgeo_reader.parse(open(path, "rb").read())
    assert np.array_equal(a["data"]["index"], b["data"]["index"]) and np.array_equal(a["data"]["type"], b["data"]["type"])
    t = np.asarray(b["data"]["type"]).reshape(-1)
    assert 0 < t.sum() < len(t) and t[:100].sum() == 0 and t[100] == 1
    assert np.abs(a["position"] - b["position"]).max() <= 1e-6
    assert np.abs(np.asarray(a["data"]["v"]) - np.asarray(b["data"]["v"])).max() <= 1e-5
    frame = sim.visualize()
    assert os.path.exists(frame) and os.path.exists(os.path.join(os.path.dirname(frame), "rigid_001_0001.poly"))
    sim.close(); ref.close()
    # (ii)
    from tests.test_gpu_async2d import _two_stiffness_scene_2d
    res, dx, groups = _two_stiffness_scene_2d(tm)
    kw = dict(unit_delta_t=2e-6, max_units=1024)
    asim = tm.create_simulation2("async_mpm").initialize(dict(res=(res, res), delta_x=dx, **kw))
    asim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    for mat, gp, xx, vv, F, B in groups:
        asim.add_particles(dict(type=mat, positions=xx, velocities=vv, F=F, B=B, params=gp))
    for _ in range(2):
        asim.step(2.5e-3)
    d = bgeo_reader.parse(asim.bgeo_bytes())["data"]
    p = asim.get_pool_particles()
    lim = np.asarray(d["limit"]).reshape(-1, 3)
    assert np.array_equal(np.asarray(d["index"]).reshape(-1), p["id"])
    assert np.array_equal(np.sort(lim[:, 0]), np.sort(p["continuous"])) and len(np.unique(lim[:, 0])) >= 2
    assert (lim[:, 0] & (lim[:, 0] - 1)).max() == 0 and lim[:, 1].min() >= 1
    asim.close()
