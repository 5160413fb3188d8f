"""The restated CPU oracle against the REFERENCE's own arithmetic.

tests/golden/{substep_*,ref_materials,ref_kernels,ref_shapes}.npz are output of the reference's solver compiled from
its sources where they lie (oracle/_ref/libmpm_ref.so; generator tests/golden/make_golden.py).  This file checks, on
any box and without the reference tree, that oracle/liboracle.so (the restatement every other test leans on)
reproduces them.  Tolerances are the fp32 parity tolerances of SURVEY §8(d); what is seen in practice is ~1e-7.
The live-reference checks at the bottom run only where libmpm_ref.so exists."""
import glob
import json
import os

import numpy as np
import pytest

from tests.common import rel_l2

HERE = os.path.dirname(__file__)
GOLD = sorted(glob.glob(os.path.join(HERE, "golden", "substep_*.npz")))
MATS = ["jelly", "snow", "sand", "water", "linear", "elastic", "von_mises", "visco"]
F_TOL = {"snow": 1e-5, "sand": 1e-5, "von_mises": 1e-5, "visco": 2e-5}


def _state(orc, g):
    return orc.State(g["in_x"], g["in_v"], g["in_B"], g["in_F"], g["in_aux"], None, g["gparams"], g["gtype"])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_the_reference_phase_by_phase(orc, path):
    g = np.load(path)
    assert "reference" in str(g["source"])
    mat = os.path.basename(path)[len("substep_"):-4]
    cfg = orc.make_config(int(g["res"]), float(g["dx"]), float(g["dt"]), planes=g["planes"].tolist(),
                          friction=float(g["friction"]))
    s = _state(orc, g)
    grid = orc.p2g(cfg, s)  # vs rasterize_optimized, src/transfer.cpp:467-569
    nz = g["nz"].astype(int)
    sel = grid[nz[:, 0], nz[:, 1], nz[:, 2]]
    assert rel_l2(sel[:, 3], g["p2g_nz"][:, 3]) <= 1e-6 and rel_l2(sel[:, :3], g["p2g_nz"][:, :3]) <= 1e-5
    assert np.count_nonzero(grid[..., 3]) == len(nz)
    # the reference's generic path (optimized=False, :193-278) gives the same grid
    assert rel_l2(g["gen_p2g_nz"], g["p2g_nz"]) <= 1e-6
    orc.grid_update(cfg, grid)  # vs src/mpm.cpp:277-372 + friction_project
    assert rel_l2(grid[nz[:, 0], nz[:, 1], nz[:, 2]][:, :3], g["upd_nz"][:, :3]) <= 1e-5
    orc.g2p(cfg, s, grid)  # vs resample_optimized, src/transfer.cpp:837-954
    assert np.abs(s.x - g["out_x"]).max() <= 1e-7
    assert rel_l2(s.v, g["out_v"]) <= 1e-5 and rel_l2(s.B, g["out_B"]) <= 1e-5
    assert rel_l2(s.F, g["out_F"]) <= F_TOL.get(mat, 1e-5)
    assert np.abs(s.aux - g["out_aux"]).max() <= 2e-6 * max(1.0, np.abs(g["out_aux"]).max())
    assert np.abs(g["gen_x"] - g["out_x"]).max() <= 1e-7 and rel_l2(g["gen_F"], g["out_F"]) <= 1e-5


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_five_reference_substeps(orc, path):
    g = np.load(path)
    cfg = orc.make_config(int(g["res"]), float(g["dx"]), float(g["dt"]), planes=g["planes"].tolist(),
                          friction=float(g["friction"]))
    s = _state(orc, g)
    for _ in range(5):
        orc.substep(cfg, s)
    assert np.array_equal(s.ids, g["out5_ids"])
    assert np.abs(s.x - g["out5_x"]).max() <= 5e-7
    assert rel_l2(s.v, g["out5_v"]) <= 1e-4 and rel_l2(s.F, g["out5_F"]) <= 1e-4


@pytest.mark.parametrize("mat", MATS)
def test_oracle_materials_match_the_reference(orc, mat):
    """calculate_force / plasticity of every registered type (src/particles.cpp) on 384 seeded states: small and large
    strains, states beyond the clamps and yield surfaces"""
    g = np.load(os.path.join(HERE, "golden", "ref_materials.npz"))
    gp, t = g[mat + "_gp"], int(g[mat + "_type"])
    F, cdg, aux = g[mat + "_F"], g[mat + "_cdg"], g[mat + "_aux"]
    n = len(F)
    force = np.stack([orc.calculate_force(t, gp, F[i], float(aux[i])).reshape(9) for i in range(n)])
    ref = g[mat + "_force"]
    ok = np.isfinite(ref).all(1)
    assert ok.mean() > 0.95
    # the stress carries an absolute error ~ 2 mu vol eps (F - R cancellation) on top of the relative one
    atol = 2 * gp[2] * gp[1] * 4e-6 if mat != "water" else 0.0
    assert np.abs(force[ok] - ref[ok]).max() <= 2e-5 * np.abs(ref[ok]).max() + atol, mat
    F2 = np.zeros_like(F); aux2 = np.zeros_like(aux)
    for i in range(n):
        f, a = orc.plasticity(t, gp, cdg[i], F[i], float(aux[i]))
        F2[i], aux2[i] = f.reshape(9), a
    ok2 = np.isfinite(g[mat + "_F2"]).all(1)
    assert ok2.mean() > 0.95
    assert np.abs(F2[ok2] - g[mat + "_F2"][ok2]).max() <= 1e-5, mat
    assert np.abs(aux2[ok2] - g[mat + "_aux2"][ok2]).max() <= 1e-5 * max(1.0, np.abs(g[mat + "_aux2"][ok2]).max()), mat


def test_oracle_kernels_and_friction_match_the_reference(orc):
    g = np.load(os.path.join(HERE, "golden", "ref_kernels.npz"))
    inv_dx = float(g["inv_dx"])
    for i, p in enumerate(g["pos"]):
        assert np.abs(orc.kernel3_dw_w(p, inv_dx) - g["fast"][i]).max() <= 1e-5 * inv_dx
        assert np.abs(orc.kernel3_dw_w(p, inv_dx, slow=True) - g["slow"][i]).max() <= 1e-5 * inv_dx
        assert np.abs(orc.kernel2_dw_w(p[:2], inv_dx) - g["k2"][i]).max() <= 1e-5 * inv_dx
        # the reference's own KATs (src/tests.cpp:19-24,35-51) hold on the reference's output
        assert abs(g["slow"][i][:, 3].sum() - 1) < 1e-5 and np.abs(g["slow"][i][:, :3].sum(0)).max() < 1e-4 * inv_dx
        assert np.abs(g["slow"][i] - g["fast"][i]).max() <= 1e-6 * inv_dx
    for row, want in zip(g["friction_in"], g["friction_out"]):
        got = orc.friction_project(row[0:3], row[3:6], row[6:9], float(row[9]))
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def _shape_cases():
    g = np.load(os.path.join(HERE, "golden", "ref_shapes.npz"))
    return g, json.loads(str(g["cases"]))


def _oracle_shapes(rows):
    planes, shapes = [], []
    for r in rows:
        if int(r[0]) == 0:
            planes.append(tuple(r[2:6]))
        else:
            shapes.append((int(r[0]), int(r[1])) + tuple(r[2:8]))
    return planes, shapes


STATIC_CASES = ["static_shapes", "slip_sphere", "particle_collision", "grid_gravity", "moving_plane", "moving_sphere",
                "shrinking_box"]


@pytest.mark.parametrize("mat", ["jelly", "sand"])
@pytest.mark.parametrize("case", STATIC_CASES)
def test_oracle_level_sets_and_config_variants_match_the_reference(orc, case, mat):
    g, cases = _shape_cases()
    c = cases[case]
    planes, shapes = _oracle_shapes(c["shapes"])
    dyn = {}
    if c.get("shapes1") is not None:  # DynamicLevelSet(0, t1, shapes, shapes1)
        p1, s1 = _oracle_shapes(c["shapes1"])
        dyn = dict(planes1=p1, shapes1=s1, t0=0.0, t1=c["t1"])
    cfg = orc.make_config(int(g["res"]), float(g["dx"]), float(g["dt"]), planes=planes, shapes=shapes, friction=c["friction"],
                          **dyn, **c["cfg"])
    a = g["in_" + mat]
    gp = g["gp_" + mat]
    t = orc.TYPE_IDS[mat]
    s = orc.State(a[:, 0:3], a[:, 3:6], a[:, 6:15], a[:, 15:24], a[:, 24], None, gp[None], np.array([t], np.int32))
    for _ in range(3):
        orc.substep(cfg, s)
    want = g["%s_%s_opt" % (case, mat)]
    assert np.array_equal(s.ids, g["%s_%s_opt_ids" % (case, mat)])
    assert np.abs(s.x - want[:, 0:3]).max() <= 5e-7
    assert rel_l2(s.v, want[:, 3:6]) <= 5e-5 and rel_l2(s.F, want[:, 6:15]) <= 5e-5


@pytest.mark.parametrize("case", ["apic_damping_only", "both_dampings"])
def test_oracle_damping_is_the_reference_generic_path_not_its_optimised_quirk(orc, case):
    """the optimised G2P damps only when BOTH dampings are non-zero and then passes the block index
    to damp_affine_momemtum (src/transfer.cpp:925-926); the generic path (:654) applies the intended damping
    (src/mpm.h:465-469).  The oracle (and the HIP path) follow the generic path; the fixture documents both."""
    g, cases = _shape_cases()
    c = cases[case]
    planes, shapes = _oracle_shapes(c["shapes"])
    cfg = orc.make_config(int(g["res"]), float(g["dx"]), float(g["dt"]), planes=planes, shapes=shapes, friction=c["friction"],
                          **c["cfg"])
    a = g["in_jelly"]
    s = orc.State(a[:, 0:3], a[:, 3:6], a[:, 6:15], a[:, 15:24], a[:, 24], None, g["gp_jelly"][None],
                  np.array([orc.TYPE_IDS["jelly"]], np.int32))
    for _ in range(3):
        orc.substep(cfg, s)
    gen, opt = g[case + "_jelly_gen"], g[case + "_jelly_opt"]
    assert rel_l2(s.v, gen[:, 3:6]) <= 5e-5 and rel_l2(s.F, gen[:, 6:15]) <= 5e-5
    assert rel_l2(opt[:, 3:6], gen[:, 3:6]) > 1e-4  # the optimised path of the reference really differs


# ----------------------------------------------------------------------------------------------- live reference
def _ref():
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so not built (needs the reference tree)")
    refmpm.set_threads(1)
    return refmpm


@pytest.mark.parametrize("mat", ["jelly", "sand", "snow"])
def test_live_reference_reproduces_its_fixture(mat):
    ref = _ref()
    g = np.load(os.path.join(HERE, "golden", "substep_%s.npz" % mat))
    sim = ref.Sim(int(g["res"]), float(g["dx"]), float(g["dt"]), shapes=[(0, 0) + tuple(p) for p in g["planes"].tolist()],
                  friction=float(g["friction"]))
    gp = g["gparams"][0]
    sim.add_particles(mat, gp[0], gp[1], g["in_x"], g["in_v"], g["in_F"], g["in_B"], g["in_aux"])
    sim.substep(5)
    d = sim.download()
    sim.close()
    assert np.array_equal(d["id"], g["out5_ids"])
    assert np.abs(d["x"] - g["out5_x"]).max() <= 1e-7 and rel_l2(d["F"], g["out5_F"]) <= 1e-6


def test_live_reference_benchmark_generator_is_the_lattice_of_the_scene_builders():
    """MPM<3>::add_particles with benchmark=125 (src/mpm.cpp:149-186) against lattice_cube, which bench.py and the
    parity tests use to seed the same scene on the device"""
    ref = _ref()
    from taichi_mpm_amd.mpm import lattice_cube
    res = 64
    sim = ref.Sim(res, 1.0 / res, 1e-4)
    sim.add_benchmark("linear", 125)
    d = sim.download()
    sim.close()
    lo = int(round(res * 0.4))
    hi = lo + int(round(res * 0.2))
    x = lattice_cube(lo, hi, 1.0 / res)
    assert len(d["x"]) == len(x) == (hi - lo) ** 3 * 8
    a = d["x"][np.lexsort(np.round(d["x"] * res * 4).astype(int).T[::-1])]
    b = x[np.lexsort(np.round(x * res * 4).astype(int).T[::-1])]
    assert np.abs(a - b).max() <= 1e-7


@pytest.mark.parametrize("verbose", [False, True], ids=["plain", "verbose"])
def test_live_reference_bgeo_bytes_equal_the_restated_writer(tmp_path, verbose):
    """MPM<3>::write_partio (src/visualize.cpp:17-100) on a live reference state vs oracle/bgeo.py on the same state"""
    ref = _ref()
    from oracle import bgeo
    g = np.load(os.path.join(HERE, "golden", "substep_jelly.npz"))
    gp = g["gparams"][0]
    sim = ref.Sim(int(g["res"]), float(g["dx"]), float(g["dt"]), verbose_bgeo=verbose)
    sim.add_particles("jelly", gp[0], gp[1], g["in_x"][:300], g["in_v"][:300], g["in_F"][:300], g["in_B"][:300])
    sim.substep(2)
    path = str(tmp_path / "f.bgeo")
    sim.write_bgeo(path)
    d = sim.download()
    sim.close()
    n = len(d["x"])
    want = bgeo.encode(d["x"], d["v"], d["id"], verbose=verbose, mass=np.full(n, gp[0], np.float32),
                       debug=np.tile(np.array([0, 4, 0], np.float32), (n, 1)), B=d["B"])  # JellyParticle::get_debug_info
    got = open(path, "rb").read()
    if not verbose:
        assert got == want
        return
    # verbose rows: every byte equal except the last-place rounding of apic_frobenius_norm (the reference's loop may
    # contract a*a + s into an FMA; numpy does not): that column within 2 ulp, everything else identical
    assert len(got) == len(want)
    a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
    diff = np.nonzero(a != b)[0]
    width = 23 * 4
    head = want.index(np.asarray(d["x"][np.argsort(d["id"])][0], ">f4").tobytes())
    assert np.all((diff - head) % width // 4 == 22) and np.all(diff >= head) and np.all(diff < head + n * width)
    fa = np.frombuffer(got[head:head + n * width], ">f4").reshape(n, 23)[:, 22]
    fb = np.frombuffer(want[head:head + n * width], ">f4").reshape(n, 23)[:, 22]
    assert np.abs(fa - fb).max() <= 3e-7 * np.abs(fb).max()


# ---------------------------------------------------------------------------------------------- CPIC rigid coupling
def test_live_reference_reproduces_its_cpic_fixture():
    """tests/golden/ref_cpic.npz is what the compiled reference (src/rigid_transfer.cpp, src/mpm_rigid_body.cpp, rigid
    branches of src/transfer.cpp) produces for the seeded scenes of tests/cpic_scenes.py"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so is not built")
    from tests import cpic_scenes as cs
    refmpm.set_threads(1)
    g = np.load(os.path.join(HERE, "golden", "ref_cpic.npz"))
    for name, body, material, n, cfg in cs.CASES[:2] + cs.CASES[-1:]:
        sim, rid = cs.build_reference(refmpm, body, material, **cfg)
        np.testing.assert_array_equal(cs.rigid_vector(sim.rigid_state(rid)), g[name + "_body0"])
        sim.substep(n)
        p = sim.download(by_id=True)
        np.testing.assert_array_equal(p["x"], g[name + "_x"])
        np.testing.assert_array_equal(p["v"], g[name + "_v"])
        np.testing.assert_array_equal(cs.rigid_vector(sim.rigid_state(rid)), g[name + "_body"])


def test_rigid_body_shim_mass_properties_are_the_textbook_ones():
    """the rigid body under the compiled reference is OURS (oracle/taichi_shim/.../rigid_body_shim.h): a solid box and a
    square shell must have their closed-form mass and inertia, whatever the mesh's position before recentring"""
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so is not built")
    from tests import cpic_scenes as cs
    hx, hy, hz, rho = 0.1, 0.06, 0.12, 400.0
    sim = refmpm.Sim(cs.RES, cs.DX, cs.DT)
    rid = sim.add_rigid(cs.box(hx, hy, hz) + np.float32([0.3, -0.2, 0.1]), codimensional=False, density=rho, initial_position=(0.5, 0.5, 0.5))
    st = sim.rigid_state(rid)
    m = rho * 8 * hx * hy * hz
    assert abs(st["mass"] - m) < 1e-5 * m
    np.testing.assert_allclose(np.diag(st["inertia"]), m / 3 * np.array([hy * hy + hz * hz, hx * hx + hz * hz, hx * hx + hy * hy]), rtol=1e-5)
    np.testing.assert_allclose(st["position"], (0.5, 0.5, 0.5), atol=1e-7)  # recentred: the body origin is the centre of mass
    rid = sim.add_rigid(cs.plate(0.2), codimensional=True, density=40.0, initial_position=(0.5, 0.5, 0.5))
    st = sim.rigid_state(rid)
    m = 40.0 * 0.16
    assert abs(st["mass"] - m) < 1e-5 * m
    np.testing.assert_allclose(np.diag(st["inertia"]), m * 0.16 / 12 * np.array([1, 2, 1]), rtol=1e-5)


def test_live_reference_reproduces_its_2d_cpic_fixture():
    from oracle import refmpm
    if not refmpm.available():
        pytest.skip("oracle/_ref/libmpm_ref.so is not built")
    from tests import cpic_scenes as cs
    refmpm.set_threads(1)
    g = np.load(os.path.join(HERE, "golden", "ref_cpic2d.npz"))
    for name, body, material, n, cfg in cs.CASES2:
        sim, rid = cs.build_reference2(refmpm, body, material, **cfg)
        np.testing.assert_array_equal(sim.rigid_state2(rid), g[name + "_body0"])
        sim.substep(n)
        p = sim.download(by_id=True)
        np.testing.assert_array_equal(p["x"], g[name + "_x"])
        np.testing.assert_array_equal(sim.rigid_state2(rid), g[name + "_body"])
