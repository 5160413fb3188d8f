"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU
oracle on identical seeded inputs, against the committed golden fixtures, and — at BASELINE.json's full sizes —
through size-independent invariants.

Tolerances (fp32): after one P2G grid mass rel-L2 <= 1e-6, momentum <= 1e-5 (LDS-atomic order);
after one G2P from an identical grid: x abs <= 1e-7 * L (L=1), v/B rel <= 1e-5, F rel <= 1e-5 (snow/sand 1e-4
near the clamp); after several steps: statistical.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tests.common import lattice_cube, make_state, rel_l2

pytestmark = pytest.mark.gpu

RES, DX, DT = 32, 1.0 / 32, 1e-4
PLANES = [(0.0, 1.0, 0.0, -0.3)]
MATS = ["jelly", "snow", "sand", "water", "linear", "elastic", "von_mises", "visco"]
F_TOL = {"snow": 1e-4, "sand": 1e-4, "von_mises": 1e-4, "visco": 1e-4}


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def make_sim(tm, state, planes=PLANES, friction=0.4, res=RES, dx=DX, dt=DT, **cfg):
    """default ctx mode: apic_b folded into the P2G affine matrix (keep_apic_b=True stores it exactly)"""
    sim = tm.create_simulation3("mpm")
    sim.initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=dt, **cfg))
    if planes:
        ls = tm.mpm.LevelSet(friction=friction)
        for p in planes:
            ls.add_plane(p[:3], d=p[3])
        sim.set_levelset(ls)
    names = {v: k for k, v in tm.MATERIAL_IDS.items()}
    for gi in range(len(state.gtype)):
        m = state.gid == gi
        sim.add_particles(dict(type=names[int(state.gtype[gi])], positions=state.x[m], velocities=state.v[m],
                               F=state.F[m], B=state.B[m], aux=state.aux[m], params=state.gparams[gi]))
    return sim


def ocfg(orc, planes=PLANES, friction=0.4, res=RES, dx=DX, dt=DT, **kw):
    return orc.make_config(res, dx, dt, planes=planes, friction=friction, **kw)


# ------------------------------------------------------------------------------------------ device math
def test_device_svd_matches_oracle(tm, orc):
    """device route (eigen-decomposition of F F^T) vs the oracle's svd convention."""
    rng = np.random.default_rng(0)
    n = 4096
    F = (np.eye(3) + rng.normal(0, 0.3, (n, 3, 3))).astype(np.float32)
    F[:8] = np.eye(3)
    F[8:16] = np.diag([1.0, 1.0, -1.0])  # reflections: the sign goes on the smallest sigma
    sim = make_sim(tm, make_state(lattice_cube(RES, 10, 12, DX), "jelly", DX))
    sim._ensure_ctx()
    U = np.zeros((n, 9), np.float32); S = np.zeros((n, 3), np.float32); V = np.zeros((n, 9), np.float32)
    fp = C.POINTER(C.c_float)
    Fc = np.ascontiguousarray(F.reshape(n, 9))
    sim._check(sim._L.mpmhip_debug_svd3(sim._ctx, n, Fc.ctypes.data_as(fp), U.ctypes.data_as(fp), S.ctypes.data_as(fp),
                                        V.ctypes.data_as(fp)))
    U = U.reshape(n, 3, 3); V = V.reshape(n, 3, 3)
    rec = np.einsum("nij,nj,nkj->nik", U, S, V)
    assert np.abs(rec - F).max() < 2e-5
    assert np.abs(np.einsum("nji,njk->nik", U, U) - np.eye(3)).max() < 1e-5
    assert np.all(np.linalg.det(U) > 0.99)
    for i in list(range(32)) + list(range(100, 400)):
        _, So, _ = orc.svd3(F[i])
        assert np.allclose(np.sort(np.abs(S[i]))[::-1], np.abs(So), atol=2e-5), (i, S[i], So)
        assert np.isclose(np.prod(S[i]), np.prod(So), atol=5e-5 * max(1, abs(np.prod(So))))
    sim.close()


@pytest.mark.parametrize("mat", MATS)
def test_device_constitutive_models_match_oracle(tm, orc, mat):
    rng = np.random.default_rng(1)
    n = 2048
    gp, t = orc.group_params(mat, 400 * DX ** 3 / 8, DX ** 3 / 8)
    F = (np.eye(3) + rng.normal(0, 0.05, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    cdg = (np.eye(3) + rng.normal(0, 0.02, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    aux = {"snow": 1.0 + rng.normal(0, 0.02, n), "water": 1.0 + rng.normal(0, 0.02, n),
           "sand": np.abs(rng.normal(0, 0.01, n))}.get(mat, np.zeros(n)).astype(np.float32)
    sim = make_sim(tm, make_state(lattice_cube(RES, 10, 12, DX), "jelly", DX))
    sim._ensure_ctx()
    fp = C.POINTER(C.c_float)
    out = np.zeros((n, 9), np.float32)
    sim._check(sim._L.mpmhip_debug_force(sim._ctx, t, gp.ctypes.data_as(fp), n, F.ctypes.data_as(fp),
                                         aux.ctypes.data_as(fp), out.ctypes.data_as(fp)))
    ref = np.stack([orc.calculate_force(t, gp, F[i], float(aux[i])).reshape(9) for i in range(n)])
    scale = np.abs(ref).max()
    # stress carries an absolute error ~ 2 mu vol eps_fp32 on both sides (F - R cancellation): compare on that scale
    assert np.abs(out - ref).max() < 3e-5 * scale + 2 * gp[2] * gp[1] * 4e-6, mat
    Fd, auxd = F.copy(), aux.copy()
    sim._check(sim._L.mpmhip_debug_plasticity(sim._ctx, t, gp.ctypes.data_as(fp), n, cdg.ctypes.data_as(fp),
                                              Fd.ctypes.data_as(fp), auxd.ctypes.data_as(fp), None))
    Fo = np.zeros_like(F); auxo = np.zeros_like(aux)
    for i in range(n):
        f, a = orc.plasticity(t, gp, cdg[i], F[i], float(aux[i]))
        Fo[i], auxo[i] = f.reshape(9), a
    if mat == "water":
        Fd = F  # water never updates dg_e (src/particles.cpp:469-478); the kernel does not store it either
    assert np.abs(Fd - Fo).max() < 2e-5, mat
    assert np.abs(auxd - auxo).max() < 2e-5, mat
    # fused kernel path: plasticity + the NEXT substep's calculate_force from one eigen-solve
    Ff, auxf, nf = F.copy(), aux.copy(), np.zeros((n, 9), np.float32)
    sim._check(sim._L.mpmhip_debug_plasticity(sim._ctx, t, gp.ctypes.data_as(fp), n, cdg.ctypes.data_as(fp),
                                              Ff.ctypes.data_as(fp), auxf.ctypes.data_as(fp), nf.ctypes.data_as(fp)))
    if mat == "water":
        Ff = F
    assert np.abs(Ff - Fo).max() < 2e-5 and np.abs(auxf - auxo).max() < 2e-5, mat
    ref2 = np.stack([orc.calculate_force(t, gp, Fo[i], float(auxo[i])).reshape(9) for i in range(n)])
    assert np.abs(nf - ref2).max() < 3e-5 * np.abs(ref2).max() + 2 * gp[2] * gp[1] * 4e-6, mat
    sim.close()


# ------------------------------------------------------------------------------------------ sort
def test_sort_is_a_permutation_in_key_order_and_drops_dead(tm, orc):
    x = lattice_cube(RES, 5, 12, DX, jitter=0.3, seed=2)  # cells 5,6 lie inside the 7-cell deletion margin
    rng = np.random.default_rng(3)
    x = x[rng.permutation(len(x))]
    s = make_state(x, "jelly", DX)
    s.v[5] = np.nan
    s.x[9, 1] = np.inf
    sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT))
    sim._ensure_ctx(extra=len(x))
    sim._groups.append((tm.MATERIAL_IDS["jelly"], s.gparams[0]))
    sim._check(sim._L.mpmhip_add_group(sim._ctx, tm.MATERIAL_IDS["jelly"], s.gparams[0].ctypes.data_as(C.POINTER(C.c_float))))
    sim._upload_new(0, s.x, s.v, s.F, s.B, s.aux)  # bypasses the Python-side near-boundary filter on purpose
    sim._n_added = len(x)
    keep = orc.clear_boundary(ocfg(orc), s)
    assert 0 < keep.sum() < len(x)
    sim.sort_particles_and_populate_grid()
    got = sim.get_particles(sort_by_id=False)
    assert len(got["id"]) == keep.sum()
    assert sorted(got["id"].tolist()) == np.nonzero(keep)[0].tolist()
    assert np.array_equal(got["x"], s.x[got["id"]]) and np.array_equal(got["F"], s.F[got["id"]])
    # the first sort also reorders the records physically (reorder_interval, src/mpm.cpp:811-813): slot order is
    # now key order = Morton(block) << 6 | cell-in-block
    base = np.floor(got["x"].astype(np.float32) * np.float32(1 / DX) - np.float32(0.5)).astype(np.int64)

    def spread(v):
        r = np.zeros_like(v)
        for b in range(10):
            r |= ((v >> b) & 1) << (3 * b)
        return r
    blk = base >> 2
    key = ((spread(blk[:, 0]) << 2 | spread(blk[:, 1]) << 1 | spread(blk[:, 2])) << 6) | ((base[:, 0] & 3) << 4) | \
        ((base[:, 1] & 3) << 2) | (base[:, 2] & 3)
    assert np.all(np.diff(key) >= 0)
    sim.close()


def test_crowded_cells_take_the_side_array_of_the_sort(tm, orc, monkeypatch):
    """k_rank packs (rank, cell) into one word per slot and spills ranks that do not fit into a side array; with the
    test knob (a 3-bit rank field) every cell of this scene spills — the substep must come out exactly as without it,
    in both orders of the slots (long runs -> one atomic per run; shuffled -> the LDS hash path after the first sort).  In the
    deterministic mode (cells in creation-id order behind the sort) bit for bit."""
    monkeypatch.setenv("MPMHIP_DETERMINISTIC", "1")
    x = lattice_cube(RES, 9, 15, DX, jitter=0.3, seed=21)
    x = np.concatenate([x, x + np.float32(1e-3), x - np.float32(1e-3)])  # 24 particles per cell
    rng = np.random.default_rng(22)
    out = {}
    for order in ("runs", "shuffled"):
        xs = x if order == "runs" else x[rng.permutation(len(x))]
        for knob in ("0", "1"):
            monkeypatch.setenv("MPMHIP_TEST_SMALL_RANK", knob)
            s = make_state(xs, "jelly", DX, perturb_F=0.02, seed=23)
            sim = make_sim(tm, s)
            for _ in range(3):
                sim.substep()
            got = sim.get_particles()
            assert len(got["id"]) == len(xs)
            out[order, knob] = got
            sim.close()
        for f in ("x", "v", "F"):
            a, b = out[order, "0"][f], out[order, "1"][f]
            assert np.array_equal(a, b), (order, f, float(np.abs(a - b).max()))
    monkeypatch.delenv("MPMHIP_TEST_SMALL_RANK")
    monkeypatch.delenv("MPMHIP_DETERMINISTIC")


def test_every_form_of_the_sort_gives_the_same_substep(tm, monkeypatch):
    """The library picks the sort's launches by size and grid: k_sort_front + k_cell_table<.., keyed> + k_perm_keyed where the key-indexed
    counters exist (res <= 508), the four launches otherwise (MPMHIP_SORT_V1=1 forces them); 16 / 32 / 64 blocks per chunk of the cell
    table; with the owner list of the grid pass (k_cell_table + k_grid_list) or without (k_cell_table_plain + k_grid_blocks).  All twelve
    combinations on one scene (dense runs + spray + leavers, so dead slots and short runs occur) must agree with the default — in the
    deterministic mode bit for bit (none of the forms changes an operand or an order of the arithmetic), and that mode with the default
    one to the order of the sums inside a cell."""
    rng = np.random.default_rng(41)
    dense = lattice_cube(RES, 8, 14, DX, jitter=0.2, seed=40)
    spray = (rng.uniform(6.0, 26.0, (2500, 3)) * DX).astype(np.float32)
    x = np.concatenate([dense, spray])

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = make_state(x, "jelly", DX, perturb_F=0.02, seed=42)
        s.v[-30:] = (0.0, 0.0, 500.0)  # leave through the wall: their slots drop out of the live range
        sim = make_sim(tm, s)
        for _ in range(4):
            sim.substep()
        got = sim.get_particles()
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
        return got

    loose = run({})
    ref = run({"MPMHIP_DETERMINISTIC": "1"})
    assert len(ref["id"]) < len(x) and np.array_equal(loose["id"], ref["id"])
    for f in ("x", "v", "F"):
        # (same sums in a different order inside a cell: without the mode the ranks are handed out by atomics)
        assert np.allclose(loose[f], ref[f], rtol=0, atol=2e-6 * max(1.0, float(np.abs(ref[f]).max()))), f
    for v1 in ("0", "1"):
        for walk in ("2", "0"):
            for ct in ("16", "32", "64"):
                got = run({"MPMHIP_SORT_V1": v1, "MPMHIP_GRID_WALK": walk, "MPMHIP_CT_BLOCKS": ct, "MPMHIP_DETERMINISTIC": "1"})
                assert np.array_equal(got["id"], ref["id"]), (v1, walk, ct)
                for f in ("x", "v", "F"):
                    assert np.array_equal(got[f], ref[f]), (v1, walk, ct, f, float(np.abs(got[f] - ref[f]).max()))


def test_packed_g2p_walk_equals_the_per_block_walk(tm, monkeypatch):
    """k_g2p_packed (chunks of 256 consecutive sorted positions, whatever blocks they belong to: csrc/k_g2p_packed.h) against k_g2p on the
    same scene: a dense cube (chunks inside one block, tiles reused along a run), spray (a chunk touches more blocks than the
    workgroup keeps tiles for: several passes) and particles that leave the domain (dead slots behind the live range).  The
    library picks the packed walk by size (from 2 M slots on); the knob forces it either way.  Deterministic mode: the two walks put
    different particles into one wave, and since round 6 nothing a particle computes depends on its wave (Jacobi sweeps and the
    refinement of an ill-conditioned F are decided per lane) — the results agree bit for bit."""
    monkeypatch.setenv("MPMHIP_DETERMINISTIC", "1")
    rng = np.random.default_rng(31)
    dense = lattice_cube(RES, 8, 14, DX, jitter=0.2, seed=30)
    spray = (rng.uniform(8.0, 24.0, (3000, 3)) * DX).astype(np.float32)   # ~6 particles per block: 256 positions span ~40 blocks
    x = np.concatenate([dense, spray])
    out = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("MPMHIP_G2P_PACKED", knob)
        s = make_state(x, "sand", DX, perturb_F=0.02, seed=32)
        s.v[-40:] = (0.0, 0.0, 500.0)  # these leave through the wall within a few substeps: deleted, their slots drop out
        sim = make_sim(tm, s)
        for _ in range(5):
            sim.substep()
        out[knob] = sim.get_particles()
        sim.close()
    monkeypatch.delenv("MPMHIP_G2P_PACKED")
    monkeypatch.delenv("MPMHIP_DETERMINISTIC")
    a, b = out["0"], out["1"]
    assert np.array_equal(a["id"], b["id"]) and len(a["id"]) < len(x)
    for f in ("x", "v", "F", "aux"):
        assert np.array_equal(a[f], b[f]), (f, float(np.abs(a[f] - b[f]).max()))


# ------------------------------------------------------------------------------------------ phases
@pytest.mark.parametrize("mat", MATS)
def test_p2g_and_grid_update_match_oracle(tm, orc, mat):
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=4)
    s = make_state(x, mat, DX, perturb_F=0.02, seed=5)
    sim = make_sim(tm, s)
    sim.sort_particles_and_populate_grid()
    sim.rasterize_optimized()
    g0 = sim.get_grid(0)
    cfg = ocfg(orc)
    ref = orc.p2g(cfg, s.copy())
    assert rel_l2(g0[..., 3], ref[..., 3]) <= 1e-6
    assert rel_l2(g0[..., :3], ref[..., :3]) <= 1e-5
    assert np.array_equal(g0[..., 3] != 0, ref[..., 3] != 0)
    sim.normalize_grid_and_apply_boundary_conditions()
    g1 = sim.get_grid(1)
    ref1 = orc.grid_update(cfg, ref.copy())
    assert rel_l2(g1[..., :3], ref1[..., :3]) <= 1e-5
    assert rel_l2(g1[..., 3], ref1[..., 3]) <= 1e-6
    sim.close()


@pytest.mark.parametrize("keep", [True, False], ids=["apic_b_stored", "apic_b_folded"])
@pytest.mark.parametrize("mat", MATS)
def test_g2p_from_identical_grid_matches_oracle(tm, orc, mat, keep):
    """keep=False is the default ctx mode: G2P does not store apic_b; the download recovers it from the P2G affine
    matrix (A - stress S)/(4 m), hence the looser bound on B in that mode."""
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=6)
    s = make_state(x, mat, DX, perturb_F=0.02, seed=7)
    cfg = ocfg(orc)
    ref = s.copy()
    grid = orc.grid_update(cfg, orc.p2g(cfg, ref))
    sim = make_sim(tm, s, keep_apic_b=keep)
    sim.sort_particles_and_populate_grid()
    sim.rasterize_optimized()                       # builds the tile / owner structure
    sim.normalize_grid_and_apply_boundary_conditions()
    sim.set_grid(grid)                              # identical grid on both sides
    sim.resample_optimized()
    got = sim.get_particles()
    orc.g2p(cfg, ref, grid)
    assert np.abs(got["x"] - ref.x).max() <= 1e-7
    assert rel_l2(got["v"], ref.v) <= 1e-5
    assert rel_l2(got["B"], ref.B) <= (1e-5 if keep else 3e-4)
    assert rel_l2(got["F"], ref.F) <= F_TOL.get(mat, 1e-5)
    assert np.abs(got["aux"] - ref.aux).max() <= 2e-5
    sim.close()


@pytest.mark.parametrize("mat", MATS)
def test_full_substep_matches_oracle(tm, orc, mat):
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=8)
    s = make_state(x, mat, DX, perturb_F=0.02, seed=9)
    sim = make_sim(tm, s)
    sim.substep()
    sim.synchronize()
    got = sim.get_particles()
    ref = s.copy()
    orc.substep(ocfg(orc), ref)
    assert len(got["x"]) == ref.n
    assert np.array_equal(got["id"], ref.ids)
    assert np.abs(got["x"] - ref.x).max() <= 2e-7
    assert rel_l2(got["v"], ref.v) <= 2e-5
    assert rel_l2(got["F"], ref.F) <= F_TOL.get(mat, 2e-5)
    sim.close()


def _angular_momentum(x, v, B, mass, dx):
    """sum_p x_p x m v_p + m eps : B_conv^T with B_conv = -dx * apic_b (tests/test_oracle_substep.py has the derivation)"""
    Bc = -dx * B.reshape(-1, 3, 3).astype(np.float64)
    eps = np.zeros((3, 3, 3))
    eps[0, 1, 2] = eps[1, 2, 0] = eps[2, 0, 1] = 1
    eps[0, 2, 1] = eps[2, 1, 0] = eps[1, 0, 2] = -1
    return np.cross(x.astype(np.float64), mass * v.astype(np.float64)).sum(0) + mass * np.einsum("ijk,pkj->i", eps, Bc)


@pytest.mark.parametrize("spin_only", [False, True], ids=["stirred", "spin_only"])
def test_apic_transfers_conserve_angular_momentum_on_the_device(tm, spin_only):
    """no oracle involved: the HIP P2G and G2P against the conservation law that defines APIC transfers — particles ->
    grid -> particles keeps the total angular momentum including the affine part (F = I: no stress; no gravity, no
    boundary).  The spin-only state (v = 0, one skew apic_b for all) makes the affine term the whole signal."""
    x = lattice_cube(RES, 10, 20, DX, jitter=0.3, seed=5)
    s = make_state(x, "jelly", DX, perturb_F=0.0, vel_scale=3.0)
    mass = float(s.gparams[0, 0])
    if spin_only:
        w = np.array([0.4, -1.1, 0.7])
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        s.v[:] = 0
        s.B[:] = (-(DX / 4.0) * Wx).reshape(1, 9).astype(np.float32)  # apic_b = -B_conv / dx, B_conv = (dx^2 / 4) [w]_x
    sim = make_sim(tm, s, planes=None, gravity=(0, 0, 0), clean_boundary=False, keep_apic_b=True)
    L_p = _angular_momentum(s.x, s.v, s.B, mass, DX)
    if spin_only:
        assert np.allclose(L_p, s.n * mass * (DX * DX / 4.0) * 2.0 * w, rtol=1e-6)
    sim.sort_particles_and_populate_grid()
    sim.rasterize_optimized()
    g = sim.get_grid(0)  # raw (m v, m) sums
    nx = RES + 1
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(nx), np.arange(nx), indexing="ij")
    X = np.stack([ii, jj, kk], -1).reshape(-1, 3) * DX
    L_g = np.cross(X, g[..., :3].reshape(-1, 3).astype(np.float64)).sum(0)
    scale = max(np.abs(np.cross(s.x.astype(np.float64), mass * s.v.astype(np.float64))).sum(), np.abs(L_p).sum())
    assert np.allclose(L_g, L_p, atol=1e-5 * scale), (L_g, L_p)
    sim.normalize_grid_and_apply_boundary_conditions()
    sim.resample_optimized()
    got = sim.get_particles()  # ids = creation order = the order of s
    L_back = _angular_momentum(s.x, got["v"], got["B"], mass, DX)  # at the positions the transfer used
    assert np.allclose(L_back, L_g, atol=1e-5 * scale), (L_back, L_g)
    sim.close()


def test_free_rotation_on_the_device_keeps_momentum_angular_momentum_and_energy(tm):
    """oracle-free: a jelly block spinning in free space for 150 substeps of the HIP path — linear momentum, total angular
    momentum (with the affine part), kinetic energy and the turning angle (tests/test_oracle_substep.py runs the same
    scene through the oracle)"""
    x = lattice_cube(RES, 10, 22, DX)
    c = x.mean(0).astype(np.float64)
    s = make_state(x, "jelly", DX, perturb_F=0.0)
    w = np.array([0.0, 0.0, 5.0])
    s.v[:] = np.cross(w, x.astype(np.float64) - c).astype(np.float32)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    s.B[:] = (-(DX / 4.0) * Wx).reshape(1, 9).astype(np.float32)
    mass = float(s.gparams[0, 0])
    sim = make_sim(tm, s, planes=None, gravity=(0, 0, 0), clean_boundary=False, keep_apic_b=True)

    def state(p):
        L = _angular_momentum(p["x"] - c.astype(np.float32), p["v"], p["B"], mass, DX)
        return (mass * p["v"].astype(np.float64)).sum(0), L, 0.5 * mass * (p["v"].astype(np.float64) ** 2).sum()
    p0, L0, k0 = state(dict(x=s.x, v=s.v, B=s.B))
    sim.run_substeps(150)
    got = sim.get_particles()
    p1, L1, k1 = state(got)
    assert len(got["x"]) == s.n and np.abs(p1 - p0).max() <= 1e-5 * mass * np.abs(s.v).sum()
    assert np.allclose(L1, L0, rtol=1e-4, atol=1e-6 * np.abs(L0).max())
    assert abs(k1 - k0) <= 0.02 * k0
    assert np.abs(got["x"].astype(np.float64).mean(0) - c).max() <= 1e-5
    r0, r1 = x[0].astype(np.float64) - c, got["x"][0].astype(np.float64) - c
    ang = 150 * DT * w[2]
    assert abs((np.arctan2(r1[1], r1[0]) - np.arctan2(r0[1], r0[0])) - ang) <= 0.03 * ang
    sim.close()


@pytest.mark.parametrize("mat", ["jelly", "elastic", "snow", "linear"])
def test_compressed_block_expands_and_stretched_block_contracts_on_the_device(tm, mat):
    """oracle-free sign check of the stress term in the HIP P2G by its physical effect"""
    x = lattice_cube(RES, 11, 21, DX)
    c = x.mean(0)
    for stretch, sign in ((0.97, +1.0), (1.03, -1.0)):
        s = make_state(x, mat, DX, perturb_F=0.0)
        s.v[:] = 0
        s.B[:] = 0
        s.F[:] = (np.eye(3) * stretch).reshape(1, 9)
        sim = make_sim(tm, s, planes=None, gravity=(0, 0, 0), clean_boundary=False)
        sim.run_substeps(3)
        got = sim.get_particles()
        radial = ((got["x"] - c) * got["v"]).sum(1)
        far = np.linalg.norm(got["x"] - c, axis=1) > 0.1
        r = sign * radial[far].astype(np.float64)
        assert r.mean() > 0 and (r > 0).mean() > 0.8 and -r[r < 0].sum() < 0.05 * r[r > 0].sum(), (mat, stretch)
        sim.close()


CONFIG_VARIANTS = {
    "apic_damping": dict(apic_damping=0.3),  # scene scripts set one of them, e.g. scripts/mls-cpic/goo_blocks.py:14
    "rpic_damping": dict(rpic_damping=0.2),
    "both_dampings": dict(apic_damping=0.3, rpic_damping=0.1),
    "grid_gravity": dict(particle_gravity=False),  # gravity applied at the grid nodes (src/mpm.cpp:281-293,526-530)
    "skew_gravity+no_clean": dict(gravity=(1.5, -9.0, 0.5), clean_boundary=False),
}


@pytest.mark.parametrize("keep", [True, False], ids=["keep_b", "fold_b"])
@pytest.mark.parametrize("variant", sorted(CONFIG_VARIANTS))
def test_config_variants_match_oracle_over_three_substeps(tm, orc, variant, keep):
    """the config keys of MPM::initialize (src/mpm.cpp:26-75) that change the substep's arithmetic: APIC / RPIC damping
    (the intended damp_affine_momemtum, src/mpm.h:465-469 — SURVEY quirk 3), grid-side gravity, a general gravity
    vector; damping acts on apic_b, so both storage modes of apic_b are covered"""
    kw = CONFIG_VARIANTS[variant]
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=81)
    s = make_state(x, "jelly", DX, perturb_F=0.02, seed=82, vel_scale=2.0)
    sim = make_sim(tm, s, keep_apic_b=keep, **kw)
    cfg = ocfg(orc, **kw)
    ref = s.copy()
    for _ in range(3):
        sim.substep()
        orc.substep(cfg, ref)
    got = sim.get_particles()
    assert len(got["x"]) == ref.n and np.array_equal(got["id"], ref.ids)
    assert np.abs(got["x"] - ref.x).max() <= 5e-7
    assert rel_l2(got["v"], ref.v) <= 5e-5 and rel_l2(got["F"], ref.F) <= 5e-5
    assert rel_l2(got["B"], ref.B) <= (5e-5 if keep else 2e-3)
    if "damping" in variant:  # the damping did something: an undamped run differs
        plain = s.copy()
        for _ in range(3):
            orc.substep(ocfg(orc), plain)
        assert rel_l2(plain.B, ref.B) > 1e-2
    sim.close()


GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "substep_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_against_committed_golden_fixture(tm, path):
    """no oracle involved: committed vectors (tests/golden/make_golden.py)."""
    g = np.load(path)
    names = {v: k for k, v in tm.MATERIAL_IDS.items()}
    mat = names[int(g["gtype"][0])]
    sim = tm.create_simulation3("mpm").initialize(dict(res=(int(g["res"]),) * 3, delta_x=float(g["dx"]),
                                                       base_delta_t=float(g["dt"])))
    ls = tm.mpm.LevelSet(friction=float(g["friction"]))
    for p in g["planes"]:
        ls.add_plane(p[:3], d=float(p[3]))
    sim.set_levelset(ls)
    sim.add_particles(dict(type=mat, positions=g["in_x"], velocities=g["in_v"], F=g["in_F"], B=g["in_B"],
                           aux=g["in_aux"], params=g["gparams"][0]))
    sim.sort_particles_and_populate_grid()
    sim.rasterize_optimized()
    g0 = sim.get_grid(0)
    nz = g["nz"].astype(int)
    sel = g0[nz[:, 0], nz[:, 1], nz[:, 2]]
    assert rel_l2(sel[:, 3], g["p2g_nz"][:, 3]) <= 1e-6 and rel_l2(sel[:, :3], g["p2g_nz"][:, :3]) <= 1e-5
    assert np.count_nonzero(g0[..., 3]) == len(nz)
    sim.normalize_grid_and_apply_boundary_conditions()
    sim.resample_optimized()
    got = sim.get_particles()
    assert np.abs(got["x"] - g["out_x"]).max() <= 2e-7
    assert rel_l2(got["v"], g["out_v"]) <= 2e-5
    assert rel_l2(got["F"], g["out_F"]) <= F_TOL.get(mat, 2e-5)
    for _ in range(4):
        sim.substep()
    sim.synchronize()
    got5 = sim.get_particles()
    assert np.array_equal(got5["id"], g["out5_ids"])
    assert np.abs(got5["x"] - g["out5_x"]).max() <= 1e-6
    assert rel_l2(got5["v"], g["out5_v"]) <= 1e-3
    sim.close()


# ------------------------------------------------------------------------------------------ multi-step / mixed
def test_twenty_steps_statistics_two_materials(tm, orc):
    """chaotic divergence is expected over many steps: compare statistics."""
    xa = lattice_cube(RES, 9, 14, DX, jitter=0.2, seed=10)
    xb = lattice_cube(RES, 15, 20, DX, jitter=0.2, seed=11)
    sa = make_state(xa, "jelly", DX, E=1e4, perturb_F=0.0, vel_scale=0.3)
    sb = make_state(xb, "sand", DX, perturb_F=0.0, vel_scale=0.3)
    s = orc.State(np.concatenate([sa.x, sb.x]), np.concatenate([sa.v, sb.v]), np.concatenate([sa.B, sb.B]),
                  np.concatenate([sa.F, sb.F]), np.concatenate([sa.aux, sb.aux]),
                  np.concatenate([np.zeros(sa.n, np.int32), np.ones(sb.n, np.int32)]),
                  np.stack([sa.gparams[0], sb.gparams[0]]), np.array([sa.gtype[0], sb.gtype[0]], np.int32))
    sim = make_sim(tm, s, friction=-1.0)
    cfg = ocfg(orc, friction=-1.0)
    ref = s.copy()
    for _ in range(20):
        orc.substep(cfg, ref)
    sim.run_substeps(20)
    sim.synchronize()
    got = sim.get_particles()
    assert len(got["x"]) == ref.n and np.array_equal(got["id"], ref.ids)
    assert np.isclose(sim.get_current_time(), 20 * DT, rtol=1e-5)
    assert np.abs(got["x"] - ref.x).max() < 2e-5
    assert np.allclose(got["x"].mean(0), ref.x.mean(0), atol=1e-6)
    mass = s.gparams[ref.gid, 0].astype(np.float64)
    mom_g = (mass[:, None] * got["v"]).sum(0); mom_r = (mass[:, None] * ref.v).sum(0)
    assert np.allclose(mom_g, mom_r, rtol=1e-4, atol=1e-4 * np.abs(mass[:, None] * ref.v).sum())
    ke_g = 0.5 * (mass * (got["v"].astype(np.float64) ** 2).sum(1)).sum()
    ke_r = 0.5 * (mass * (ref.v.astype(np.float64) ** 2).sum(1)).sum()
    assert np.isclose(ke_g, ke_r, rtol=1e-3)
    sim.close()


def test_step_semantics_match_reference_loop(tm):
    """MPM::step (src/mpm.cpp:428-439): dt<0 => one substep; else substep while t + base_dt < request_t."""
    s = make_state(lattice_cube(RES, 12, 14, DX), "jelly", DX)
    sim = make_sim(tm, s, planes=None)
    sim.step(-1)
    assert np.isclose(sim.get_current_time(), DT)
    t, req, n = np.float32(DT), np.float32(DT) + np.float32(1e-3), 0
    while t + np.float32(DT) < req:
        t += np.float32(DT); n += 1
    sim.step(1e-3)
    assert np.isclose(sim.get_current_time(), float(t), rtol=1e-6) and n in (9, 10)
    sim.close()


# ------------------------------------------------------------------------------------------ edge cases
def test_edge_cases_empty_single_ragged_and_capacity(tm, orc):
    # empty simulation: substeps are no-ops
    sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, max_particles=64))
    sim.run_substeps(3)
    sim.synchronize()
    assert sim.get_num_particles() == 0
    # a single particle, then a ragged add (n not a multiple of the wave size) on the same ctx (forces a grow)
    one = make_state(np.array([[0.5, 0.5, 0.5]], np.float32), "snow", DX, perturb_F=0.0)
    sim.add_particles(dict(type="snow", positions=one.x, velocities=one.v, params=one.gparams[0]))
    sim.substep(); sim.synchronize()
    assert sim.get_num_particles() == 1
    x = lattice_cube(RES, 10, 13, DX, jitter=0.2, seed=12)[:197]
    rag = make_state(x, "jelly", DX)
    sim.add_particles(dict(type="jelly", positions=rag.x, velocities=rag.v, F=rag.F, params=rag.gparams[0]))
    sim.substep(); sim.synchronize()
    assert sim.get_num_particles() == 198
    p = sim.get_particles()
    assert np.all(np.isfinite(p["x"])) and set(p["gid"].tolist()) == {0, 1}
    sim.close()
    # raw C ABI: capacity overflow and bad group are reported, not crashed
    sim2 = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, max_particles=100))
    sim2._ensure_ctx()
    fp = C.POINTER(C.c_float)
    rc = sim2._L.mpmhip_add_particles(sim2._ctx, 3, 10, rag.x.ctypes.data_as(fp), None, None, None, None)
    assert rc == -1 and b"unknown group" in sim2._L.mpmhip_last_error(sim2._ctx)
    gi = sim2._L.mpmhip_add_group(sim2._ctx, tm.MATERIAL_IDS["jelly"], rag.gparams[0].ctypes.data_as(fp))
    assert gi == 0
    big = np.tile(rag.x, (8, 1))
    rc = sim2._L.mpmhip_add_particles(sim2._ctx, 0, len(big), big.ctypes.data_as(fp), None, None, None, None)
    assert rc == -4 and b"capacity" in sim2._L.mpmhip_last_error(sim2._ctx)
    assert sim2._L.mpmhip_add_group(sim2._ctx, 9, rag.gparams[0].ctypes.data_as(fp)) == -1  # unknown material id
    assert sim2._L.mpmhip_p2g(sim2._ctx) == -1  # needs a sort first
    sim2.close()


def test_a_block_table_overflow_is_reported_and_leaves_the_ctx_inside_its_arrays(tm, monkeypatch):
    """More active blocks than max_blocks: sticky error bit 1, reported at the next synchronising call as ECAPACITY — also when the
    caller keeps stepping first.  In both forms of the sort: with key-indexed counters the blocks that found no slot keep their
    counts (nobody walks their rows), which must not carry a later sort's permutation outside its array (k_perm_keyed's bound)."""
    rng = np.random.default_rng(61)
    x = (rng.uniform(7.5, RES - 7.5, (6000, 3)) * DX).astype(np.float32)  # ~200 blocks of the 512 the grid has
    for v1 in ("0", "1"):
        monkeypatch.setenv("MPMHIP_SORT_V1", v1)
        s = make_state(x, "jelly", DX)
        sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, max_particles=8192, max_blocks=64))
        sim.add_particles(dict(type="jelly", positions=s.x, velocities=s.v, F=s.F, params=s.gparams[0]))
        sim._ensure_ctx()  # (particles added before the first step are staged on the host until the ctx exists)
        L, ctx = sim._L, sim._ctx
        for _ in range(6):  # (a caller that does not look at the return codes: the substep may already report it, and keeps launching)
            assert L.mpmhip_substep(ctx) in (0, -4)
        rc = L.mpmhip_synchronize(ctx)
        msg = L.mpmhip_last_error(ctx)
        assert rc == -4 and b"exceed max_blocks" in msg, (v1, rc, msg)
        sim.close()
    monkeypatch.delenv("MPMHIP_SORT_V1")


def test_particles_leaving_the_domain_are_deleted_like_the_reference(tm, orc):
    """clear_boundary_particles (src/mpm.cpp:582-633): near-wall particles vanish; clean_boundary=False keeps them."""
    x = lattice_cube(RES, 8, 11, DX, jitter=0.1, seed=13)
    s = make_state(x, "jelly", DX, perturb_F=0.0, vel_scale=0.0)
    s.v[:] = (-30.0, 0.0, 0.0)  # moves 0.1 cells per step towards the x=0 wall
    s.B[:] = 0
    for clean in (True, False):
        sim = make_sim(tm, s, planes=None, gravity=(0, 0, 0), clean_boundary=clean)
        cfg = ocfg(orc, planes=[], gravity=(0, 0, 0), clean_boundary=clean)
        ref = s.copy()
        for _ in range(15):
            orc.substep(cfg, ref)
        sim.run_substeps(15)
        sim.synchronize()
        got = sim.get_particles()
        assert len(got["x"]) == ref.n
        assert np.array_equal(got["id"], ref.ids)
        if clean:
            assert ref.n < s.n
        else:
            assert ref.n == s.n
        sim.close()


# ------------------------------------------------------------------------------------------ BASELINE sizes
def _config_run(tm, res, cells, mat, steps, **matkw):
    dx = 1.0 / res
    lo = res // 2 - cells // 2
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=1e-4))
    sim.set_levelset(tm.mpm.LevelSet(friction=-1.0).add_plane((0, 1, 0), d=-0.1))
    sim.add_particles(dict(type=mat, cube=(lo, lo + cells), **matkw))
    n = sim.get_num_particles()
    sim.sort_particles_and_populate_grid()
    sim.rasterize_optimized()
    g0 = sim.get_grid(0)
    mass = 400.0 * dx ** 3 / 8
    # P2G conserves mass and momentum: sum_i m_i = N m ; sum_i (m v)_i = N m g dt (particles at rest + gravity)
    assert np.isclose(g0[..., 3].sum(dtype=np.float64), n * mass, rtol=1e-6)
    mom = g0[..., :3].reshape(-1, 3).sum(0, dtype=np.float64)
    assert np.allclose(mom, [0, n * mass * -10.0 * 1e-4, 0], atol=1e-5 * n * mass * 1e-3)
    # undeformed lattice at rest: every touched node gets exactly v = g dt after normalisation
    sim.normalize_grid_and_apply_boundary_conditions()
    g1 = sim.get_grid(1)
    act = g1[..., 3] > 0
    assert np.allclose(g1[act][:, 1], -1e-3, rtol=1e-4) and np.abs(g1[act][:, [0, 2]]).max() < 1e-7
    sim.resample_optimized()
    sim.run_substeps(steps)
    sim.synchronize()
    p = sim.get_particles(sort_by_id=False)
    assert len(p["x"]) == n and len(np.unique(p["id"])) == n
    assert np.all(np.isfinite(p["x"])) and np.all(np.isfinite(p["F"]))
    t = (steps + 1) * 1e-4
    assert np.allclose(p["v"][:, 1].mean(), -10.0 * t, rtol=2e-3)         # free fall of the centre of mass
    assert np.abs(np.linalg.det(p["F"].reshape(-1, 3, 3)) - 1).max() < 1e-2
    sim.close()
    return n


def test_config2_128cubed_1M_jelly_invariants(tm):
    """BASELINE config C2: 128^3 grid, 50^3 cells x 8 = 1 000 000 fixed-corotated particles."""
    assert _config_run(tm, 128, 50, "jelly", 5) == 1000000


def test_config3_256cubed_8M_sand_invariants(tm):
    """BASELINE config C3: 256^3 grid, 100^3 cells x 8 = 8 000 000 Drucker-Prager sand particles."""
    assert _config_run(tm, 256, 100, "sand", 3) == 8000000


def test_config5_512cubed_64M_water_and_elastic_invariants(tm):
    """BASELINE config C5 on ONE GPU: 512^3 sparse blocked grid, 8 clusters of 100^3 cells x 8 = 64 000 000 particles
    (4 water, 4 Hencky-elastic).  Size-independent properties: particle count, P2G mass/momentum conservation on the
    sparse block structure, finite state after stepping, free fall of the centre of mass."""
    import ctypes as C

    import psutil
    if psutil.virtual_memory().available < 48 << 30:
        pytest.skip("needs ~48 GB of host memory for the 64 M-particle staging buffers")
    from taichi_mpm_amd.mpm import F_AUX, F_V
    res, cells = 512, 100
    dx = 1.0 / res
    n_expected = 8 * cells ** 3 * 8
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=1e-4,
                                                       max_particles=n_expected + 1024, reorder_interval=0))
    k = 0
    for ox in (78, 334):
        for oy in (78, 334):
            for oz in (78, 334):
                sim.add_particles(dict(type="water" if k % 2 == 0 else "elastic", cube_lo=(ox, oy, oz), cube_cells=cells))
                k += 1
    n = sim.get_num_particles()
    assert n == n_expected
    sim.sort_particles_and_populate_grid()
    sim.rasterize_optimized()
    g0 = sim.get_grid(0)
    mass = 400.0 * dx ** 3 / 8
    assert np.isclose(g0[..., 3].sum(dtype=np.float64), n * mass, rtol=1e-6)
    mom = g0[..., :3].reshape(-1, 3).sum(0, dtype=np.float64)
    assert np.allclose(mom, [0, n * mass * -10.0 * 1e-4, 0], atol=1e-5 * n * mass * 1e-3)
    assert int((g0[..., 3] > 0).sum()) == 8 * 103 ** 3  # base cells 77..177 per axis -> nodes 77..179
    del g0
    sim.normalize_grid_and_apply_boundary_conditions()
    sim.resample_optimized()
    steps = 3
    sim.run_substeps(steps)
    sim.synchronize()
    prof = sim.profile()
    assert prof["particles"] == n and prof["active_blocks"] == 8 * 26 ** 3  # cells 78..177 -> base 77..177: 26 blocks
    v = np.zeros((n, 3), np.float32)
    assert sim._check(sim._L.mpmhip_download(sim._ctx, F_V, v.ctypes.data_as(C.c_void_p), n)) == n
    assert np.all(np.isfinite(v))
    assert np.allclose(v[:, 1].mean(dtype=np.float64), -10.0 * (steps + 1) * 1e-4, rtol=2e-3)
    del v
    aux = np.zeros(n, np.float32)
    assert sim._check(sim._L.mpmhip_download(sim._ctx, F_AUX, aux.ctypes.data_as(C.c_void_p), n)) == n
    assert np.all(np.isfinite(aux)) and aux.max() <= 1.001  # water j stays ~1 in free fall; elastic aux = 0
    sim.close()


# ------------------------------------------------------------------------------------------ level-set shapes
SHAPE_SCENES = {
    "ball": dict(shapes=[(1, 0, 0.45, 0.2, 0.45, 0.12)], friction=0.3),
    "box": dict(shapes=[(2, 0, 0.2, 0.1, 0.2, 0.5, 0.31, 0.5)], friction=-1.0),
    "container": dict(shapes=[(2, 1, 0.27, 0.27, 0.27, 0.56, 0.56, 0.56)], friction=-2.0),
    "plane+ball": dict(planes=[(0.0, 1.0, 0.0, -0.3)], shapes=[(1, 1, 0.4, 0.4, 0.4, 0.2)], friction=0.5),
}


def _levelset(tm, planes=(), shapes=(), friction=-1.0):
    ls = tm.mpm.LevelSet(friction=friction)
    for p in planes:
        ls.add_plane(p[:3], d=p[3])
    for sh in shapes:
        if sh[0] == 1:
            ls.add_sphere(sh[2:5], sh[5], bool(sh[1]))
        else:
            ls.add_cuboid(sh[2:5], sh[5:8], bool(sh[1]))
    return ls


@pytest.mark.parametrize("collide", [False, True], ids=["grid_bc", "grid_bc+particle_collision"])
@pytest.mark.parametrize("scene", sorted(SHAPE_SCENES))
def test_levelset_shapes_and_particle_collision_match_oracle(tm, orc, scene, collide):
    """spheres / cuboids / containers as grid boundary (src/mpm.cpp:296-372) and particle_collision_resolution
    (:414-426) against the oracle, five substeps"""
    sc = SHAPE_SCENES[scene]
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=51)
    s = make_state(x, "jelly", DX, perturb_F=0.02, seed=52, vel_scale=2.0)
    sim = make_sim(tm, s, planes=None, particle_collision=collide)
    ls = _levelset(tm, sc.get("planes", ()), sc["shapes"], sc["friction"])
    sim.set_levelset(ls)
    cfg = orc.make_config(RES, DX, DT, planes=ls.planes, friction=sc["friction"], shapes=ls.non_planes,
                          particle_collision=collide)
    ref = s.copy()
    for _ in range(5):
        sim.substep()
        orc.substep(cfg, ref)
    got = sim.get_particles()
    assert len(got["x"]) == ref.n and np.array_equal(got["id"], ref.ids)
    assert np.abs(got["x"] - ref.x).max() <= 1e-6
    assert rel_l2(got["v"], ref.v) <= 1e-4
    assert rel_l2(got["F"], ref.F) <= 1e-4
    moved = np.abs(ref.v - s.v).max()
    assert moved > 1e-3  # the boundary did something in this scene
    sim.close()


@pytest.mark.parametrize("scene", sorted(SHAPE_SCENES))
def test_delete_particles_inside_level_set_matches_oracle(tm, orc, scene):
    """general_action 'delete_particles_inside_level_set' (src/mpm.cpp:962-974): exactly the particles with phi < 0 go,
    before the first substep and again in the middle of a run (keys are rebuilt), and the run continues like the
    oracle's on the survivors"""
    sc = SHAPE_SCENES[scene]
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=71)
    s = make_state(x, "jelly", DX, perturb_F=0.02, seed=72, vel_scale=2.0)
    sim = make_sim(tm, s, planes=None)
    ls = _levelset(tm, sc.get("planes", ()), sc["shapes"], sc["friction"])
    sim.set_levelset(ls)
    cfg = orc.make_config(RES, DX, DT, planes=ls.planes, friction=sc["friction"], shapes=ls.non_planes)
    ref = s.copy()
    total = 0
    for rounds in range(2):
        keep = orc.delete_inside_levelset(cfg, ref)
        n_before = sim.get_num_particles()
        assert sim.general_action(dict(action="delete_particles_inside_level_set")) == ""
        assert n_before - sim.get_num_particles() == int((~keep).sum())
        total += int((~keep).sum())
        ref = ref.select(keep)
        for _ in range(3):
            sim.substep()
            orc.substep(cfg, ref)
        if rounds == 0:  # push everything 1.5 cells towards the solid so that the second call has work to do
            shift = np.array([0.0, -1.5 * DX, 0.0], np.float32) if scene != "container" else np.array([1.5 * DX, 0, 0], np.float32)
            got = sim.get_particles(sort_by_id=False)
            sim.upload(tm.mpm.F_X, got["x"] + shift)
            order = np.argsort(got["id"])
            assert np.array_equal(got["id"][order], ref.ids)
            ref.x = (got["x"] + shift).astype(np.float32)[order]  # bit-identical positions for the second decision
    got = sim.get_particles()
    assert total > 0 and len(got["x"]) == ref.n and np.array_equal(got["id"], ref.ids)
    assert np.abs(got["x"] - ref.x).max() <= 1e-6 and rel_l2(got["v"], ref.v) <= 1e-4
    sim.close()


# ------------------------------------------------------------------------------------------ calculate_energy
def test_calculate_energy_matches_numpy(tm, orc):
    """MPM<dim>::calculate_energy (src/mpm.cpp:1078-1110): grid kinetic energy after P2G + particle potential energy
    (linear :323-327, jelly :400-407, elastic :785-796); other types have no potential_energy() in the reference."""
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=61)
    parts = [make_state(x[i::3], m, DX, perturb_F=0.05, seed=62 + i) for i, m in enumerate(("jelly", "linear", "elastic"))]
    s = orc.State(np.concatenate([p.x for p in parts]), np.concatenate([p.v for p in parts]),
                  np.concatenate([p.B for p in parts]), np.concatenate([p.F for p in parts]),
                  np.concatenate([p.aux for p in parts]), np.concatenate([np.full(p.n, i, np.int32) for i, p in enumerate(parts)]),
                  np.concatenate([p.gparams for p in parts]), np.concatenate([p.gtype for p in parts]))
    sim = make_sim(tm, s, planes=None, particle_gravity=False)
    kin, pot = sim.calculate_energy()
    g = orc.p2g(ocfg(orc, planes=(), particle_gravity=False), s.copy()).astype(np.float64)
    m = g[..., 3]
    ref_kin = (0.5 * (g[..., :3] ** 2).sum(-1)[m > 0] / m[m > 0]).sum()
    F = s.F.reshape(-1, 3, 3).astype(np.float64)
    U, sig, Vt = np.linalg.svd(F)
    R = U @ Vt
    J = np.linalg.det(F)
    gp = s.gparams[s.gid].astype(np.float64)
    vol, mu, la = gp[:, 1], gp[:, 2], gp[:, 3]
    e_j = vol * (mu * ((F - R) ** 2).sum((1, 2)) + 0.5 * la * (J - 1) ** 2)
    eps = 0.5 * (F + F.transpose(0, 2, 1)) - np.eye(3)
    e_l = vol * (mu * (eps ** 2).sum((1, 2)) + 0.5 * la * np.trace(eps, axis1=1, axis2=2) ** 2)
    ls = np.log(sig)
    e_e = vol * (mu * (ls ** 2).sum(1) + 0.5 * la * ls.sum(1) ** 2)
    ref_pot = np.where(s.gid == 0, e_j, np.where(s.gid == 1, e_l, e_e)).sum()
    assert np.isclose(kin, ref_kin, rtol=1e-5)
    assert np.isclose(pot, ref_pot, rtol=1e-4)
    assert np.isclose(float(sim.general_action(dict(action="calculate_energy"))), kin + pot)
    sim.close()
    sand = make_state(x[:64], "sand", DX)
    sim = make_sim(tm, sand, planes=None)
    with pytest.raises(tm.mpm.MPMError, match="potential_energy"):
        sim.calculate_energy()
    sim.close()


# ------------------------------------------------------------------------------------------ snapshots
def test_snapshot_restart_continues_the_run(tm, orc, tmp_path):
    """general_action save / load (src/mpm.cpp:940-960): a restarted run continues where the saved one stopped —
    the blob carries the raw records incl. the P2G affine matrices, so the restart does not go through the apic_b
    recovery and differs only by the in-cell summation order."""
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=71)
    sa, sb = make_state(x[::2], "snow", DX, seed=72), make_state(x[1::2], "sand", DX, seed=73)
    s = orc.State(np.concatenate([sa.x, sb.x]), np.concatenate([sa.v, sb.v]), np.concatenate([sa.B, sb.B]),
                  np.concatenate([sa.F, sb.F]), np.concatenate([sa.aux, sb.aux]),
                  np.concatenate([np.zeros(sa.n, np.int32), np.ones(sb.n, np.int32)]),
                  np.concatenate([sa.gparams, sb.gparams]), np.concatenate([sa.gtype, sb.gtype]))
    a = make_sim(tm, s)
    a.run_substeps(6)
    path = str(tmp_path / "snap.bin")
    a.frame = 3
    assert a.general_action(dict(action="save", file_name=path)) == ""
    a.run_substeps(6)
    ref = a.get_particles()
    b = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT))
    ls = tm.mpm.LevelSet(friction=0.4)
    for p in PLANES:
        ls.add_plane(p[:3], d=p[3])
    b.set_levelset(ls)
    assert b.general_action(dict(action="load", file_name=path)) == ""
    assert b.frame == 3 and b.get_num_particles() == s.n
    assert np.isclose(b.get_current_time(), 6 * DT, rtol=1e-5)
    b.run_substeps(6)
    got = b.get_particles()
    assert np.array_equal(got["id"], ref["id"]) and np.array_equal(got["gid"], ref["gid"])
    assert np.abs(got["x"] - ref["x"]).max() <= 1e-6
    assert rel_l2(got["v"], ref["v"]) <= 1e-5 and rel_l2(got["F"], ref["F"]) <= 1e-4
    assert np.isclose(b.get_current_time(), a.get_current_time(), rtol=1e-6)
    with open(path, "r+b") as f:  # a corrupted blob is rejected, not loaded
        f.seek(8); f.write(b"XXXX")
    with pytest.raises(tm.mpm.MPMError, match="snapshot"):
        b.general_action(dict(action="load", file_name=path))
    a.close(); b.close()


# ------------------------------------------------------------------------------------------ long runs / reorder
def test_physical_reorder_does_not_change_the_run(tm, orc):
    """sort_allocator (src/mpm.cpp:752-768, every reorder_interval substeps): moving the records into sorted order
    mid-run must not change the simulation — with the cells in creation-id order (deterministic mode) not by a bit: where a
    record lies never enters the arithmetic."""
    x = lattice_cube(RES, 9, 17, DX, jitter=0.2, seed=81)
    s = make_state(x, "snow", DX, perturb_F=0.02, seed=82, vel_scale=4.0)
    outs = []
    for interval in (0, 1, 3):
        sim = make_sim(tm, s, reorder_interval=interval, deterministic=True)
        sim.run_substeps(10)
        outs.append(sim.get_particles())
        sim.close()
    for o in outs[1:]:
        assert np.array_equal(o["id"], outs[0]["id"])
        for f in ("x", "v", "F", "aux"):
            assert np.array_equal(o[f], outs[0][f]), f


@pytest.mark.parametrize("mat", ["sand", "water", "snow", "visco"])
def test_column_collapse_stays_sane_for_400_substeps(tm, mat):
    """a block dropped on a frictional floor inside a container: after 400 substeps every particle is still there,
    finite, inside the container and above the floor, nothing moves faster than free fall + sound, and the centre
    of mass has gone down"""
    x = lattice_cube(RES, 10, 18, DX, jitter=0.15, seed=91)
    s = make_state(x, mat, DX, perturb_F=0.0, vel_scale=0.0)
    s.v[:] = 0
    s.B[:] = 0
    sim = make_sim(tm, s, planes=None, particle_collision=True)
    ls = tm.mpm.LevelSet(friction=0.3).add_plane((0, 1, 0), d=-0.28).add_cuboid((0.24, 0.2, 0.24), (0.76, 0.9, 0.76), True)
    sim.set_levelset(ls)
    y0 = s.x[:, 1].mean()
    sim.run_substeps(400)
    sim.synchronize()
    p = sim.get_particles()
    assert len(p["x"]) == s.n
    for k in ("x", "v", "F", "aux"):
        assert np.all(np.isfinite(p[k])), k
    assert p["x"][:, 1].min() >= 0.28 - 1e-4
    assert (p["x"][:, [0, 2]] >= 0.24 - 1e-4).all() and (p["x"][:, [0, 2]] <= 0.76 + 1e-4).all()
    assert np.abs(p["v"]).max() < 20.0
    assert p["x"][:, 1].mean() < y0 - 1e-3
    sim.close()


def test_growing_the_ctx_keeps_clocks_and_results(tm, orc):
    """adding particles beyond the capacity re-creates the ctx: current_t, the residual of step()'s request_t and the
    particle state carry over, so the run equals one that had the capacity from the start"""
    xa = lattice_cube(RES, 9, 13, DX, jitter=0.2, seed=91)
    xb = lattice_cube(RES, 14, 18, DX, jitter=0.2, seed=92)
    sa, sb = make_state(xa, "jelly", DX, seed=93), make_state(xb, "sand", DX, seed=94)

    def run(cap):
        sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, max_particles=cap, keep_apic_b=True, deterministic=True))
        sim.add_particles(dict(type="jelly", positions=sa.x, velocities=sa.v, F=sa.F, B=sa.B, aux=sa.aux, params=sa.gparams[0]))
        sim.step(3.5 * DT)  # 3 substeps, residual 0.5 dt stays in request_t
        sim.add_particles(dict(type="sand", positions=sb.x, velocities=sb.v, F=sb.F, B=sb.B, aux=sb.aux, params=sb.gparams[0]))
        sim.step(1.6 * DT)  # request_t = 5.1 dt -> two more substeps
        out = sim.get_particles()
        t = sim.get_current_time()
        sim.close()
        return out, t
    small, t_small = run(sa.n + 8)      # must grow at the second add_particles
    big, t_big = run(sa.n + sb.n + 64)  # never grows
    assert np.isclose(t_small, 5 * DT, rtol=1e-6) and t_small == t_big
    assert np.array_equal(small["id"], big["id"]) and np.array_equal(small["gid"], big["gid"])
    for f in ("x", "v", "F", "B", "aux"):  # (deterministic mode: the two runs agree bit for bit)
        assert np.array_equal(small[f], big[f]), f


def test_benchmark_rasterize_and_resample_are_bounded_rounds_that_leave_the_state_alone(tm, capsys):
    """the reference's performance harness (src/mpm.cpp:516-523, 554-561: with benchmark_rasterize / benchmark_resample the substep
    loops forever over `Timer("Rasterize x 20"); base_delta_t = 0; 20 x rasterize_optimized`): here one bounded round — as a
    method, and through the config keys before the first substep — after which the run continues from the untouched state"""
    res, dx = 64, 1.0 / 64
    x = lattice_cube(res, 20, 44, dx, jitter=0.2, seed=3)
    s = make_state(x, "sand", dx, perturb_F=0.02, seed=4, vel_scale=1.0)

    def scene(**cfg):
        sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=1e-4, deterministic=True, **cfg))
        sim.add_particles(dict(type="sand", positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
        return sim
    a, b, c = scene(), scene(), scene(benchmark_rasterize=True, benchmark_resample=True)
    a.run_substeps(3); b.run_substeps(3)
    t = b.get_current_time()
    r1, r2 = b.benchmark_rasterize(), b.benchmark_resample(rounds=10)
    assert r1["name"] == "Rasterize x 20" and r2["name"] == "Resample x 10" and r1["particles"] == s.n
    assert 0.001 < r1["ns_per_particle"] < 50 and 0.001 < r2["ns_per_particle"] < 50
    assert b.get_current_time() == t
    a.run_substeps(2); b.run_substeps(2)
    pa, pb = a.get_particles(), b.get_particles()
    assert np.array_equal(pa["id"], pb["id"])
    for f in ("x", "v", "F", "aux"):  # (two runs in the deterministic mode: bit for bit)
        assert np.array_equal(pa[f], pb[f]), f
    capsys.readouterr()
    c.substep()   # config keys: the two rounds run once, in front of the first substep
    out = capsys.readouterr().out
    assert "Rasterize x 20:" in out and "Resample x 20:" in out and "ns per particle" in out
    c.run_substeps(4)
    pc = c.get_particles()
    assert np.array_equal(pa["x"], pc["x"]) and np.array_equal(pa["F"], pc["F"])
    capsys.readouterr()
    c.substep()
    assert capsys.readouterr().out == ""
    for sim in (a, b, c):
        sim.close()
