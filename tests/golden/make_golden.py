"""Generates tests/golden/substep_*.npz, ref_materials.npz, ref_kernels.npz and ref_shapes.npz from the REFERENCE's own
solver (oracle/_ref/libmpm_ref.so = /root/reference/src/{mpm,transfer,particles}.cpp compiled where they lie, see
oracle/Makefile: ref_mpm).  Run in the container that has /root/reference, from the repo root:
    make -C oracle ref_mpm && python tests/golden/make_golden.py
Round 1 wrote these fixtures from the restated oracle ("parity unpinned"); since round 2 they hold REFERENCE output, so
that the restated oracle (tests/test_golden_cpu.py) and the HIP library (tests/test_gpu_parity.py,
tests/test_gpu_ref.py) are both checked against the reference's arithmetic on any box, without the reference tree.
What the compiled reference cannot pin is stated in oracle/taichi_shim/taichi/common/util.h (svd / polar_decomp are
the shim's, in double precision; the level set is analytic)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (only for the parameter rows / type ids the fixtures carry)
from oracle import refmpm as ref  # noqa: E402
from tests.common import MAT_KW, lattice_cube, make_state  # noqa: E402

RES, DX, DT = 32, 1.0 / 32, 1e-4
PLANES = [(0.0, 1.0, 0.0, -0.3)]
FRICTION = 0.4
MATS = ("jelly", "snow", "sand", "water", "linear", "elastic", "von_mises", "visco")


def ref_sim(s, mat, shapes, friction, shapes1=None, t1=1.0, **cfg):
    sim = ref.Sim(RES, DX, DT, shapes=shapes, friction=friction, **cfg)
    if shapes1 is not None:
        sim.set_levelset(shapes, friction, shapes1=shapes1, t0=0.0, t1=t1)
    gp = s.gparams[0]
    sim.add_particles(mat, gp[0], gp[1], s.x, s.v, s.F, s.B, s.aux, **MAT_KW.get(mat, {}))
    return sim


def substep_fixture(here, mat):
    x = lattice_cube(RES, 9, 15, DX, jitter=0.2, seed=11)
    s = make_state(x, mat, DX, perturb_F=0.02, seed=12, **MAT_KW.get(mat, {}))
    shapes = [(0, 0) + tuple(p) for p in PLANES]
    sim = ref_sim(s, mat, shapes, FRICTION)
    sim.sort()
    sim.p2g(optimized=True)       # rasterize_optimized -> block_op_normal, src/transfer.cpp:467-569
    g_p2g = sim.download_grid()
    sim.grid_update()             # src/mpm.cpp:277-372
    g_upd = sim.download_grid()
    sim.g2p(optimized=True)       # resample_optimized -> block_op_normal, src/transfer.cpp:837-954
    a = sim.download()
    sim.close()
    nz = np.argwhere(g_p2g[..., 3] != 0)
    # the generic (optimized=False) path on the same input: src/transfer.cpp:193-278, :585-687
    gen = ref_sim(s, mat, shapes, FRICTION, optimized=False)
    gen.sort(); gen.p2g(optimized=False)
    gg = gen.download_grid()
    gen.grid_update(); gen.g2p(optimized=False)
    ga = gen.download()
    gen.close()
    b = ref_sim(s, mat, shapes, FRICTION)
    b.substep(5)                  # MPM<3>::substep, src/mpm.cpp:452-575
    b5 = b.download()
    b.close()
    np.savez_compressed(
        os.path.join(here, "substep_%s.npz" % mat), res=RES, dx=DX, dt=DT, planes=np.array(PLANES, np.float32),
        friction=FRICTION, gparams=s.gparams, gtype=s.gtype, source="reference (oracle/_ref/libmpm_ref.so)",
        in_x=s.x, in_v=s.v, in_B=s.B, in_F=s.F, in_aux=s.aux,
        nz=nz.astype(np.int16), p2g_nz=g_p2g[nz[:, 0], nz[:, 1], nz[:, 2]], upd_nz=g_upd[nz[:, 0], nz[:, 1], nz[:, 2]],
        out_x=a["x"], out_v=a["v"], out_B=a["B"], out_F=a["F"], out_aux=a["aux"],
        gen_p2g_nz=gg[nz[:, 0], nz[:, 1], nz[:, 2]], gen_x=ga["x"], gen_v=ga["v"], gen_B=ga["B"], gen_F=ga["F"],
        out5_x=b5["x"], out5_v=b5["v"], out5_F=b5["F"], out5_aux=b5["aux"], out5_ids=b5["id"])
    print(mat, s.n, "particles,", len(nz), "nodes")


def materials_fixture(here):
    """calculate_force / plasticity / get_allowed_dt / potential_energy of every registered particle type
    (src/particles.cpp) on seeded deformation gradients: small strains, large strains, and states past the yield
    surfaces / clamps"""
    rng = np.random.default_rng(2024)
    n = 384
    out = {}
    vol = DX ** 3 / 8
    mass = 400.0 * vol
    for mat in MATS:
        kw = MAT_KW.get(mat, {})
        amp = np.repeat([0.01, 0.05, 0.2], n // 3)[:, None, None]
        F = (np.eye(3) + rng.normal(0, 1, (n, 3, 3)) * amp).astype(np.float32).reshape(n, 9)
        cdg = (np.eye(3) + rng.normal(0, 1, (n, 3, 3)) * amp * 0.3).astype(np.float32).reshape(n, 9)
        aux = {"snow": 1.0 + rng.normal(0, 0.05, n), "water": 1.0 + rng.normal(0, 0.05, n),
               "sand": np.abs(rng.normal(0, 0.02, n)), "visco": np.full(n, 1000.0)}.get(mat, np.zeros(n)).astype(np.float32)
        if mat in ("sand", "elastic", "von_mises"):  # Hencky models take log(sigma): keep det F > 0 (reference: NaN otherwise)
            bad = np.linalg.det(F.reshape(n, 3, 3)) <= 0.05
            F[bad] = np.eye(3, dtype=np.float32).reshape(9)
        v = rng.normal(0, 1, (n, 3)).astype(np.float32)
        force = ref.calculate_force(mat, mass, vol, F, aux, **kw)
        F2, aux2 = ref.plasticity(mat, mass, vol, cdg, F, aux, **kw)
        force2 = ref.calculate_force(mat, mass, vol, F2, aux2, **kw)
        adt, pot = ref.particle_scalars(mat, mass, vol, F, aux, v, DX, **kw)
        gp, t = orc.group_params(mat, mass, vol, **kw)
        out.update({mat + "_F": F, mat + "_cdg": cdg, mat + "_aux": aux, mat + "_v": v, mat + "_force": force,
                    mat + "_F2": F2, mat + "_aux2": aux2, mat + "_force2": force2, mat + "_allowed_dt": adt,
                    mat + "_potential": pot, mat + "_gp": gp, mat + "_type": t})
    np.savez_compressed(os.path.join(here, "ref_materials.npz"), mass=mass, vol=vol, dx=DX, **out)
    print("materials:", n, "states x", len(MATS), "types")


def illcond_states(seed=77):
    """deformation gradients F = U diag(sigma) V^T with prescribed singular values: condition numbers 1 .. 1e4 in four
    patterns (spread / two large / two small and nearly equal / two small and exactly equal), repeated and nearly repeated
    singular values around 1, singular values at and below sand's 1e-4 clamp (src/particles.cpp:603-604), and det F < 0
    (one singular value negated: the convention of the reference's svd puts the sign on the smallest).  Returns
    (F [n, 9] float32, cond [n], tag [n] (index into TAGS), negdet [n] bool); shared by the generator and the tests."""
    rng = np.random.default_rng(seed)

    def rot():
        q, r = np.linalg.qr(rng.normal(0, 1, (3, 3)))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        return q

    rows = []  # (sigma triple, cond, tag, negdet)
    for c in (1.0, 10.0, 1e2, 1e3, 1e4):
        for tag, sig in ((0, (1.0, c ** -0.5, 1.0 / c)), (1, (1.0, 1.0, 1.0 / c)), (2, (1.0, 1.001 / c, 1.0 / c)),
                         (3, (c ** (2 / 3), c ** (-1 / 3), c ** (-1 / 3)))):
            for scale in (1.0, 0.8, 1.3):
                for _ in range(2):
                    rows.append((np.array(sig) * scale, c, tag, False))
        for tag, sig in ((0, (1.0, c ** -0.5, 1.0 / c)), (1, (1.0, 1.0, 1.0 / c))):
            if c <= 1e2:
                for _ in range(3):
                    rows.append((np.array(sig), c, tag, True))
    for sig in ((2.0, 2.0, 1.0), (1.0, 1.0, 1.0 + 1e-6), (1.0, 1.0, 1.0), (1.0 + 1e-6, 1.0, 1.0 - 1e-6), (1.5, 1.5, 1.5)):
        for _ in range(3):
            rows.append((np.array(sig), max(sig) / min(sig), 4, False))
    for sig in ((1.0, 0.5, 1e-4), (1.0, 0.5, 5e-5), (1.0, 1e-4, 1e-4), (1.0, 2e-4, 0.9e-4)):
        for _ in range(3):
            rows.append((np.array(sig), max(sig) / min(sig), 5, False))
    F, cond, tag, neg = [], [], [], []
    for sig, c, t, n in rows:
        sg = np.array(sig, np.float64)
        if n:
            sg[np.argmin(sg)] *= -1.0
        F.append((rot() @ np.diag(sg) @ rot().T).reshape(9))
        cond.append(c); tag.append(t); neg.append(n)
    return np.array(F, np.float32), np.array(cond), np.array(tag, np.int32), np.array(neg, bool)


ILLCOND_TAGS = ("spread", "two_large", "two_small_close", "two_small_equal", "repeated", "sand_clamp")


def illcond_fixture(here):
    """calculate_force / plasticity of every particle type (src/particles.cpp:207-242,391-416,599-647,701-732,786-812) on
    ILL-CONDITIONED deformation gradients (illcond_states): where do the device's tolerances stop holding?  The reference
    evaluates svd(F) (here: the shim's double-precision Jacobi, rounded to float); the device takes sigma from an
    eigen-solve of F F^T in fp32, refined on F itself when the condition number asks for it (csrc/mpm_math.h)."""
    F, cond, tag, neg = illcond_states()
    n = len(F)
    rng = np.random.default_rng(78)
    cdg = (np.eye(3) + rng.normal(0, 0.02, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    vol = DX ** 3 / 8
    mass = 400.0 * vol
    out = dict(F=F, cdg=cdg, cond=cond, tag=tag, negdet=neg)
    # singular values of the float32 matrices as they are, in double precision (descending, sign on the last: det F < 0)
    sv = np.linalg.svd(F.reshape(n, 3, 3).astype(np.float64), compute_uv=False)
    sv[:, 2] *= np.sign(np.linalg.det(F.reshape(n, 3, 3).astype(np.float64)))
    out["sigma"] = sv
    for mat in MATS:
        kw = MAT_KW.get(mat, {})
        aux = {"snow": np.ones(n), "water": np.ones(n), "visco": np.full(n, 1000.0)}.get(mat, np.zeros(n)).astype(np.float32)
        use = ~neg if mat in ("sand", "elastic", "von_mises") else np.ones(n, bool)  # Hencky: log of a negative sigma is NaN
        with np.errstate(all="ignore"):
            force = ref.calculate_force(mat, mass, vol, F, aux, **kw)
            F2, aux2 = ref.plasticity(mat, mass, vol, cdg, F, aux, **kw)
            force2 = ref.calculate_force(mat, mass, vol, F2, aux2, **kw)
        gp, t = orc.group_params(mat, mass, vol, **kw)
        out.update({mat + "_aux": aux, mat + "_use": use, mat + "_force": force, mat + "_F2": F2, mat + "_aux2": aux2,
                    mat + "_force2": force2, mat + "_gp": gp, mat + "_type": t})
    np.savez_compressed(os.path.join(here, "ref_illcond.npz"), mass=mass, vol=vol, dx=DX, tags=np.array(ILLCOND_TAGS), **out)
    print("illcond:", n, "states x", len(MATS), "types; cond", sorted(set(cond.tolist()))[:3], "...", cond.max())


def kernels_fixture(here):
    """MPMKernel<3,2>, MPMFastKernel32, MPMKernel<2,2> (src/kernel.h) and friction_project (src/mpm_fwd.h:25-57)"""
    rng = np.random.default_rng(7)
    pos = (rng.uniform(3, 20, (64, 3))).astype(np.float32)
    inv_dx = 32.0
    slow = np.stack([ref.kernel3_dw_w(p, inv_dx, fast=False) for p in pos])
    fast = np.stack([ref.kernel3_dw_w(p, inv_dx, fast=True) for p in pos])
    k2 = np.stack([ref.kernel2_dw_w(p[:2], inv_dx) for p in pos])
    start = np.array([ref.stencil_start(float(p[0])) for p in pos], np.int32)
    fr_in, fr_out = [], []
    for mu in (-1.0, -2.0, -2.4, 0.0, 0.3, 1.5):
        for _ in range(12):
            v, vb = rng.normal(0, 1, 3).astype(np.float32), rng.normal(0, 0.3, 3).astype(np.float32)
            n = rng.normal(0, 1, 3); n = (n / np.linalg.norm(n)).astype(np.float32)
            fr_in.append(np.concatenate([v, vb, n, [mu]]).astype(np.float32))
            fr_out.append(ref.friction_project(v, vb, n, mu))
    np.savez_compressed(os.path.join(here, "ref_kernels.npz"), pos=pos, inv_dx=inv_dx, slow=slow, fast=fast, k2=k2, start=start,
                        friction_in=np.stack(fr_in), friction_out=np.stack(fr_out))
    print("kernels:", len(pos), "positions;", len(fr_in), "friction cases")


def shapes_fixture(here):
    """level-set shapes (plane + sphere + inside-out cuboid container), a MOVING plane and sphere
    (boundary_velocity = -dphi/dt n dx, src/mpm.cpp:323-342), particle_collision (src/mpm.cpp:414-426,566-569)
    and the config variants of MPM::initialize; three substeps each"""
    cases = {
        "static_shapes": dict(shapes=[(0, 0, 0, 1, 0, -0.3), (1, 0, 0.5, 0.3, 0.5, 0.12), (2, 1, 0.2, 0.2, 0.2, 0.8, 0.8, 0.8)],
                              friction=0.3, cfg={}),
        "slip_sphere": dict(shapes=[(1, 0, 0.5, 0.33, 0.5, 0.1)], friction=-2.3, cfg={}),
        # DynamicLevelSet(t0 = 0, t1, levelset(t0), levelset(t1)): a floor rising at 0.8 m/s, a ball moving at
        # (0.3, 1.0, -0.2) m/s, and the shrinking container of scripts/async/balls.py:38-49
        "moving_plane": dict(shapes=[(0, 0, 0, 1, 0, -0.3)], shapes1=[(0, 0, 0, 1, 0, -0.308)], t1=0.01, friction=0.4, cfg={}),
        "moving_sphere": dict(shapes=[(1, 0, 0.5, 0.25, 0.5, 0.1)], shapes1=[(1, 0, 0.503, 0.26, 0.498, 0.1)], t1=0.01,
                              friction=-1.0, cfg={}),
        "shrinking_box": dict(shapes=[(2, 1, 0.26, 0.26, 0.26, 0.50, 0.50, 0.50)], shapes1=[(2, 1, 0.28, 0.26, 0.28, 0.48, 0.50, 0.48)],
                              t1=0.01, friction=-2.0, cfg={}),
        "particle_collision": dict(shapes=[(0, 0, 0, 1, 0, -0.32), (1, 0, 0.5, 0.3, 0.5, 0.1)], friction=0.2,
                                   cfg=dict(particle_collision=True)),
        "grid_gravity": dict(shapes=[(0, 0, 0, 1, 0, -0.3)], friction=0.4, cfg=dict(particle_gravity=False)),
        "apic_damping_only": dict(shapes=[(0, 0, 0, 1, 0, -0.3)], friction=0.4, cfg=dict(apic_damping=0.3)),
        "both_dampings": dict(shapes=[(0, 0, 0, 1, 0, -0.3)], friction=0.4, cfg=dict(apic_damping=0.3, rpic_damping=0.1)),
    }
    out = {}
    x = lattice_cube(RES, 10, 15, DX, jitter=0.2, seed=21)
    for mat in ("jelly", "sand"):
        s = make_state(x, mat, DX, perturb_F=0.02, seed=22, vel_scale=2.0)
        out["in_" + mat] = np.concatenate([s.x, s.v, s.B, s.F, s.aux[:, None]], 1)
        out["gp_" + mat] = s.gparams[0]
        for name, c in cases.items():
            for optimized in (True, False):
                if not optimized and name not in ("static_shapes", "moving_sphere", "apic_damping_only", "both_dampings"):
                    continue  # (the generic path repeats the same boundary code: two cases keep the fixture small)
                sim = ref_sim(s, mat, c["shapes"], c["friction"], shapes1=c.get("shapes1"), t1=c.get("t1", 1.0),
                              optimized=optimized, **c["cfg"])
                sim.substep(3)
                d = sim.download()
                sim.close()
                key = "%s_%s_%s" % (name, mat, "opt" if optimized else "gen")
                out[key] = np.concatenate([d["x"], d["v"], d["F"], d["aux"][:, None]], 1)
                out[key + "_ids"] = d["id"]
    import json
    np.savez_compressed(os.path.join(here, "ref_shapes.npz"), res=RES, dx=DX, dt=DT, cases=json.dumps(
        {k: dict(shapes=[list(map(float, sh)) for sh in c["shapes"]], friction=c["friction"], cfg=c["cfg"],
                 shapes1=[list(map(float, sh)) for sh in c["shapes1"]] if "shapes1" in c else None, t1=c.get("t1", 1.0))
         for k, c in cases.items()}), **out)
    print("shapes:", len(cases), "cases x 2 materials x 2 paths")


def mpm2d_state(res=64, lo=(20, 25), cells=20, seed=5):
    """seeded 2D scene shared by the generator and the tests: 4 particles per cell, stirred velocities, perturbed F"""
    rng = np.random.default_rng(seed)
    dx = 1.0 / res
    cc = np.stack(np.meshgrid(np.arange(lo[0], lo[0] + cells), np.arange(lo[1], lo[1] + cells), indexing="ij"), -1).reshape(-1, 2)
    off = 0.25 * np.array([[-1, -1], [1, -1], [-1, 1], [1, 1]])
    x = ((cc[:, None, :] + 0.5 + off[None]) * dx).reshape(-1, 2) + rng.uniform(-0.1, 0.1, (len(cc) * 4, 2)) * dx
    n = len(x)
    c = x.mean(0)
    v = 3.0 * np.stack([-(x[:, 1] - c[1]), x[:, 0] - c[0]], 1) + rng.normal(0, 0.1, (n, 2))
    F = np.eye(2).reshape(1, 4) + rng.normal(0, 0.02, (n, 4))
    B = rng.normal(0, 0.01, (n, 4))
    return x.astype(np.float32), v.astype(np.float32), F.astype(np.float32), B.astype(np.float32)


MPM2D_CASES = {
    "floor": dict(shapes=[(0, 0, 0, 1, 0, -0.37)], friction=0.4, cfg={}),
    "disc+rising_floor": dict(shapes=[(0, 0, 0, 1, 0, -0.37), (1, 0, 0.5, 0.33, 0, 0.08)],
                              shapes1=[(0, 0, 0, 1, 0, -0.375), (1, 0, 0.5, 0.33, 0, 0.08)], t1=0.01, friction=-1.0, cfg={}),
    "damped+grid_gravity": dict(shapes=[(2, 1, 0.2, 0.2, 0, 0.8, 0.8, 0)], friction=-2.5,
                                cfg=dict(apic_damping=0.2, rpic_damping=0.1, particle_gravity=False)),
}


def mpm2d_fixture(here):
    """MPM<2> of the reference (create_simulation2('mpm'): the generic rasterize / resample, src/transfer.cpp:193-278,585-687)
    for every particle type, three substeps"""
    import json
    res, dt = 64, 1e-4
    dx = 1.0 / res
    x, v, F, B = mpm2d_state(res)
    vol = dx ** 2 / 4
    out = dict(x=x, v=v, F=F, B=B)
    for mat in MATS:
        aux0 = {"snow": 1.0, "water": 1.0, "visco": 1000.0}.get(mat, 0.0)
        aux = np.full(len(x), aux0, np.float32)
        gp, t = orc.group_params(mat, 400.0 * vol, vol, **MAT_KW.get(mat, {}))
        out["gp_" + mat] = gp
        for name, c in MPM2D_CASES.items():
            if name != "floor" and mat not in ("jelly", "sand", "water"):
                continue
            sim = ref.Sim(res, dx, dt, dim=2, shapes=c["shapes"], friction=c["friction"], **c["cfg"])
            if "shapes1" in c:
                sim.set_levelset(c["shapes"], c["friction"], shapes1=c["shapes1"], t0=0.0, t1=c["t1"])
            sim.add_particles(mat, 400.0 * vol, vol, x, v, F, B, aux, **MAT_KW.get(mat, {}))
            sim.substep(3)
            d = sim.download()
            sim.close()
            out["%s_%s" % (name, mat)] = np.concatenate([d["x"], d["v"], d["F"], d["B"], d["aux"][:, None]], 1)
            out["%s_%s_ids" % (name, mat)] = d["id"]
    np.savez_compressed(os.path.join(here, "ref_mpm2d.npz"), res=res, dx=dx, dt=dt, cases=json.dumps(MPM2D_CASES), **out)
    print("mpm2d:", len(x), "particles")


def cpic_fixture(here):
    """CPIC rigid coupling (src/rigid_transfer.cpp, src/mpm_rigid_body.cpp, rigid branches of src/transfer.cpp): per case
    the boundary particles, the colored distance field and the particles' colours after sort + rasterize + gather_cdf,
    then the particles and the body after n whole substeps"""
    from tests import cpic_scenes as cs
    out = {}
    for name, body, material, n, cfg in cs.CASES:
        sim, rid = cs.build_reference(ref, body, material, **cfg)
        out[name + "_body0"] = cs.rigid_vector(sim.rigid_state(rid))
        out[name + "_inertia"] = sim.rigid_state(rid)["inertia"]
        out[name + "_samples"] = sim.rigid_samples(rid)["pos"]
        sim.sort(); sim.rasterize_rigid_boundary()
        st, d = sim.download_cdf()
        nz = np.flatnonzero(st.reshape(-1))
        out[name + "_cdf_idx"], out[name + "_cdf_states"], out[name + "_cdf_dist"] = nz.astype(np.int32), st.reshape(-1)[nz], d.reshape(-1)[nz]
        sim.gather_cdf()
        pc = sim.particle_cdf()
        o = np.argsort(sim.download(by_id=False)["id"], kind="stable")
        out[name + "_p_states"], out[name + "_p_near"] = pc["states"][o], pc["near"][o].astype(np.int8)
        out[name + "_p_dist"], out[name + "_p_normal"] = pc["distance"][o], pc["normal"][o]
        # (the phases above are the first third of a substep; start over for the whole-substep run)
        sim, rid = cs.build_reference(ref, body, material, **cfg)
        sim.substep(n)
        p = sim.download(by_id=True)
        out[name + "_x"], out[name + "_v"], out[name + "_F"] = p["x"], p["v"], p["F"]
        o = np.argsort(sim.download(by_id=False)["id"], kind="stable")
        out[name + "_states"] = sim.particle_cdf()["states"][o]
        out[name + "_body"] = cs.rigid_vector(sim.rigid_state(rid))
        print("cpic", name, len(p["x"]), "particles,", len(nz), "coloured nodes,", int((out[name + "_states"] != 0).sum()), "coloured particles")
    np.savez_compressed(os.path.join(here, "ref_cpic.npz"), **out)


def cpic2d_fixture(here):
    """the same for MPM<2> (generic transfers with the colour test, segments instead of triangles)"""
    from tests import cpic_scenes as cs
    out = {}
    for name, body, material, n, cfg in cs.CASES2:
        sim, rid = cs.build_reference2(ref, body, material, **cfg)
        out[name + "_body0"] = sim.rigid_state2(rid)
        out[name + "_samples"] = sim.rigid_samples2(rid)
        sim.sort(); sim.rasterize_rigid_boundary()
        st, d = sim.download_cdf2()
        nz = np.flatnonzero(st.reshape(-1))
        out[name + "_cdf_idx"], out[name + "_cdf_states"], out[name + "_cdf_dist"] = nz.astype(np.int32), st.reshape(-1)[nz], d.reshape(-1)[nz]
        sim.gather_cdf()
        pc = sim.particle_cdf2()
        o = np.argsort(sim.download(by_id=False)["id"], kind="stable")
        out[name + "_p_states"], out[name + "_p_near"] = pc["states"][o], pc["near"][o].astype(np.int8)
        out[name + "_p_dist"], out[name + "_p_normal"] = pc["distance"][o], pc["normal"][o]
        sim, rid = cs.build_reference2(ref, body, material, **cfg)
        sim.substep(n)
        p = sim.download(by_id=True)
        out[name + "_x"], out[name + "_v"], out[name + "_F"] = p["x"], p["v"], p["F"]
        o = np.argsort(sim.download(by_id=False)["id"], kind="stable")
        out[name + "_states"] = sim.particle_cdf2()["states"][o]
        out[name + "_body"] = sim.rigid_state2(rid)
        print("cpic2d", name, len(p["x"]), "particles,", len(nz), "coloured nodes,", int((out[name + "_states"] != 0).sum()), "coloured particles")
    np.savez_compressed(os.path.join(here, "ref_cpic2d.npz"), **out)


def joints_fixture(here):
    """the reference's joints (src/articulation.cpp) on free bodies: state the joints are set up on, state after the drift,
    state after one MPM::articulate (src/mpm.h:278-319), and the bodies after further whole rounds of
    articulate + advect_rigid_bodies (what a substep does to bodies that touch no particle)"""
    import json
    from tests import cpic_scenes as cs

    def states(sim, n):
        out = np.zeros((n + 1, 33), np.float32)
        out[0, 3] = 1.0  # the background body: at the origin, unrotated, no inverse mass / inertia
        for b in range(1, n + 1):
            st = sim.rigid_state(b)
            out[b] = np.concatenate([st["position"], st["rotation"], st["velocity"], st["angular_velocity"], [st["mass"], st["inv_mass"]],
                                     st["inertia"].ravel(), st["inv_inertia"].ravel()])
        return out
    out = {}
    for name, joints in cs.JOINT_CASES.items():
        sim = ref.Sim(RES, DX, cs.JOINT_DT, gravity=(0, -10, 0))
        for body in cs.JOINT_BODIES:
            kw = dict(body)
            sim.add_rigid(kw.pop("mesh"), **kw)
        for b, (v, w) in enumerate(cs.JOINT_VELOCITIES):
            sim.rigid_set_velocity(b + 1, v=v, w=w)
        nb = len(cs.JOINT_BODIES)
        out[name + "/setup"] = states(sim, nb)
        for j in joints:
            sim.general_action(action="add_articulation", **j)
        for _ in range(cs.JOINT_DRIFT):
            sim.advect_rigid_bodies()
        out[name + "/drifted"] = states(sim, nb)
        ref._chk(ref.lib().ref_phase(sim.h, 10, 1))
        out[name + "/articulated"] = states(sim, nb)
        for _ in range(20):
            sim.advect_rigid_bodies()
            ref._chk(ref.lib().ref_phase(sim.h, 10, 1))
        out[name + "/after20"] = states(sim, nb)
        sim.close()
    np.savez_compressed(os.path.join(here, "ref_joints.npz"), cases=json.dumps(cs.JOINT_CASES), dt=cs.JOINT_DT, **out)
    print("ref_joints.npz:", ", ".join(cs.JOINT_CASES))


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ref.set_threads(1)  # the generic P2G of the reference is racy with more than one thread
    what = sys.argv[1:] or (list(MATS) + ["materials", "illcond", "kernels", "shapes", "mpm2d", "cpic", "cpic2d", "joints"])
    for w in what:
        if w in MATS:
            substep_fixture(here, w)
        elif w == "materials":
            materials_fixture(here)
        elif w == "illcond":
            illcond_fixture(here)
        elif w == "kernels":
            kernels_fixture(here)
        elif w == "shapes":
            shapes_fixture(here)
        elif w == "mpm2d":
            mpm2d_fixture(here)
        elif w == "cpic":
            cpic_fixture(here)
        elif w == "cpic2d":
            cpic2d_fixture(here)
        elif w == "joints":
            joints_fixture(here)


if __name__ == "__main__":
    main()
