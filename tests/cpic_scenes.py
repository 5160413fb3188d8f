"""Seeded CPIC scenes (a rigid body cutting through / pushing a block of particles) shared by the fixture generator
(tests/golden/make_golden.py -> ref_cpic.npz) and the tests."""
import numpy as np

from oracle import oracle as orc

RES, DX, DT = 32, 1.0 / 32, 1e-4
VOL = DX ** 3 / 8
MASS = VOL * 400.0


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


# free bodies
BODIES = {
    # (bodies sit OFF the lattice: a node exactly on a surface gets its side from rounding noise, in any implementation)
    "plate": dict(mesh=plate(), codimensional=True, density=40.0, friction=0.3, initial_position=(0.503, 0.497, 0.501),
                  initial_rotation=(20.0, 0.0, 10.0)),
    "box": dict(mesh=box(), codimensional=False, density=400.0, friction0=0.2, friction1=-1.0, initial_position=(0.5, 0.52, 0.5),
                initial_rotation=(0.0, 30.0, 15.0), initial_velocity=(0.3, -0.5, 0.1), initial_angular_velocity=(0.0, 2.0, 1.0)),
}
# a scripted plate: position(t) = P0 + VEL t + AMP sin(OMEGA t), Euler angles(t) = E0 + RATE t (degrees)
SCRIPT = dict(p0=(0.502, 0.55, 0.497), vel=(0.2, -1.0, 0.0), amp=(0.0, 0.0, 0.02), omega=40.0, e0=(10.0, 0.0, 5.0), rate=(0.0, 90.0, 30.0))
# (case name, body, material, substeps, simulation config)
CASES = [("plate_jelly", "plate", "jelly", 5, dict(penalty=1e3)), ("box_jelly", "box", "jelly", 5, dict(penalty=1e3)),
         ("plate_sand", "plate", "sand", 5, dict(penalty=1e3)), ("box_water", "box", "water", 5, dict(penalty=1e3)),
         ("scripted_plate_jelly", "scripted", "jelly", 8, dict())]


def script_functions():
    """the scripts as float32 python callables (what a scene script hands to add_particles(type='rigid', ...))"""
    f32 = np.float32
    s = SCRIPT

    def pos(t):
        t = f32(t)
        return [f32(s["p0"][k]) + f32(s["vel"][k]) * t + f32(s["amp"][k]) * f32(np.sin(f32(s["omega"]) * t)) for k in range(3)]

    def rot(t):
        t = f32(t)
        return [f32(s["e0"][k]) + f32(s["rate"][k]) * t for k in range(3)]
    return pos, rot


def group_row(material):
    return orc.group_params(material, MASS, VOL)[0]


def build_reference(refmpm, body, material, **cfg):
    x, v = block_of_particles()
    ref = refmpm.Sim(RES, DX, DT, gravity=(0, -10, 0), **cfg)
    if body == "scripted":
        s = SCRIPT
        rid = ref.add_rigid(plate(), script=refmpm.rigid_script(s["p0"], s["vel"], s["amp"], s["omega"], s["e0"], s["rate"]),
                            codimensional=True, friction=0.4)
    else:
        b = dict(BODIES[body])
        rid = ref.add_rigid(b.pop("mesh"), **b)
    ref.add_particles(material, MASS, VOL, x, v)
    return ref, rid


def build_device(tm, body, material, **cfg):
    x, v = block_of_particles()
    sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, gravity=(0, -10, 0),
                                                       max_particles=len(x) + 16, **cfg))
    if body == "scripted":
        pos, rot = script_functions()
        rid = int(sim.add_particles(dict(type="rigid", mesh=plate(), codimensional=True, friction=0.4, scripted_position=pos,
                                         scripted_rotation=rot)))
    else:
        rid = int(sim.add_particles(dict(type="rigid", **BODIES[body])))
    sim.add_particles(dict(type=material, positions=x, velocities=v, params=group_row(material)))
    return sim, rid


def rigid_vector(state):
    """position 3, quaternion 4, velocity 3, angular velocity 3, mass"""
    return np.concatenate([state["position"], state["rotation"], state["velocity"], state["angular_velocity"], [state["mass"]]]).astype(np.float32)


# ---------------------------------------------------------------------------------------------- 2D (MPM<2>)
RES2, DX2 = 64, 1.0 / 64
VOL2 = DX2 ** 2 / 4
MASS2 = VOL2 * 400.0


def bar2(half=0.15):
    return np.array([[[-half, 0.0], [half, 0.0]]], np.float32)


def box2(hx=0.08, hy=0.05):
    c = np.array([[-hx, -hy], [hx, -hy], [hx, hy], [-hx, hy]], np.float32)
    return np.array([[c[i], c[(i + 1) % 4]] for i in range(4)], np.float32)


def block2(lo=22, hi=42, seed=0):
    rng = np.random.default_rng(seed)
    g = np.arange(lo, hi) + 0.25
    X = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    X = np.concatenate([X, X + 0.5]) + rng.uniform(-0.2, 0.2, (2 * len(X), 2))
    return (X * DX2).astype(np.float32), rng.normal(0, 0.3, X.shape).astype(np.float32)


BODIES2 = {
    "bar": dict(mesh=bar2(), codimensional=True, density=40.0, friction=0.3, initial_position=(0.503, 0.497), initial_rotation=20.0),
    "box": dict(mesh=box2(), codimensional=False, density=400.0, friction0=0.2, friction1=-1.0, initial_position=(0.5, 0.52),
                initial_rotation=30.0, initial_velocity=(0.3, -0.5), initial_angular_velocity=2.0),
}
SCRIPT2 = dict(p0=(0.502, 0.55), vel=(0.2, -1.0), a0=10.0, rate=180.0)
CASES2 = [("bar_jelly", "bar", "jelly", 5, dict(penalty=1e3)), ("box_sand", "box", "sand", 5, dict(penalty=1e3)),
          ("scripted_bar_water", "scripted", "water", 8, dict())]


def build_reference2(refmpm, body, material, **cfg):
    x, v = block2()
    ref = refmpm.Sim(RES2, DX2, DT, dim=2, gravity=(0, -10), **cfg)
    if body == "scripted":
        s = SCRIPT2
        rid = ref.add_rigid2(bar2(), script=[1, s["p0"][0], s["p0"][1], s["vel"][0], s["vel"][1], 1, s["a0"], s["rate"]],
                             codimensional=True, friction=0.4)
    else:
        b = dict(BODIES2[body])
        rid = ref.add_rigid2(b.pop("mesh"), **b)
    ref.add_particles(material, MASS2, VOL2, x, v)
    return ref, rid


def build_device2(tm, body, material, **cfg):
    x, v = block2()
    sim = tm.create_simulation2("mpm").initialize(dict(res=(RES2,) * 2, delta_x=DX2, base_delta_t=DT, gravity=(0, -10),
                                                       max_particles=len(x) + 16, **cfg))
    f32 = np.float32
    if body == "scripted":
        s = SCRIPT2
        rid = int(sim.add_particles(dict(type="rigid", mesh=bar2(), codimensional=True, friction=0.4,
                                         scripted_position=lambda t: [f32(s["p0"][k]) + f32(s["vel"][k]) * f32(t) for k in range(2)],
                                         scripted_rotation=lambda t: f32(s["a0"]) + f32(s["rate"]) * f32(t))))
    else:
        rid = int(sim.add_particles(dict(type="rigid", **BODIES2[body])))
    gp = orc.group_params(material, MASS2, VOL2)[0]
    sim.add_particles(dict(type=material, positions=x, velocities=v, params=gp))
    return sim, rid


# ---------------------------------------------------------------------------------------------- joints (src/articulation.cpp)
# three free boxes with sizeable velocities; the joints are set up on these poses, the bodies then drift for JOINT_DRIFT
# substeps (advect_rigid_bodies only) before MPM::articulate is compared — a joint whose anchors still coincide does nothing
JOINT_BODIES = [
    dict(mesh=box(0.10, 0.06, 0.08), codimensional=False, density=40.0, friction=0.3, initial_position=(0.35, 0.50, 0.50),
         initial_rotation=(10.0, 20.0, 30.0)),
    dict(mesh=box(0.07, 0.09, 0.05), codimensional=False, density=60.0, friction=0.3, initial_position=(0.60, 0.52, 0.48),
         initial_rotation=(-15.0, 5.0, 40.0)),
    dict(mesh=box(0.05, 0.05, 0.09), codimensional=False, density=90.0, friction=0.3, initial_position=(0.50, 0.70, 0.55),
         initial_rotation=(0.0, -25.0, 12.0)),
]
JOINT_VELOCITIES = [((0.3, -0.1, 0.2), (1.0, 2.0, -0.5)), ((-0.2, 0.1, 0.0), (-0.7, 0.4, 1.5)), ((0.05, 0.3, -0.15), (0.6, -1.1, 0.8))]
JOINT_DRIFT, JOINT_DT = 40, 1e-3
JOINT_CASES = {
    "rotation": [dict(type="rotation", obj0=1, obj1=2)],
    "frozen": [dict(type="frozen", obj0=1, obj1=2)],
    "distance": [dict(type="distance", obj0=1, obj1=2, offset0=(0.05, 0.0, 0.0), offset1=(-0.03, 0.01, 0.0))],
    "distance_rod": [dict(type="distance", obj0=2, obj1=3, target_distance=0.2, penalty=5e3)],
    "distance_background": [dict(type="distance", obj0=3, obj1=0, offset0=(0.0, 0.02, 0.0), offset1=(0.5, 0.8, 0.5))],
    "axial_rotation": [dict(type="axial_rotation", obj0=1, obj1=2, axis=(0.0, 0.0, 2.0), offset0=(0.1, 0.0, 0.0))],
    "motor": [dict(type="motor", obj0=1, obj1=2, axis=(0.0, 1.0, 0.0), power=3.0, axis_length=0.05)],
    "stepper_background": [dict(type="stepper", obj0=1, obj1=0, axis=(0.0, 0.0, 1.0), angular_velocity=2.0)],
    "chain": [dict(type="rotation", obj0=1, obj1=2), dict(type="axial_rotation", obj0=2, obj1=3, axis=(1.0, 1.0, 0.0)),
              dict(type="distance", obj0=1, obj1=3, offset0=(0.0, 0.03, 0.0)), dict(type="motor", obj0=3, obj1=0, axis=(0.0, 1.0, 0.0), power=-2.0)],
}
