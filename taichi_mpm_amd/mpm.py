"""Host-side mirror of the reference's MPM plugin surface, on top of the C ABI (include/mpmhip.h).

Two layers, as in the reference:
  * `Simulation3D` == the object `tc_core.create_simulation3('mpm')` returns (class MPM<3>, src/mpm.h:56-489):
    `initialize(config)`, `add_particles(config) -> str`, `set_levelset(ls)`, `step(dt)`,
    `get_current_time()`, `general_action(config) -> str`, `test()`, `get_debug_information()`,
    `visualize()`, `get_mpi_world_rank()`, `get_vis_resolution()`, attribute `frame`  (scripts/async/async_mpm.py:25-32,76-287).
  * `MPM` == the Python driver class scene scripts instantiate (`tc.dynamics.MPM(**kwargs)`; the twin that IS in
    the reference tree is AsyncMPM, scripts/async/async_mpm.py:17-300): kwargs config, `add_particles(**kw)`,
    `set_levelset`, `step`, `simulate`.

Configs are flat dicts (reference: taichi `Config`, string->string; `P(**kwargs)`).  Errors raise `MPMError`
(reference: TC_ASSERT / TC_ERROR abort).  Scene tooling the hot path does not need (textures, Poisson-disk sampling,
rendering) is out of scope: `add_particles` takes explicit `positions=` or the built-in `benchmark=` generator
(src/mpm.cpp:149-186) or a `cube=(lo, hi)` lattice; `type='rigid'` takes `mesh=` triangles or `mesh_fn='file.obj'`
(CPIC rigid bodies, `add_rigid_body`), joints come through `general_action(action='add_articulation', ...)`.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

from . import _lib
from .materials import MATERIAL_IDS, group_params, initial_aux

F_X, F_V, F_B, F_F, F_AUX, F_GID, F_ID, F_STATES = range(8)
_WIDTH = {F_X: 3, F_V: 3, F_B: 9, F_F: 9, F_AUX: 1, F_GID: 1, F_ID: 1, F_STATES: 1}


class MPMError(RuntimeError):
    pass


class LevelSet:
    """Analytic stand-in for taichi's LevelSet (scripts/async/async_mpm.py:129-137 `create_levelset`): a union of
    solids — half-spaces, spheres, axis-aligned cuboids — with one friction code (README.md:326-330).  Method names
    and argument order follow the calls the scene scripts make (`add_plane(normal, d)`, `add_sphere(center, radius,
    inside_out)`, `add_cuboid(lower, upper, inside_out)`, `set_friction(f)`)."""

    def __init__(self, friction=-1.0, delta_x=None):
        self.friction = float(friction)
        self.shapes = []  # (type, inside_out, p[6]) — include/mpmhip.h: mpmhip_shape
        self.delta_x = delta_x  # cell size of the grid the driver created it for (MPM.create_levelset)

    def get_delta_x(self):  # (scripts/async/slope.py:69)
        if self.delta_x is None:
            raise MPMError("this LevelSet was not created through MPM.create_levelset(): it has no grid")
        return self.delta_x

    def set_friction(self, f):
        self.friction = float(f)
        return self

    def _add(self, type_, inside_out, p):
        if len(self.shapes) >= _lib.MAX_SHAPES:
            raise MPMError("at most %d level-set shapes are supported" % _lib.MAX_SHAPES)
        p = [float(v) for v in p]
        self.shapes.append((int(type_), int(bool(inside_out)), p + [0.0] * (6 - len(p))))
        return self

    def add_plane(self, normal, d=None, point=None):
        """free space is { x : n.x + d > 0 }.  Either `d` or a `point` on the plane."""
        n = np.asarray(normal, np.float64)
        n = n / np.linalg.norm(n)
        if d is None:
            d = -float(np.dot(n, np.asarray(point, np.float64)))
        return self._add(0, 0, [n[0], n[1], n[2], d])

    def add_sphere(self, center, radius, inside_out=False):
        """solid ball; inside_out=True: the ball is the free space (e.g. scripts/mls-cpic: add_sphere(c, 0.3, True))"""
        c = _vec3(center, None)
        return self._add(1, inside_out, [c[0], c[1], c[2], radius])

    def add_cuboid(self, lower, upper, inside_out=False):
        """solid box; inside_out=True: the box is a container (e.g. add_cuboid((0,.2,.05), (.95,.95,.95), True))"""
        lo, hi = _vec3(lower, None), _vec3(upper, None)
        return self._add(2, inside_out, list(lo) + list(hi))

    @property
    def planes(self):
        return [tuple(p[:4]) for t_, _, p in self.shapes if t_ == 0]

    @property
    def non_planes(self):
        """(type, inside_out, params...) rows in the form oracle.make_config(shapes=...) takes"""
        return [(t_, io) + tuple(p[:4] if t_ == 1 else p) for t_, io, p in self.shapes if t_ != 0]


class DynamicLevelSet:
    """taichi's DynamicLevelSet as the python driver builds it every frame (scripts/async/async_mpm.py:119-127):
    `initialize(t0, t1, levelset(t0), levelset(t1))`; the two key frames are blended linearly in time on the device
    (include/mpmhip.h: mpmhip_set_levelset_keyframes)."""

    def __init__(self):
        self.t0, self.t1, self.levelset0, self.levelset1 = 0.0, 1.0, None, None

    def initialize(self, t0, t1, levelset0, levelset1):
        if not float(t1) > float(t0):
            raise MPMError("DynamicLevelSet needs t0 < t1")
        self.t0, self.t1, self.levelset0, self.levelset1 = float(t0), float(t1), levelset0, levelset1
        return self


def _vec3(v, default):
    if v is None:
        v = default
    if np.isscalar(v):
        return (float(v),) * 3
    v = tuple(float(a) for a in v)
    if len(v) != 3:
        raise MPMError("expected a 3-vector, got %r" % (v,))
    return v


def load_obj_triangles(path):
    """triangles (n, 3, 3) of a Wavefront .obj file (v / f records; polygons are fanned) — what the reference loads through
    taichi's Mesh from `mesh_fn`"""
    verts, tris = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == "f":
                idx = [int(w.split("/")[0]) for w in t[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    tris.append([verts[idx[0]], verts[idx[k]], verts[idx[k + 1]]])
    if not tris:
        raise MPMError("no faces in %s" % path)
    return np.asarray(tris, np.float32)


# config keys of MPM<dim>::initialize / substep that change the physics and are NOT implemented here: a scene that sets them to
# anything but the inert default is refused instead of being simulated differently (key: inert value, where the reference reads it)
UNSUPPORTED_KEYS = {

    "expr_leaky_levelset": (0, "src/mpm.cpp:300"), "gravity_cutting": (False, "src/mpm.cpp:347"),
    "remove_particles": (0, "src/mpm.cpp:586"), "sand_climb": (False, "src/mpm.h:281"), "sand_crawler": (False, "src/mpm.h"),
    "stork_nod": (False, "src/mpm.h"), "coupling_iterations": (1, "src/mpm.cpp:467"), "cdf_expand": (0, "src/rigid_transfer.cpp:82"),
    "energy_experiment": (False, "src/mpm.cpp:68"), "visualize_cdf": (False, "src/mpm.cpp:474"),
    "visualize_particle_cdf": (False, "src/mpm.cpp:488"),
}


def check_unsupported_keys(cfg):
    for k, (inert, where) in UNSUPPORTED_KEYS.items():
        if k in cfg and cfg[k] != inert and not (isinstance(inert, bool) and bool(cfg[k]) == inert):
            raise MPMError("config key %r = %r (%s) is not implemented by this library" % (k, cfg[k], where))


class Simulation3D:
    """MPM<3> (src/mpm.h:56-489) backed by libmpmhip."""

    def __init__(self):
        self._L = _lib.load()
        self._ctx = None
        self._cfg = None
        self._staged = []  # (group params, material, arrays) before the ctx exists
        self._groups = []  # (material id, params) in ctx order
        self._levelset = None
        self.frame = 0
        self.config = {}
        self._n_added = 0
        self._rigids = []  # ctypes keep-alives (config struct, script callbacks) of the rigid bodies, index = body id - 1
        self._free_bodies = 0  # bodies that move under impulses (not scripted, density > 0)
        self._script_error = None  # first exception raised by a scripted_position / scripted_rotation callback

    # ---------------------------------------------------------------- lifecycle
    def initialize(self, config):
        """MPM<dim>::initialize, src/mpm.cpp:26-75 (keys: README.md:234-256)."""
        cfg = dict(config)
        if "delta_t" in cfg:  # src/mpm.cpp:41-42
            raise MPMError("Please use 'base_delta_t' instead of 'delta_t'")
        check_unsupported_keys(cfg)
        if "res" not in cfg:
            raise MPMError("config key 'res' is required")
        res = cfg["res"]
        res = (int(res),) * 3 if np.isscalar(res) else tuple(int(r) for r in res)
        if len(res) != 3:
            raise MPMError("this build implements the 3D path (MPM<3>); res must have 3 entries")
        self.res = res
        self.delta_x = float(cfg.get("delta_x", 1.0 / res[0]))  # python default, async_mpm.py:40-41
        self.base_delta_t = float(cfg.get("base_delta_t", 1e-4)) * float(cfg.get("dt_multiplier", 1.0))
        self.gravity = _vec3(cfg.get("gravity"), (0.0, -10.0, 0.0))
        self.particle_gravity = bool(cfg.get("particle_gravity", True))
        self.apic_damping = float(cfg.get("apic_damping", 0.0))
        self.rpic_damping = float(cfg.get("rpic_damping", 0.0))
        self.clean_boundary = bool(cfg.get("clean_boundary", True))
        self.particle_collision = bool(cfg.get("particle_collision", False))  # src/mpm.cpp:566-569
        self.rigid_body_levelset_collision = bool(cfg.get("rigid_body_levelset_collision", False))  # src/mpm.cpp:535-538
        self.dirichlet = float(cfg.get("dirichlet_boundary_radius", 0.0)) > 0.0   # src/mpm.cpp:541-544 -> :401-412 (3D: y > 0.525 at rest)
        self._bench_keys = tuple(k for k in ("rasterize", "resample") if cfg.get("benchmark_" + k, False))  # src/mpm.cpp:516-523, 554-561
        self.reorder_interval = int(cfg.get("reorder_interval", 1000))  # src/mpm.cpp:45
        # apic_b is only ever consumed through the P2G affine matrix A, which is stored; keeping a separate copy
        # costs 48 of the 180 bytes G2P writes per particle.  keep_apic_b=True stores it (exact downloads of B);
        # otherwise download recovers it from A on demand (include/mpmhip.h: discard_apic_b).
        self.discard_apic_b = not bool(cfg.get("keep_apic_b", False))
        # optimized=False: the reference's generic transfer path (src/mpm.cpp:508-515,546-552).  Same kernels here; the
        # one arithmetic difference of that path, the position clamp of src/transfer.cpp:668-670, is switched on
        self.optimized = bool(cfg.get("optimized", True))
        # bitwise reproducible runs (include/mpmhip.h: mpmhip_config.deterministic): in-cell order by creation id behind every sort
        self.deterministic = bool(cfg.get("deterministic", False))
        self.max_particles = int(cfg.get("max_particles", 0))
        self.max_blocks = int(cfg.get("max_blocks", 0))
        self.device = int(cfg.get("device", 0))
        self.verbose_bgeo = bool(cfg.get("verbose_bgeo", False))  # src/visualize.cpp:22
        self.frame_directory = cfg.get("frame_directory")  # injected by the python driver, async_mpm.py:49
        self.frame_count = 0  # src/mpm.h:334
        self.penalty = float(cfg.get("penalty", 0.0))               # CPIC, src/mpm.cpp:35
        self.pushing_force = float(cfg.get("pushing_force", 20000.0))  # src/mpm.cpp:40
        self.articulation_iterations = int(cfg.get("articulation_iterations", 100))  # src/mpm.h:279-280
        self.config = cfg
        return self

    def _create(self, capacity):
        c = _lib.Config()
        c.res[:] = self.res
        c.dx, c.dt = self.delta_x, self.base_delta_t
        c.gravity[:] = self.gravity
        c.particle_gravity = int(self.particle_gravity)
        c.apic_damping, c.rpic_damping = self.apic_damping, self.rpic_damping
        c.clean_boundary = int(self.clean_boundary)
        c.n_planes = 0  # the level set (any mix of shapes) is installed right after creation
        c.particle_collision = int(self.particle_collision)
        c.max_particles = int(capacity)
        c.max_blocks = self.max_blocks
        c.device = self.device
        c.reorder_interval = self.reorder_interval
        c.discard_apic_b = int(self.discard_apic_b)
        c.generic_path = int(not self.optimized)
        c.deterministic = int(self.deterministic)
        ctx = C.c_void_p()
        rc = self._L.mpmhip_create(C.byref(c), C.byref(ctx))
        if rc != 0:
            raise MPMError("mpmhip_create failed (%d): %s" % (rc, self._L.mpmhip_last_error(None).decode()))
        self._ctx, self._cfg, self._capacity = ctx, c, int(capacity)
        self._check(self._L.mpmhip_set_rigid_coupling(self._ctx, self.penalty, self.pushing_force))
        self._check(self._L.mpmhip_set_articulation_iterations(self._ctx, self.articulation_iterations))
        self._check(self._L.mpmhip_set_dirichlet(self._ctx, int(self.dirichlet)))
        self._check(self._L.mpmhip_set_rigid_levelset_collision(self._ctx, int(self.rigid_body_levelset_collision)))
        self._apply_levelset()
        for mat, params in self._groups:
            self._check(self._L.mpmhip_add_group(self._ctx, mat, params.ctypes.data_as(C.POINTER(C.c_float))))

    def _check(self, rc):
        if rc < 0:
            raise MPMError("libmpmhip error %d: %s" % (rc, self._L.mpmhip_last_error(self._ctx).decode()))
        return rc

    def _ensure_ctx(self, extra=0):
        """create the device context on first use; grow it (download, recreate, re-upload) when full."""
        need = self._n_added + extra
        if self._ctx is None:
            cap = max(self.max_particles, need) if self.max_particles > 0 else max(need, 1024)
            self._create(cap)
            for gi, arrs in self._staged:
                self._upload_new(gi, *arrs)
            self._staged = []
        elif need > self._capacity:
            # in place (include/mpmhip.h: mpmhip_reserve): clocks, stream, rigid bodies and joints, async table, partition
            # and halo boxes all stay — scenes that keep adding particles need no max_particles, with or without bodies
            cap = max(int(need * 1.25), self.max_particles)
            self._check(self._L.mpmhip_reserve(self._ctx, cap))
            self._capacity = cap

    def __del__(self):
        try:
            if self._ctx is not None:
                self._L.mpmhip_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    close = __del__

    # ---------------------------------------------------------------- particles
    def _near_boundary(self, x):  # src/mpm.h:269-276 (applied at creation: src/mpm.cpp:129-132)
        X = x / self.delta_x
        return (X.min(1) < 7.0) | ((X - np.asarray(self.res)).max(1) > -7.0)

    def _upload_new(self, gi, x, v, F, B, aux):
        fp = C.POINTER(C.c_float)

        def ptr(a, w):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32).reshape(len(x), w)
            keep.append(a)
            return a.ctypes.data_as(fp)
        keep = []
        self._check(self._L.mpmhip_add_particles(self._ctx, gi, len(x), ptr(x, 3), ptr(v, 3), ptr(F, 9), ptr(B, 9),
                                                 ptr(aux, 1)))

    def add_particles(self, config):
        """MPM<dim>::add_particles, src/mpm.cpp:77-270.  Returns "" (rigid bodies, which return an id, are out
        of scope)."""
        cfg = dict(config)
        ptype = cfg.get("type")
        if ptype == "rigid":  # src/mpm.cpp:80-83: returns the rigid body's index as a string
            return str(self.add_rigid_body(cfg))
        if ptype not in MATERIAL_IDS:
            raise MPMError("unknown particle type %r" % (ptype,))
        dx = self.delta_x
        maximum = float(cfg.get("ppc", cfg.get("maximum", 8)))
        if cfg.get("benchmark", 0):  # src/mpm.cpp:149-186
            b = int(cfg["benchmark"])
            if len(set(self.res)) != 1:
                raise MPMError("benchmark particles need a cubic grid")
            if b == 125:
                s = 0.1
            elif b == 8000:
                s = 0.4
            else:
                raise MPMError("s must be 125 or 8000")
            lower = int(round(self.res[0] * (0.5 - s)))
            higher = lower + int(round(self.res[0] * 2 * s))
            x = lattice_cube(lower, higher, dx)
            maximum = 1.0  # create_particle(..., 1, config): vol = dx^3 (src/mpm.cpp:178)
        elif "cube" in cfg:
            lo, hi = cfg["cube"]
            x = lattice_cube(int(lo), int(hi), dx)
        elif "cube_lo" in cfg:  # cube of `cube_cells`^3 cells with its lower corner at cell cube_lo = (i, j, k)
            x = lattice_cube(0, int(cfg["cube_cells"]), dx) + (np.asarray(cfg["cube_lo"], np.float64) * dx).astype(np.float32)
        elif "positions" in cfg:
            x = np.ascontiguousarray(cfg["positions"], np.float32).reshape(-1, 3)
        else:
            raise MPMError("add_particles needs one of: benchmark=, cube=(lo,hi), positions= "
                           "(density_tex / point_cloud sampling is scene tooling outside this build)")
        keep = ~self._near_boundary(x.astype(np.float64))  # "particle out of box or near boundary. Ignored."
        x = x[keep]
        n = len(x)
        vol = dx ** 3 / maximum  # src/mpm.cpp:134-135
        mass = vol * float(cfg.get("density", 400.0))
        params, mat = group_params(ptype, mass, vol, **{k: v for k, v in cfg.items() if isinstance(v, (int, float))})
        if "params" in cfg:  # explicit float[16] row (include/mpmhip.h), e.g. to share one row with a checker
            params = np.ascontiguousarray(cfg["params"], np.float32).reshape(16).copy()
        v0 = np.tile(np.asarray(_vec3(cfg.get("initial_velocity"), (0, 0, 0)), np.float32), (n, 1))
        if "velocities" in cfg:
            v0 = np.ascontiguousarray(cfg["velocities"], np.float32).reshape(-1, 3)[keep]
        dg = float(cfg.get("initial_dg", 1.0))  # src/particles.h:120
        F = np.tile((np.eye(3, dtype=np.float32) * dg).reshape(1, 9), (n, 1))
        if "F" in cfg:
            F = np.ascontiguousarray(cfg["F"], np.float32).reshape(-1, 9)[keep]
        B = np.zeros((n, 9), np.float32)
        if "B" in cfg:
            B = np.ascontiguousarray(cfg["B"], np.float32).reshape(-1, 9)[keep]
        aux = np.full(n, initial_aux(ptype, **cfg), np.float32)
        if "aux" in cfg:
            aux = np.ascontiguousarray(cfg["aux"], np.float32).reshape(-1)[keep]
        gi = len(self._groups)
        self._groups.append((mat, params))
        if self._ctx is not None:
            self._ensure_ctx(extra=n)
            self._check(self._L.mpmhip_add_group(self._ctx, mat, params.ctypes.data_as(C.POINTER(C.c_float))))
            self._upload_new(gi, x, v0, F, B, aux)
        else:
            self._staged.append((gi, (x, v0, F, B, aux)))
        self._n_added += n
        return ""

    # ---------------------------------------------------------------- CPIC rigid bodies
    def add_rigid_body(self, cfg):
        """MPM::add_rigid_particle (src/mpm_rigid_body.cpp:130-252) with the keys of the scene scripts
        (scripts/mls-cpic/*.py): codimensional (mandatory), density, friction | friction0 + friction1, restitution, scale,
        initial_position | scripted_position, initial_rotation | scripted_rotation (Euler angles, degrees),
        initial_velocity, initial_angular_velocity, rotation_axis, linear_damping, angular_damping, recenter,
        reverse_vertices.  The mesh: mesh=(n, 3, 3) triangles or mesh_fn='file.obj'.  Scripts are callables t -> 3-vector
        (tc.function13 / tc.constant_function13 of the reference's scenes).  Returns the body's index (>= 1)."""
        # check_scripting_parameters, src/mpm_rigid_body.cpp:15-56
        for bad, use in (("scripted", None), ("position", "initial_position"), ("rotation", "initial_rotation")):
            if bad in cfg:
                raise MPMError("'%s' is deprecated. Please remove." % bad if use is None else "Use '%s' instead of '%s'." % (use, bad))
        if "codimensional" not in cfg:
            raise MPMError("rigid bodies need the key 'codimensional'")
        if "scripted_position" in cfg and ("initial_position" in cfg or "initial_velocity" in cfg):
            raise MPMError("scripted_position and initial_position / initial_velocity cannot coexist.")
        if "scripted_position" not in cfg and "initial_position" not in cfg:
            raise MPMError("Please specify one (and only one) of 'scripted_position' and 'initial_position'.")
        if "scripted_rotation" in cfg and ("initial_rotation" in cfg or "initial_angular_velocity" in cfg):
            raise MPMError("scripted_rotation and initial_rotation / initial_angular_velocity cannot coexist!")
        if "friction" in cfg and ("friction0" in cfg or "friction1" in cfg):
            raise MPMError("friction and friction0 / friction1 cannot coexist!")
        if ("friction0" in cfg) != ("friction1" in cfg):
            raise MPMError("friction0 and friction1 must be specified simultaneuously.")
        if "mesh" in cfg:
            tri = np.ascontiguousarray(cfg["mesh"], np.float32).reshape(-1, 9)
        elif "mesh_fn" in cfg:
            tri = load_obj_triangles(cfg["mesh_fn"]).reshape(-1, 9)
        else:
            raise MPMError("a rigid body needs mesh=(n, 3, 3) triangles or mesh_fn='file.obj'")
        r = _lib.RigidConfig()
        r.codimensional = int(bool(cfg["codimensional"]))
        r.recenter = int(bool(cfg.get("recenter", True)))
        r.reverse_vertices = int(bool(cfg.get("reverse_vertices", False)))
        r.density = float(cfg.get("density", 0.0))
        f0, f1 = (cfg["friction"],) * 2 if "friction" in cfg else (cfg.get("friction0", 0.0), cfg.get("friction1", 0.0))
        r.friction[:] = (float(f0), float(f1))
        r.restitution = float(cfg.get("restitution", 0.0))
        r.scale[:] = _vec3(cfg.get("scale"), (1.0, 1.0, 1.0))
        r.initial_position[:] = _vec3(cfg.get("initial_position"), (0, 0, 0))
        r.initial_rotation[:] = _vec3(cfg.get("initial_rotation"), (0, 0, 0))
        r.initial_velocity[:] = _vec3(cfg.get("initial_velocity"), (0, 0, 0))
        r.initial_angular_velocity[:] = _vec3(cfg.get("initial_angular_velocity"), (0, 0, 0))
        r.rotation_axis[:] = _vec3(cfg.get("rotation_axis"), (0, 0, 0))
        r.linear_damping = float(cfg.get("linear_damping", 0.0))
        r.angular_damping = float(cfg.get("angular_damping", 0.0))

        def script(fn):
            def call(_user, t, out):
                # (an exception cannot cross the C frames between here and step(): ctypes would print and drop it and the
                # body would jump to a zeroed pose — it is stashed, the body keeps a finite pose, and the stepping call
                # that triggered the script re-raises it)
                try:
                    v = fn(float(t))
                    out[0], out[1], out[2] = float(v[0]), float(v[1]), float(v[2])
                    last[:] = [out[0], out[1], out[2]]
                except BaseException as e:  # noqa: BLE001
                    if self._script_error is None:
                        self._script_error = e
                    out[0], out[1], out[2] = last
            last = [0.0, 0.0, 0.0]
            return _lib.SCRIPT_FN(call)
        if cfg.get("scripted_position") is not None:
            r.scripted_position = script(cfg["scripted_position"])
        if cfg.get("scripted_rotation") is not None:
            r.scripted_rotation = script(cfg["scripted_rotation"])
        free = cfg.get("scripted_position") is None and float(cfg.get("density", 0.0)) > 0.0
        if free and self._free_bodies >= 1 or (free or self._free_bodies) and len(self._rigids) >= 1:
            # MPM::rigidify (src/mpm.cpp:468, libccd) is not part of this library: say so instead of letting bodies
            # pass through each other silently
            import warnings
            warnings.warn("taichi_mpm_amd: rigid-rigid collisions (MPM::rigidify, src/mpm.cpp:468) are not implemented — "
                          "bodies interact only through the material and through joints", RuntimeWarning, stacklevel=3)
        self._free_bodies += int(free)
        self._ensure_ctx()
        rid = self._check(self._L.mpmhip_add_rigid_body(self._ctx, C.byref(r), len(tri), tri.ctypes.data_as(C.POINTER(C.c_float))))
        self._check_script()
        self._rigids.append(r)  # keeps the callbacks alive for the life of the simulation
        return rid

    def get_rigid_state(self, rid):
        o = np.zeros(33, np.float32)
        self._check(self._L.mpmhip_rigid_get_state(self._ctx, int(rid), o.ctypes.data_as(C.POINTER(C.c_float))))
        return dict(position=o[0:3], rotation=o[3:7], velocity=o[7:10], angular_velocity=o[10:13], mass=float(o[13]),
                    inv_mass=float(o[14]), inertia=o[15:24].reshape(3, 3), inv_inertia=o[24:33].reshape(3, 3))

    def set_rigid_velocity(self, rid, velocity=None, angular_velocity=None):
        fp = C.POINTER(C.c_float)
        v = np.ascontiguousarray(velocity, np.float32) if velocity is not None else None
        w = np.ascontiguousarray(angular_velocity, np.float32) if angular_velocity is not None else None
        self._check(self._L.mpmhip_rigid_set_velocity(self._ctx, int(rid), v.ctypes.data_as(fp) if v is not None else None,
                                                      w.ctypes.data_as(fp) if w is not None else None))

    def get_rigid_samples(self, rid=-1):
        """the boundary particles of a body (all bodies: rid < 0): world position, body-frame offset, body index"""
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        n = self._check(self._L.mpmhip_rigid_get_samples(self._ctx, int(rid), 0, None, None, None))
        pos, off, body = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        if n:
            self._check(self._L.mpmhip_rigid_get_samples(self._ctx, int(rid), n, pos.ctypes.data_as(fp), off.ctypes.data_as(fp),
                                                         body.ctypes.data_as(ip)))
        return dict(pos=pos, offset=off, body=body)

    def download_cdf(self):
        """(states, distance) of every grid node, dense (res+1)^3: the colored distance field after rasterize_rigid_boundary"""
        shp = tuple(r + 1 for r in self.res)
        st, d = np.zeros(shp, np.uint32), np.zeros(shp, np.float32)
        self._check(self._L.mpmhip_download_cdf(self._ctx, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(C.POINTER(C.c_float))))
        return st, d

    def download_boundary(self):
        """gather_cdf's results per live particle in slot order: dict(normal, distance, near) — valid until the next G2P"""
        n = self.get_num_particles()
        o = np.zeros((n, 5), np.float32)
        m = self._check(self._L.mpmhip_download_boundary(self._ctx, o.ctypes.data_as(C.POINTER(C.c_float)), n))
        o = o[:m]
        return dict(normal=o[:, 0:3].copy(), distance=o[:, 3].copy(), near=o[:, 4].astype(np.int32))

    def get_num_particles(self):
        if self._ctx is None:
            return self._n_added
        return int(self._check(self._L.mpmhip_num_particles(self._ctx)))

    def get_particles(self, sort_by_id=True):
        """dict of numpy arrays (x, v, B, F, aux, gid, id) — the serialised particle fields of
        src/particles.h:52-66 that the hot path owns."""
        self._ensure_ctx()
        n = self.get_num_particles()
        out = {}
        for name, f in (("x", F_X), ("v", F_V), ("B", F_B), ("F", F_F), ("aux", F_AUX), ("gid", F_GID), ("id", F_ID), ("states", F_STATES)):
            dt = np.int32 if f in (F_GID, F_ID, F_STATES) else np.float32
            a = np.zeros((n, _WIDTH[f]), dt)
            got = self._check(self._L.mpmhip_download(self._ctx, f, a.ctypes.data_as(C.c_void_p), n))
            a = a[:got]
            out[name] = a[:, 0] if _WIDTH[f] == 1 else a
        if sort_by_id:
            order = np.argsort(out["id"], kind="stable")
            out = {k: v[order] for k, v in out.items()}
        return out

    # ---------------------------------------------------------------- level set
    def set_levelset(self, levelset):
        """Simulation::set_levelset: a LevelSet (static) or a DynamicLevelSet (two key frames)"""
        self._levelset = levelset
        if self._ctx is not None:
            self._apply_levelset()

    @staticmethod
    def _shape_array(ls):
        arr = (_lib.Shape * max(len(ls.shapes), 1))()
        for i, (t_, io, p) in enumerate(ls.shapes):
            arr[i].type, arr[i].inside_out = t_, io
            arr[i].p[:] = p
        return arr

    def _apply_levelset(self):
        ls = self._levelset
        if ls is None:
            return
        if isinstance(ls, DynamicLevelSet):
            l0, l1 = ls.levelset0, ls.levelset1
            # key frame times are relative to the ctx clock (a re-created ctx restarts it at 0)
            off = getattr(self, "_time_offset", 0.0)
            self._check(self._L.mpmhip_set_levelset_keyframes(self._ctx, ls.t0 - off, ls.t1 - off, len(l0.shapes), self._shape_array(l0),
                                                              len(l1.shapes), self._shape_array(l1), l0.friction))
            return
        self._check(self._L.mpmhip_set_levelset_shapes(self._ctx, len(ls.shapes), self._shape_array(ls), ls.friction))

    # ---------------------------------------------------------------- stepping
    def _check_script(self):
        if self._script_error is not None:
            e, self._script_error = self._script_error, None
            raise MPMError("a scripted_position / scripted_rotation callback raised %r" % (e,)) from e

    # --- the reference's own performance harness (src/mpm.cpp:516-523, 554-561): with the config key benchmark_rasterize /
    # benchmark_resample the substep never returns — `while (true) { Timer("Rasterize x 20"); base_delta_t = 0; 20 x
    # rasterize_optimized }`.  Here ONE bounded round: the kernel launched `rounds` times on the current state with dt = 0,
    # bracketed by device events, reported like the reference's timer plus the ns per particle its TC_PROFILE_TPE prints;
    # the state is restored afterwards and the run goes on (config keys: before the first substep).
    def _benchmark(self, what, rounds):
        self._ensure_ctx()
        n = self.get_num_particles()
        snap = np.empty(int(self._check(self._L.mpmhip_snapshot_size(self._ctx))), np.uint8)
        self._check(self._L.mpmhip_snapshot_save(self._ctx, snap.ctypes.data_as(C.c_void_p), len(snap)))
        # the benchmark borrows the profiler: the caller's level, sampling interval and accumulated timings come back afterwards
        level, every = getattr(self, "_prof_level", 0), getattr(self, "_prof_every", 1)
        before = self.profile(reset=True)
        self.set_profiling(3 if what == "rasterize" else 2)
        self._check(self._L.mpmhip_set_dt(self._ctx, C.c_float(0.0)))  # base_delta_t = 0: nothing moves
        self._check(self._L.mpmhip_run_substeps(self._ctx, int(rounds)))
        ms = self.profile(reset=True)["phases"]["p2g" if what == "rasterize" else "g2p"]
        self.set_profiling(level, every)
        self._prof_carry = {"substeps": before["substeps"], "phases": dict(before["phases"])} if before["substeps"] else None
        self._check(self._L.mpmhip_set_dt(self._ctx, C.c_float(self.base_delta_t)))
        self._check(self._L.mpmhip_snapshot_load(self._ctx, snap.ctypes.data_as(C.c_void_p), len(snap)))
        out = {"name": "%s x %d" % ("Rasterize" if what == "rasterize" else "Resample", rounds), "ms": ms, "particles": n,
               "ns_per_particle": 1e6 * ms / max(n * rounds, 1)}
        print("%s: %.3f ms (%.3f ns per particle per launch, %d particles)" % (out["name"], ms, out["ns_per_particle"], n))
        return out

    def benchmark_rasterize(self, rounds=20):
        return self._benchmark("rasterize", rounds)

    def benchmark_resample(self, rounds=20):
        return self._benchmark("resample", rounds)

    def _pending_benchmarks(self):
        if self._bench_keys:
            keys, self._bench_keys = self._bench_keys, ()
            for k in keys:
                self._benchmark(k, 20)

    def step(self, dt):
        """MPM<dim>::step, src/mpm.cpp:428-439: dt<0 => exactly one substep."""
        self._ensure_ctx()
        self._pending_benchmarks()
        rc = self._L.mpmhip_step(self._ctx, float(dt))
        self._check_script()
        self._check(rc)

    def substep(self):
        self._ensure_ctx()
        self._pending_benchmarks()
        rc = self._L.mpmhip_substep(self._ctx)
        self._check_script()
        self._check(rc)

    def run_substeps(self, n):
        self._ensure_ctx()
        rc = self._L.mpmhip_run_substeps(self._ctx, int(n))
        self._check_script()
        self._check(rc)

    def synchronize(self):
        self._ensure_ctx()
        self._check(self._L.mpmhip_synchronize(self._ctx))

    def set_stream(self, hip_stream):
        """run the ctx on an existing hipStream_t (int handle; 0/None = the ctx's own stream)"""
        self._ensure_ctx()
        self._stream = hip_stream or None
        self._check(self._L.mpmhip_set_stream(self._ctx, C.c_void_p(hip_stream or None)))

    def get_current_time(self):
        t0 = getattr(self, "_time_offset", 0.0)
        return t0 + (self._L.mpmhip_current_time(self._ctx) if self._ctx is not None else 0.0)

    # phase-level (names of the reference's member functions)
    def sort_particles_and_populate_grid(self):
        self._ensure_ctx(); self._check(self._L.mpmhip_sort(self._ctx))

    def rasterize_optimized(self):
        self._ensure_ctx(); self._check(self._L.mpmhip_p2g(self._ctx))

    def normalize_grid_and_apply_boundary_conditions(self):
        self._ensure_ctx(); self._check(self._L.mpmhip_grid_update(self._ctx))

    def resample_optimized(self):
        self._ensure_ctx(); self._check(self._L.mpmhip_g2p(self._ctx))

    def rasterize_rigid_boundary(self):  # src/rigid_transfer.cpp:17-115
        self._ensure_ctx(); self._check(self._L.mpmhip_rasterize_rigid_boundary(self._ctx))

    def gather_cdf(self):  # src/rigid_transfer.cpp:121-275
        self._ensure_ctx(); self._check(self._L.mpmhip_gather_cdf(self._ctx))

    def articulate(self):  # src/mpm.h:278-319
        self._ensure_ctx(); self._check(self._L.mpmhip_articulate(self._ctx))

    def advect_rigid_bodies(self):  # src/mpm_rigid_body.cpp:255-286
        self._ensure_ctx(); self._check(self._L.mpmhip_advect_rigid_bodies(self._ctx))

    def get_grid(self, which=1):
        self._ensure_ctx()
        g = np.zeros((self.res[0] + 1, self.res[1] + 1, self.res[2] + 1, 4), np.float32)
        self._check(self._L.mpmhip_download_grid(self._ctx, int(which), g.ctypes.data_as(C.POINTER(C.c_float))))
        return g

    def set_grid(self, grid):
        self._ensure_ctx()
        g = np.ascontiguousarray(grid, np.float32)
        assert g.shape == (self.res[0] + 1, self.res[1] + 1, self.res[2] + 1, 4)
        self._check(self._L.mpmhip_upload_grid(self._ctx, g.ctypes.data_as(C.POINTER(C.c_float))))

    def upload(self, field, array):
        self._ensure_ctx()
        a = np.ascontiguousarray(array, np.int32 if field in (F_GID, F_ID, F_STATES) else np.float32)
        self._check(self._L.mpmhip_upload(self._ctx, field, a.ctypes.data_as(C.c_void_p), len(a)))

    # ---------------------------------------------------------------- profiling (TC_PROFILE, src/mpm.cpp:464-572)
    def set_profiling(self, level=1, every=1):
        """0 off, 1 every phase, 2 only G2P, 3 only P2G, 4 the parts of a tiled substep (include/mpmhip.h); every: levels 2 / 3
        bracket the kernel in every `every`-th substep only"""
        self._ensure_ctx(); self._check(self._L.mpmhip_set_profiling(self._ctx, int(level)))
        self._check(self._L.mpmhip_set_profile_sampling(self._ctx, int(every)))
        self._prof_level, self._prof_every = int(level), int(every)

    def set_deterministic(self, on=True):
        """config key `deterministic` of a live simulation, from the next sort on (include/mpmhip.h: mpmhip_set_deterministic)"""
        self.deterministic = bool(on)
        if self._ctx is not None:
            self._check(self._L.mpmhip_set_deterministic(self._ctx, int(self.deterministic)))

    def cond_census(self):
        """cond(F) = sigma_max / sigma_min over the live particles (include/mpmhip.h: mpmhip_debug_cond_census): max, quantiles from
        the eighth-octave histogram (upper bin edges), the share of particles / waves the transfer kernels' eigen-solve refines"""
        self._ensure_ctx()
        out = (C.c_double * 264)()
        self._check(self._L.mpmhip_debug_cond_census(self._ctx, out))
        n, hist = out[0], np.asarray(out[8:264])
        cum = np.cumsum(hist)

        def q(p):
            # (upper edge of the eighth-octave bin the quantile falls in, never above the exact maximum)
            return min(float(2.0 ** ((int(np.searchsorted(cum, p * n)) + 1) / 8.0)), float(out[4])) if n else None
        return {"particles": int(n), "max": float(out[4]), "median": q(0.5), "p99": q(0.99), "p999": q(0.999), "p9999": q(0.9999),
                "frac_particles_refined": out[1] / n if n else 0.0, "frac_waves_with_refinement": out[3] / out[2] if out[2] else 0.0,
                "beyond_1e2": float(hist[int(8 * np.log2(100.0)):].sum() / n) if n else 0.0}

    def g2p_kernel(self):
        """name of the G2P kernel the next substep's plain blocks get (measurement helper: bench.py names its roofline after it)"""
        self._ensure_ctx()
        return "k_g2p_packed" if self._check(self._L.mpmhip_debug_g2p_is_packed(self._ctx)) else "k_g2p"

    def copy_bandwidth(self, nbytes=1 << 30, iters=5):
        """GB/s (read + written) of a plain streaming copy on this GPU: the measured yardstick next to the nominal peak"""
        self._ensure_ctx()
        out = C.c_double()
        self._check(self._L.mpmhip_debug_copy_bandwidth(self._ctx, int(nbytes), int(iters), C.byref(out)))
        return out.value

    def profile(self, reset=False):
        self._ensure_ctx()
        buf = C.create_string_buffer(1024)
        self._check(self._L.mpmhip_profile(self._ctx, buf, len(buf)))
        out = json.loads(buf.value.decode())
        carry = getattr(self, "_prof_carry", None)
        if carry:  # timings accumulated before a benchmark_rasterize / benchmark_resample run borrowed the profiler (_benchmark)
            out["substeps"] += carry["substeps"]
            for k, v in carry["phases"].items():
                out["phases"][k] = out["phases"].get(k, 0.0) + v
        if reset:
            self._check(self._L.mpmhip_profile_reset(self._ctx))
            self._prof_carry = None
        return out

    # ---------------------------------------------------------------- async limits (AsyncMPM, first half)
    def enable_async(self, unit_delta_t=1e-6, max_units=8192, cfl_dt_mul=1.0, strength_dt_mul=1.0):
        """config keys of AsyncMPM<dim>::initialize (src/async/async_mpm.cpp:24-27).  Builds the scheduler-block table; the
        stepping itself (AsyncMPM::advance) is not part of this build — see include/mpmhip.h."""
        self._ensure_ctx()
        a = _lib.AsyncConfig(float(unit_delta_t), int(max_units), float(cfl_dt_mul), float(strength_dt_mul), 0)
        self._check(self._L.mpmhip_async_enable(self._ctx, C.byref(a)))

    def update_dt_limits(self):
        """AsyncMPM<dim>::update_dt_limits (src/async/async_mpm.cpp:90-164): per-block strength / CFL / power-of-two limits"""
        self._check(self._L.mpmhip_async_update_dt_limits(self._ctx))

    def async_blocks(self):
        """(dict(coord, strength, cfl, continuous, count) of the non-empty scheduler blocks, (min_delta_t_int, max_delta_t_int))"""
        cap = 1 << 20
        coord = np.zeros((cap, 3), np.int32)
        arrs = [np.zeros(cap, np.int64) for _ in range(4)]
        mm = np.zeros(2, np.int64)
        lp, ip = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        n = self._check(self._L.mpmhip_async_blocks(self._ctx, cap, coord.ctypes.data_as(ip), *(a.ctypes.data_as(lp) for a in arrs),
                                                    mm.ctypes.data_as(lp)))
        return dict(coord=coord[:n], strength=arrs[0][:n], cfl=arrs[1][:n], continuous=arrs[2][:n], count=arrs[3][:n]), tuple(int(v) for v in mm)

    # ---------------------------------------------------------------- misc surface
    JOINT_TYPES = {"rotation": 0, "frozen": 1, "distance": 2, "axial_rotation": 3, "motor": 4, "stepper": 5}

    def add_articulation(self, cfg):
        """general_action(action='add_articulation', ...) (src/mpm.cpp:923-933): a joint between two rigid bodies, with the
        keys of src/articulation.cpp — type ('rotation', 'frozen', 'distance', 'axial_rotation', 'motor', 'stepper'), obj0,
        obj1 (body indices as add_particles(type='rigid') returns them; obj1 absent = the background body), offset0,
        offset1, target_distance, penalty, axis, axis_length, power, angular_velocity."""
        name = cfg.get("type")
        if name not in self.JOINT_TYPES:
            raise MPMError("unknown articulation type %r (registered: %s)" % (name, ", ".join(sorted(self.JOINT_TYPES))))
        if "obj0" not in cfg:
            raise MPMError("add_articulation needs 'obj0'")
        j = _lib.JointConfig()
        j.type = self.JOINT_TYPES[name]
        j.obj0, j.obj1 = int(cfg["obj0"]), int(cfg.get("obj1", 0))
        j.offset0[:] = _vec3(cfg.get("offset0"), (0, 0, 0))
        j.has_offset1 = int("offset1" in cfg)
        j.offset1[:] = _vec3(cfg.get("offset1"), (0, 0, 0))
        j.has_target_distance = int("target_distance" in cfg)
        j.target_distance = float(cfg.get("target_distance", 0.0))
        j.penalty = float(cfg.get("penalty", -1.0))
        j.axis[:] = _vec3(cfg.get("axis"), (0, 0, 0))
        j.axis_length = float(cfg.get("axis_length", -1.0))
        j.power = float(cfg.get("power", 0.0))
        j.angular_velocity = float(cfg.get("angular_velocity", 0.0))
        self._ensure_ctx()
        self._check(self._L.mpmhip_add_articulation(self._ctx, C.byref(j)))
        return ""

    def general_action(self, config):
        """MPM<dim>::general_action, src/mpm.cpp:920-978."""
        action = config.get("action")
        if action == "add_articulation":  # src/mpm.cpp:923-933
            return self.add_articulation(config)
        if action == "calculate_energy":  # src/mpm.cpp:936-938 -> :1078-1110
            k, p = self.calculate_energy()
            return str(k + p)
        if action == "save":  # src/mpm.cpp:940-949: whole-state snapshot
            self.save_snapshot(config["file_name"])
            return ""
        if action == "load":  # src/mpm.cpp:950-960
            self.load_snapshot(config["file_name"])
            return ""
        if action == "delete_particles_inside_level_set":  # src/mpm.cpp:962-974
            self._ensure_ctx()
            n = C.c_int64()
            self._check(self._L.mpmhip_delete_particles_inside_level_set(self._ctx, C.byref(n)))
            return ""
        if action == "export":  # particle fields as .npz (not a reference action; handy for post-processing)
            p = self.get_particles()
            np.savez(config["file_name"], t=self.get_current_time(), frame=self.frame, **p)
            return ""
        raise MPMError("general_action(action=%r) is outside the scope of this build" % (action,))

    def save_snapshot(self, path):
        """raw records + group table + clocks (include/mpmhip.h: mpmhip_snapshot_save)"""
        self._ensure_ctx()
        n = int(self._check(self._L.mpmhip_snapshot_size(self._ctx)))
        buf = np.empty(n, np.uint8)
        self._check(self._L.mpmhip_snapshot_save(self._ctx, buf.ctypes.data_as(C.c_void_p), n))
        with open(path, "wb") as f:
            f.write(np.array([self.frame], np.int64).tobytes())
            f.write(buf.tobytes())

    def load_snapshot(self, path):
        """into a simulation initialised with the same grid; level set and config come from the script, as in the
        reference.  Replaces all particles and groups."""
        raw = np.fromfile(path, np.uint8)
        self.frame = int(raw[:8].view(np.int64)[0])
        blob = np.ascontiguousarray(raw[8:])
        n_groups = int(blob[12:16].view(np.uint32)[0])
        n_slots = int(blob[16:24].view(np.int64)[0])
        if self._ctx is None:
            self._staged, self._groups = [], []
            self._create(max(self.max_particles, int(n_slots * 1.25) + 1024))
        elif n_slots > self._capacity:  # grown in place: bodies added by the scene before the load stay (mpmhip_reserve)
            cap = max(self.max_particles, int(n_slots * 1.25) + 1024)
            self._check(self._L.mpmhip_reserve(self._ctx, cap))
            self._capacity = cap
        self._check(self._L.mpmhip_snapshot_load(self._ctx, blob.ctypes.data_as(C.c_void_p), len(blob)))
        rows = blob[80:80 + 80 * n_groups].view(np.float32).reshape(n_groups, 20)
        self._groups = [(int(r[16:17].view(np.int32)[0]), r[:16].copy()) for r in rows]
        self._n_added = n_slots
        self._time_offset = 0.0

    def calculate_energy(self):
        """(kinetic, potential): grid kinetic energy after a fresh P2G + sum of the particles' potential energies
        (src/mpm.cpp:1078-1110).  Types without potential_energy() raise, as the reference aborts."""
        self._ensure_ctx()
        k, p = C.c_double(), C.c_double()
        self._check(self._L.mpmhip_calculate_energy(self._ctx, C.byref(k), C.byref(p)))
        return k.value, p.value

    def test(self):  # src/mpm.cpp:577-580
        return True

    def get_debug_information(self):  # src/mpm.cpp:635-639
        return ""

    def visualize(self):
        """MPM<3>::visualize -> write_bgeo (src/visualize.cpp:156-159, src/mpm.h:333-337): the next
        `frame_directory/%04d.bgeo`, frame numbers starting at 1."""
        if not self.frame_directory:
            raise MPMError("visualize() needs the config key 'frame_directory'")
        self.frame_count += 1
        os.makedirs(self.frame_directory, exist_ok=True)
        path = os.path.join(self.frame_directory, "%04d.bgeo" % self.frame_count)
        self.write_partio(path)
        # "Start from 1. (0 is the background rigid body.)": every body's mesh in world space next to the frame (src/mpm.h:338-343)
        for rid in range(1, self.get_num_rigid_bodies()):
            self.write_rigid_body(rid, os.path.join(self.frame_directory, "rigid_%03d_%04d" % (rid, self.frame_count)))
        return path

    def get_num_rigid_bodies(self):
        """rigids.size(): the background body included"""
        return int(self._L.mpmhip_num_rigid_bodies(self._ctx)) if self._ctx is not None else 1

    def get_rigid_mesh(self, rid):
        """the body's triangles in world space, (n, 3, 3)"""
        fp = C.POINTER(C.c_float)
        n = self._check(self._L.mpmhip_rigid_get_mesh(self._ctx, int(rid), 0, None))
        tri = np.zeros((n, 3, 3), np.float32)
        if n:
            self._check(self._L.mpmhip_rigid_get_mesh(self._ctx, int(rid), n, tri.ctypes.data_as(fp)))
        return tri

    def write_rigid_body(self, rid, file_name):
        """MPM<3>::write_rigid_body (src/visualize.cpp:131-153): `file_name`.obj with three `v` records per triangle and one `f`
        record per triangle (the number format of the reference's fmt build is not pinned: %.9g here)"""
        tri = self.get_rigid_mesh(rid)
        with open(file_name + ".obj", "w") as f:
            for t in tri:
                for v in t:
                    f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
            for k in range(len(tri)):
                f.write("f %d %d %d\n" % (3 * k + 1, 3 * k + 2, 3 * k + 3))
        return file_name + ".obj"

    def write_partio(self, file_name):
        """MPM<dim>::write_partio (src/visualize.cpp:17-100): Houdini .bgeo v5 with the reference's attributes, rows
        assembled on the device (include/mpmhip.h: mpmhip_write_bgeo)."""
        self._ensure_ctx()
        self._check(self._L.mpmhip_write_bgeo(self._ctx, os.fsencode(file_name), int(self.verbose_bgeo)))

    def bgeo_bytes(self, verbose=None):
        """the .bgeo image of the current state as bytes (no file)"""
        self._ensure_ctx()
        verbose = int(self.verbose_bgeo if verbose is None else verbose)
        n = C.c_size_t()
        self._check(self._L.mpmhip_bgeo_size(self._ctx, verbose, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        w = C.c_size_t()
        self._check(self._L.mpmhip_bgeo_encode(self._ctx, verbose, buf.ctypes.data_as(C.c_void_p), n.value, C.byref(w)))
        return buf[:w.value].tobytes()

    def get_mpi_world_rank(self):  # scripts/async/async_mpm.py:198-199
        return 0

    def get_vis_resolution(self):
        """what the reference's driver sizes its video frames with (scripts/async/async_mpm.py:79-81: `.x`, `.y`).  The
        method lives in the absent taichi core (Simulation<dim>); here: the grid resolution in x and y."""
        import types
        return types.SimpleNamespace(x=int(self.res[0]), y=int(self.res[1]))

    def get_name(self):  # src/mpm.h:486-488
        return "mpm"


def create_simulation2(name):
    """tc_core.create_simulation2 (scripts/async/async_mpm.py:25-32): MPM<2>
    (TC_IMPLEMENTATION(Simulation2D, MPM2D, "mpm"), src/mpm.cpp:983-986)"""
    if name == "async_mpm":  # TC_IMPLEMENTATION(Simulation2D, AsyncMPM2D, "async_mpm"), src/async/async_mpm.cpp:423-427
        from .async_mpm import AsyncSimulation2D
        return AsyncSimulation2D()
    if name != "mpm":
        raise MPMError("no Simulation2D implementation named %r (registered: 'mpm', 'async_mpm')" % (name,))
    from .mpm2d import Simulation2D
    return Simulation2D()


def create_simulation3(name):
    """tc_core.create_simulation3 (scripts/async/async_mpm.py:25-32); only 'mpm' is registered here
    (TC_IMPLEMENTATION(Simulation3D, MPM3D, "mpm"), src/mpm.cpp:986-988)."""
    if name == "async_mpm":  # TC_IMPLEMENTATION(Simulation3D, AsyncMPM3D, "async_mpm"), src/async/async_mpm.cpp:423-427
        from .async_mpm import AsyncSimulation3D
        return AsyncSimulation3D()
    if name != "mpm":
        raise MPMError("no Simulation3D implementation named %r (registered: 'mpm', 'async_mpm')" % (name,))
    return Simulation3D()


def lattice_cube(lower, higher, dx):
    """8 particles per cell at cell centre +- 0.25 dx (src/mpm.cpp:164-180)."""
    r = np.arange(lower, higher, dtype=np.float64)
    ii, jj, kk = np.meshgrid(r, r, r, indexing="ij")
    cells = np.stack([ii, jj, kk], -1).reshape(-1, 1, 3) + 0.5
    signs = np.array([[1 if (i % 2) else -1, 1 if (i // 2 % 2) else -1, 1 if (i // 4 % 2) else -1] for i in range(8)],
                     np.float64)
    x = (cells + 0.25 * signs[None]) * dx
    return x.reshape(-1, 3).astype(np.float32)


class MPM:
    """Scene-script driver: `tc.dynamics.MPM(**kwargs)` shape (scripts/benchmark/benchmark_3d.py:9-27;
    in-tree twin: AsyncMPM, scripts/async/async_mpm.py:17-300)."""
    simulation_name = "mpm"

    def __init__(self, snapshot_interval=20, **kwargs):
        res = kwargs["res"]
        # the reference's driver derives its output directory from task_id (tc.get_output_path, scripts/async/async_mpm.py:36-49)
        # and injects frame_directory = <directory>/frames; here: 'output_directory' names it (default: none — nothing is written
        # unless the script asks for it), an explicit 'frame_directory' is kept as given
        self.snapshot_interval = snapshot_interval
        self.task_id = kwargs.get("task_id", os.path.splitext(os.path.basename(sys.argv[0] or "mpm"))[0])
        self.directory = kwargs.pop("output_directory", None)
        # (snapshots every snapshot_interval frames only with an explicit output directory: a script that names just its
        # frame_directory gets frames and nothing else)
        self.snapshot_directory = os.path.join(self.directory, "snapshots") if self.directory is not None else None
        if self.directory is not None:
            kwargs.setdefault("frame_directory", os.path.join(self.directory, "frames"))
        elif kwargs.get("frame_directory"):
            self.directory = os.path.dirname(os.path.normpath(kwargs["frame_directory"]))
        self.frame_dt = kwargs.get("frame_dt", 0.01)
        kwargs.setdefault("frame_dt", self.frame_dt)
        self.num_frames = kwargs.get("num_frames", 1000)
        if len(res) not in (2, 3):
            raise MPMError("res must have 2 or 3 entries")
        if len(res) == 2 and self.simulation_name != "mpm":
            raise MPMError("create_simulation2(%r) is not part of this build" % self.simulation_name)
        self.c = create_simulation3(self.simulation_name) if len(res) == 3 else create_simulation2("mpm")  # async_mpm.py:25-32
        if "delta_x" not in kwargs:
            kwargs["delta_x"] = 1.0 / res[0]  # async_mpm.py:40-41
        self.c.initialize(kwargs)
        self.res = tuple(res)
        self.c.frame = 0
        self.levelset_generator = None
        self.simulation_total_time = 0.0

    def add_particles(self, **kwargs):
        return self.c.add_particles(kwargs)

    def create_levelset(self):
        return LevelSet(delta_x=1.0 / self.res[0])

    def update_levelset(self, t0, t1):  # scripts/async/async_mpm.py:119-127
        if self.levelset_generator is None:
            return
        self.c.set_levelset(DynamicLevelSet().initialize(t0, t1, self.levelset_generator(t0), self.levelset_generator(t1)))

    def set_levelset(self, levelset, is_dynamic_levelset=False):  # scripts/async/async_mpm.py:129-137
        if is_dynamic_levelset:
            self.levelset_generator = levelset  # a function t -> LevelSet, sampled at both ends of every frame
        else:
            self.levelset_generator = None
            self.c.set_levelset(levelset)

    def get_current_time(self):
        return self.c.get_current_time()

    def step(self, step_t):
        import time
        t = self.c.get_current_time()
        self.update_levelset(t, t + step_t)
        T = time.time()
        self.c.step(step_t)
        self.c.synchronize()
        self.simulation_total_time += time.time() - T  # (the frame counter is the frame loop's, scripts/async/async_mpm.py:246)

    def general_action(self, **kwargs):
        return self.c.general_action(kwargs)

    def add_articulation(self, **kwargs):  # scripts/async/async_mpm.py:277-279
        kwargs["action"] = "add_articulation"
        return self.c.general_action(kwargs)

    def visualize(self):  # scripts/async/async_mpm.py:201-202
        return self.c.visualize()

    # ---- the rest of the driver's surface (scripts/async/async_mpm.py:185-300)
    def get_directory(self):
        return self.directory

    def make_video(self):
        raise MPMError("make_video(): rendering is outside the scope of this build (the frames are .bgeo files)")

    def test(self):
        return self.c.test()

    def get_mpi_world_rank(self):
        return self.c.get_mpi_world_rank()

    def get_debug_information(self):
        return self.c.get_debug_information()

    def clear_output_directory(self):
        """the frames of an earlier run (.bgeo, the bodies' .obj / .poly); snapshots stay, as in the reference (:211-215)"""
        frames = getattr(self.c, "frame_directory", None)
        if frames and os.path.isdir(frames):
            for f in os.listdir(frames):
                if f.endswith((".bgeo", ".obj", ".poly")):
                    os.remove(os.path.join(frames, f))

    def delete_particles_inside_level_set(self):  # :281-284 (a dynamic level set is sampled at the current time first)
        t = self.c.get_current_time()
        self.update_levelset(t, t + 1)
        self.c.general_action(dict(action="delete_particles_inside_level_set"))

    def action(self, **kwargs):
        self.c.general_action(kwargs)

    def save(self, fn):
        self.action(action="save", file_name=fn)

    def load(self, fn):
        """(scripted motions are Python callables of THIS process, re-attached by the script that adds the bodies: nothing to
        pass along, unlike the reference's function addresses, :292-297)"""
        self.action(action="load", file_name=fn)

    def get_snapshot_file_name(self, iteration):
        if self.snapshot_directory is None:
            raise MPMError("snapshots need an output directory: MPM(output_directory=...)")
        return os.path.join(self.snapshot_directory, "%04d.tcb" % iteration)

    def _frames(self, num_frames, frame_update, update_frequency, print_profile_info, per_frame):
        # the loop and the snapshot names follow the simulation object's frame counter (scripts/async/async_mpm.py:236-248:
        # `while self.c.frame < self.num_frames`), which snapshots carry: after load() the run RESUMES at the loaded frame
        n = self.num_frames if num_frames is None else num_frames
        while self.c.frame < n:
            for _ in range(update_frequency):
                if frame_update:
                    frame_update(self.get_current_time(), self.frame_dt / update_frequency)
                self.step(self.frame_dt / update_frequency)
            if getattr(self.c, "frame_directory", None):  # one .bgeo per frame, as async_mpm.py:243
                self.visualize()
            if print_profile_info and hasattr(self.c, "profile"):
                print(json.dumps(self.c.profile(reset=True)))
            self.c.frame += 1
            per_frame(self.c.frame)

    def simulate(self, num_frames=None, print_profile_info=False, frame_update=None, clear_output_directory=False,
                 update_frequency=1, **_ignored):
        """python frame loop (scripts/async/async_mpm.py:217-248): per frame `update_frequency` x step(frame_dt /
        update_frequency), a frame file, [profile print], a snapshot every `snapshot_interval` frames when the driver has
        an output directory."""
        if print_profile_info and hasattr(self.c, "set_profiling"):  # (the 2D simulation has no phase profile)
            self.c.set_profiling(True)
        if clear_output_directory:
            self.clear_output_directory()

        def after(done):
            if self.snapshot_directory is not None and self.snapshot_interval and done % self.snapshot_interval == 0:
                os.makedirs(self.snapshot_directory, exist_ok=True)
                self.save(self.get_snapshot_file_name(done))
        self._frames(num_frames, frame_update, update_frequency, print_profile_info, after)

    def simulate_with_energy(self, num_frames=None, print_profile_info=False, frame_update=None, clear_output_directory=False,
                             update_frequency=1):
        """the same loop, returning the total energy after every frame (:250-272)"""
        if clear_output_directory:
            self.clear_output_directory()
        energy = []
        self._frames(num_frames, frame_update, update_frequency, print_profile_info,
                     lambda done: energy.append(float(self.general_action(action="calculate_energy"))))
        return energy


class AsyncMPM(MPM):
    """scripts/async/async_mpm.py:17-300 `AsyncMPM(**kwargs)`: the same driver over create_simulation3('async_mpm')
    (block-local time steps; config keys unit_delta_t, max_units, cfl_dt_mul, strength_dt_mul)"""
    simulation_name = "async_mpm"
