"""MPM<2> — the reference's 2D simulation (`tc_core.create_simulation2('mpm')`, src/mpm.cpp:983-986) on top of the
mpmhip2d_* entry points of the C ABI (include/mpmhip.h).  Same surface as Simulation3D (taichi_mpm_amd/mpm.py) where it
applies: `initialize`, `add_particles`, `set_levelset`, `step`, `get_current_time`, `get_particles`.  MPM<2> runs the
generic transfer path (src/transfer.cpp:280-283,697-700)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .materials import MATERIAL_IDS, group_params, initial_aux
from .mpm import DynamicLevelSet, LevelSet, MPMError, check_unsupported_keys


def lattice_square(lower, higher, dx):
    """4 particles per cell at cell centre +- 0.25 dx (the 2D analogue of the 3D benchmark lattice)"""
    r = np.arange(lower, higher, dtype=np.float64)
    ii, jj = np.meshgrid(r, r, indexing="ij")
    cells = np.stack([ii, jj], -1).reshape(-1, 1, 2) + 0.5
    signs = np.array([[-1, -1], [1, -1], [-1, 1], [1, 1]], np.float64)
    return ((cells + 0.25 * signs[None]) * dx).reshape(-1, 2).astype(np.float32)


class Simulation2D:
    def __init__(self):
        self._L = _lib.load()
        self._ctx = None
        self._staged = []
        self._groups = []
        self._levelset = None
        self._n_added = 0
        self.frame = 0
        self._rigids = []  # ctypes keep-alives of the rigid bodies' configs / script callbacks

    def initialize(self, config):
        cfg = dict(config)
        if "delta_t" in cfg:  # src/mpm.cpp:41-42
            raise MPMError("Please use 'base_delta_t' instead of 'delta_t'")
        check_unsupported_keys(cfg)
        for k in ("benchmark_rasterize", "benchmark_resample"):  # (built for the 3D simulation only)
            if cfg.get(k, False):
                raise MPMError("config key %r (src/mpm.cpp:516-538, 554-561) is not implemented by the 2D simulation" % k)
        res = cfg["res"]
        res = (int(res),) * 2 if np.isscalar(res) else tuple(int(r) for r in res)
        if len(res) != 2:
            raise MPMError("Simulation2D needs a 2-entry res")
        self.res = res
        self.delta_x = float(cfg.get("delta_x", 1.0 / res[0]))
        self.base_delta_t = float(cfg.get("base_delta_t", 1e-4)) * float(cfg.get("dt_multiplier", 1.0))
        g = cfg.get("gravity", (0.0, -10.0))
        self.gravity = (float(g[0]), float(g[1]))
        self.config = cfg
        self.max_particles = int(cfg.get("max_particles", 0))
        self.verbose_bgeo = bool(cfg.get("verbose_bgeo", False))  # src/visualize.cpp:22
        self.frame_directory = cfg.get("frame_directory")  # injected by the python driver, async_mpm.py:49
        self.frame_count = 0
        return self

    def _check(self, rc):
        if rc < 0:
            raise MPMError("libmpmhip error %d: %s" % (rc, self._L.mpmhip2d_last_error(self._ctx).decode()))
        return rc

    def _ensure_ctx(self, extra=0):
        if self._ctx is not None:
            if self._resident_particles() + extra > self._capacity:
                raise MPMError("2D particle capacity exceeded: pass max_particles to initialize()")
            return
        cfg = self.config
        c = _lib.Config2D()
        c.res[:] = self.res
        c.dx, c.dt = self.delta_x, self.base_delta_t
        c.gravity[:] = self.gravity
        c.particle_gravity = int(bool(cfg.get("particle_gravity", True)))
        c.apic_damping, c.rpic_damping = float(cfg.get("apic_damping", 0.0)), float(cfg.get("rpic_damping", 0.0))
        c.clean_boundary = int(bool(cfg.get("clean_boundary", True)))
        c.particle_collision = int(bool(cfg.get("particle_collision", False)))
        self._capacity = max(self.max_particles, int((self._n_added + extra) * 1.5) + 1024)
        c.max_particles = self._capacity
        c.device = int(cfg.get("device", 0))
        ctx = C.c_void_p()
        rc = self._L.mpmhip2d_create(C.byref(c), C.byref(ctx))
        if rc != 0:
            raise MPMError("mpmhip2d_create failed (%d): %s" % (rc, self._L.mpmhip2d_last_error(None).decode()))
        self._ctx = ctx
        self._check(self._L.mpmhip2d_set_rigid_coupling(ctx, float(cfg.get("penalty", 0.0)), float(cfg.get("pushing_force", 20000.0))))
        self._check(self._L.mpmhip2d_set_articulation_iterations(ctx, int(cfg.get("articulation_iterations", 100))))  # src/mpm.h:279-280
        self._check(self._L.mpmhip2d_set_rigid_levelset_collision(ctx, int(bool(cfg.get("rigid_body_levelset_collision", False)))))  # src/mpm.cpp:535-538
        d = float(cfg.get("dirichlet_boundary_radius", 0.0))  # src/mpm.cpp:541-544 -> apply_dirichlet_boundary_conditions, :374-399
        vel = float(cfg.get("dirichlet_boundary_velocity", 0.0))
        self._check(self._L.mpmhip2d_set_dirichlet(ctx, int(d > 0.0), float(cfg.get("dirichlet_distance_left", d)), float(cfg.get("dirichlet_distance_right", d)),
                                                   float(cfg.get("dirichlet_boundary_left", vel)), float(cfg.get("dirichlet_boundary_right", vel))))
        self._apply_levelset()
        for gi, (mat, params, arrs) in enumerate(self._staged):
            self._add(mat, params, *arrs)
        self._staged = []

    def _resident_particles(self):
        """particles held by the object's arrays (what max_particles bounds)"""
        return self._n_added

    def close(self):
        if self._ctx is not None:
            self._L.mpmhip2d_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _add(self, mat, params, x, v, F, B, aux):
        fp = C.POINTER(C.c_float)
        gi = self._check(self._L.mpmhip2d_add_group(self._ctx, mat, params.ctypes.data_as(fp)))
        keep = []

        def ptr(a, w):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32).reshape(len(x), w) if w > 1 else np.ascontiguousarray(a, np.float32).reshape(len(x))
            keep.append(a)
            return a.ctypes.data_as(fp)
        self._check(self._L.mpmhip2d_add_particles(self._ctx, gi, len(x), ptr(x, 2), ptr(v, 2), ptr(F, 4), ptr(B, 4), ptr(aux, 1)))

    def add_particles(self, config):
        """MPM<2>::add_particles (src/mpm.cpp:77-270) with explicit `positions=` (n, 2) or a `square=(lo, hi)` lattice"""
        cfg = dict(config)
        ptype = cfg.get("type")
        if ptype == "rigid":  # src/mpm.cpp:80-83
            return str(self.add_rigid_body(cfg))
        if ptype not in MATERIAL_IDS:
            raise MPMError("unknown particle type %r" % (ptype,))
        dx = self.delta_x
        maximum = float(cfg.get("ppc", cfg.get("maximum", 4)))
        if "square" in cfg:
            x = lattice_square(int(cfg["square"][0]), int(cfg["square"][1]), dx)
        elif "positions" in cfg:
            x = np.ascontiguousarray(cfg["positions"], np.float32).reshape(-1, 2)
        else:
            raise MPMError("add_particles needs positions= or square=(lo, hi)")
        X = x.astype(np.float64) / dx
        keep = ~((X.min(1) < 7.0) | ((X - np.asarray(self.res)).max(1) > -7.0))  # src/mpm.cpp:129-132
        x = x[keep]
        n = len(x)
        vol = dx ** 2 / maximum  # pow<dim>(delta_x) / maximum, src/mpm.cpp:134
        mass = vol * float(cfg.get("density", 400.0))
        params, mat = group_params(ptype, mass, vol, **{k: v for k, v in cfg.items() if isinstance(v, (int, float))})
        if "params" in cfg:
            params = np.ascontiguousarray(cfg["params"], np.float32).reshape(16).copy()

        def sel(key, w, default):
            if key in cfg:
                a = np.ascontiguousarray(cfg[key], np.float32)
                return (a.reshape(-1, w) if w > 1 else a.reshape(-1))[keep]
            return default
        v = sel("velocities", 2, None)
        F = sel("F", 4, None)
        B = sel("B", 4, None)
        aux = sel("aux", 1, np.full(n, initial_aux(ptype, **cfg), np.float32))
        if self._ctx is None:
            self._staged.append((mat, params, (x, v, F, B, aux)))
        else:
            self._ensure_ctx(extra=n)
            self._add(mat, params, x, v, F, B, aux)
        self._groups.append((mat, params))
        self._n_added += n
        return ""

    def set_levelset(self, levelset):
        self._levelset = levelset
        if self._ctx is not None:
            self._apply_levelset()

    @staticmethod
    def _shapes(ls):
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
            self._check(self._L.mpmhip2d_set_levelset(self._ctx, len(l0.shapes), self._shapes(l0), len(l1.shapes), self._shapes(l1),
                                                      ls.t0, ls.t1, l0.friction))
        else:
            self._check(self._L.mpmhip2d_set_levelset(self._ctx, len(ls.shapes), self._shapes(ls), -1, None, 0.0, 1.0, ls.friction))

    def step(self, dt):
        self._ensure_ctx()
        self._check(self._L.mpmhip2d_step(self._ctx, float(dt)))

    def substep(self):
        self._ensure_ctx()
        self._check(self._L.mpmhip2d_substep(self._ctx))

    def run_substeps(self, n):
        for _ in range(int(n)):
            self.substep()

    def synchronize(self):
        self.get_num_particles()

    def get_current_time(self):
        return self._L.mpmhip2d_current_time(self._ctx) if self._ctx is not None else 0.0

    # ---------------------------------------------------------------- CPIC rigid bodies (segments)
    def add_rigid_body(self, cfg):
        """MPM<2>::add_rigid_particle (src/mpm_rigid_body.cpp:130-252, dim = 2): mesh=(n, 2, 2) segments; keys as in 3D
        (codimensional is mandatory; initial_rotation / scripted_rotation are one angle in degrees)"""
        if "codimensional" not in cfg:
            raise MPMError("rigid bodies need the key 'codimensional'")
        if "scripted_position" not in cfg and "initial_position" not in cfg:
            raise MPMError("Please specify one (and only one) of 'scripted_position' and 'initial_position'.")
        seg = np.ascontiguousarray(cfg["mesh"], np.float32).reshape(-1, 4)
        r = _lib.RigidConfig2D()
        r.codimensional = int(bool(cfg["codimensional"]))
        r.recenter = int(bool(cfg.get("recenter", True)))
        r.reverse_vertices = int(bool(cfg.get("reverse_vertices", False)))
        r.density = float(cfg.get("density", 0.0))
        f0, f1 = (cfg["friction"],) * 2 if "friction" in cfg else (cfg.get("friction0", 0.0), cfg.get("friction1", 0.0))
        r.friction[:] = (float(f0), float(f1))
        r.restitution = float(cfg.get("restitution", 0.0))
        sc = cfg.get("scale", (1.0, 1.0))
        r.scale[:] = (float(sc[0]), float(sc[1]))
        p0 = cfg.get("initial_position", (0.0, 0.0))
        r.initial_position[:] = (float(p0[0]), float(p0[1]))
        r.initial_rotation = float(cfg.get("initial_rotation", 0.0))
        v0 = cfg.get("initial_velocity", (0.0, 0.0))
        r.initial_velocity[:] = (float(v0[0]), float(v0[1]))
        r.initial_angular_velocity = float(cfg.get("initial_angular_velocity", 0.0))
        r.linear_damping = float(cfg.get("linear_damping", 0.0))
        r.angular_damping = float(cfg.get("angular_damping", 0.0))
        if cfg.get("scripted_position") is not None:
            fn = cfg["scripted_position"]

            def pos(_u, t, out, fn=fn):
                v = fn(float(t))
                out[0], out[1] = float(v[0]), float(v[1])
            r.scripted_position = _lib.SCRIPT_FN(pos)
        if cfg.get("scripted_rotation") is not None:
            gn = cfg["scripted_rotation"]

            def rot(_u, t, out, gn=gn):
                out[0] = float(gn(float(t)))
            r.scripted_rotation = _lib.SCRIPT_FN(rot)
        self._ensure_ctx()
        rid = self._check(self._L.mpmhip2d_add_rigid_body(self._ctx, C.byref(r), len(seg), seg.ctypes.data_as(C.POINTER(C.c_float))))
        self._rigids.append(r)
        return rid

    def get_rigid_state(self, rid):
        """position 2, angle (radians), velocity 2, angular velocity, mass, inv_mass, inertia, inv_inertia"""
        o = np.zeros(10, np.float32)
        self._check(self._L.mpmhip2d_rigid_get_state(self._ctx, int(rid), o.ctypes.data_as(C.POINTER(C.c_float))))
        return o

    def get_rigid_samples(self, rid=-1):
        n = self._check(self._L.mpmhip2d_rigid_get_samples(self._ctx, int(rid), 0, None))
        pos = np.zeros((n, 2), np.float32)
        if n:
            self._check(self._L.mpmhip2d_rigid_get_samples(self._ctx, int(rid), n, pos.ctypes.data_as(C.POINTER(C.c_float))))
        return pos

    def get_rigid_mesh(self, rid):
        """the body's segments in world space, (n, 2, 2)"""
        fp = C.POINTER(C.c_float)
        n = self._check(self._L.mpmhip2d_rigid_get_mesh(self._ctx, int(rid), 0, None))
        seg = np.zeros((n, 2, 2), np.float32)
        if n:
            self._check(self._L.mpmhip2d_rigid_get_mesh(self._ctx, int(rid), n, seg.ctypes.data_as(fp)))
        return seg

    def write_rigid_body(self, rid, file_name):
        """MPM<2>::write_rigid_body (src/visualize.cpp:105-130): `file_name`.poly — POINTS, POLYS (one per segment), END"""
        seg = self.get_rigid_mesh(rid).reshape(-1, 2)
        with open(file_name + ".poly", "w") as f:
            f.write("POINTS\n")
            for i, v in enumerate(seg):
                f.write("%d: %.9g %.9g 0.0\n" % (i + 1, v[0], v[1]))
            f.write("POLYS\n")
            for i in range(1, len(seg) // 2 + 1):
                f.write("%d: %d %d\n" % (i, 2 * i - 1, 2 * i))
            f.write("END\n")
        return file_name + ".poly"

    def cdf_phase(self):
        """rasterize_rigid_boundary + gather_cdf as one phase (parity tests; substep() runs them itself)"""
        self._ensure_ctx(); self._check(self._L.mpmhip2d_cdf_phase(self._ctx))

    def download_cdf(self):
        shp = tuple(r + 1 for r in self.res)
        st, d = np.zeros(shp, np.uint32), np.zeros(shp, np.float32)
        self._check(self._L.mpmhip2d_download_cdf(self._ctx, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(C.POINTER(C.c_float))))
        return st, d

    def download_colours(self):
        """per live particle in slot order: states, boundary distance, normal, near flag"""
        n = self.get_num_particles()
        st, d, nr, near = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.int32)
        fp = C.POINTER(C.c_float)
        got = self._check(self._L.mpmhip2d_download_colours(self._ctx, n, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(fp),
                                                            nr.ctypes.data_as(fp), near.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(states=st[:got], distance=d[:got], normal=nr[:got], near=near[:got])

    def get_num_particles(self):
        if self._ctx is None:
            return self._n_added
        return int(self._check(self._L.mpmhip2d_num_particles(self._ctx)))

    def get_particles(self, sort_by_id=True):
        self._ensure_ctx()
        n = self.get_num_particles()
        out = dict(x=np.zeros((n, 2), np.float32), v=np.zeros((n, 2), np.float32), F=np.zeros((n, 4), np.float32),
                   B=np.zeros((n, 4), np.float32), aux=np.zeros(n, np.float32), gid=np.zeros(n, np.int32), id=np.zeros(n, np.int32))
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        got = self._check(self._L.mpmhip2d_download(self._ctx, n, out["x"].ctypes.data_as(fp), out["v"].ctypes.data_as(fp),
                                                    out["F"].ctypes.data_as(fp), out["B"].ctypes.data_as(fp), out["aux"].ctypes.data_as(fp),
                                                    out["gid"].ctypes.data_as(ip), out["id"].ctypes.data_as(ip)))
        out = {k: a[:got] for k, a in out.items()}
        if sort_by_id:
            o = np.argsort(out["id"], kind="stable")
            out = {k: a[o] for k, a in out.items()}
        return out

    def get_grid(self):
        self._ensure_ctx()
        g = np.zeros((self.res[0] + 1, self.res[1] + 1, 3), np.float32)
        self._check(self._L.mpmhip2d_download_grid(self._ctx, g.ctypes.data_as(C.POINTER(C.c_float))))
        return g

    def test(self):
        return True

    def get_name(self):
        return "mpm"

    def get_mpi_world_rank(self):
        return 0

    def get_debug_information(self):  # src/mpm.cpp:635-639
        return ""

    def get_vis_resolution(self):  # (scripts/async/async_mpm.py:79-81; see Simulation3D.get_vis_resolution)
        import types
        return types.SimpleNamespace(x=int(self.res[0]), y=int(self.res[1]))

    def add_articulation(self, cfg):
        """general_action(action='add_articulation', type='rotation', obj0=, obj1=) (src/mpm.cpp:923-933; the joint of
        scripts/mls-cpic/sand_wheel_2D.py:88): both bodies share one angular velocity.  The other joint types are 3D only."""
        if cfg.get("type") != "rotation":
            raise MPMError("add_articulation(type=%r): only 'rotation' is built for 2D simulations" % (cfg.get("type"),))
        if "obj0" not in cfg:
            raise MPMError("add_articulation needs 'obj0'")
        j = _lib.JointConfig()
        j.type, j.obj0, j.obj1 = 0, int(cfg["obj0"]), int(cfg.get("obj1", 0))
        self._ensure_ctx()
        self._check(self._L.mpmhip2d_add_articulation(self._ctx, C.byref(j)))
        return ""

    def general_action(self, config):
        """MPM<2>::general_action, src/mpm.cpp:920-978"""
        action = config.get("action")
        if action == "add_articulation":
            return self.add_articulation(config)
        if action == "save":  # src/mpm.cpp:940-949: whole-state snapshot
            self.save_snapshot(config["file_name"])
            return ""
        if action == "load":  # src/mpm.cpp:950-960
            self.load_snapshot(config["file_name"])
            return ""
        raise MPMError("general_action(%r) is not part of the 2D build" % (action,))

    def save_snapshot(self, path):
        """groups, particles, clocks, the rigid bodies' records and joints, an asynchronous stepper's pools and block table
        (include/mpmhip.h: mpmhip2d_snapshot_save)"""
        self._ensure_ctx()
        n = int(self._check(self._L.mpmhip2d_snapshot_size(self._ctx)))
        buf = np.empty(max(n, 1), np.uint8)
        self._check(self._L.mpmhip2d_snapshot_save(self._ctx, buf.ctypes.data_as(C.c_void_p), n))
        with open(path, "wb") as f:
            f.write(np.array([self.frame, self.frame_count], np.int64).tobytes())
            f.write(buf[:n].tobytes())

    def load_snapshot(self, path):
        """into a simulation set up with the same grid and scene (level set, configuration, the rigid bodies — added before the
        load — come from the script, as in the reference); replaces all particles and groups"""
        raw = np.fromfile(path, np.uint8)
        self.frame, self.frame_count = (int(v) for v in raw[:16].view(np.int64))
        blob = np.ascontiguousarray(raw[16:])
        self._ensure_ctx()
        self._check(self._L.mpmhip2d_snapshot_load(self._ctx, blob.ctypes.data_as(C.c_void_p), len(blob)))
        n_groups = int(blob[12:16].view(np.uint32)[0])
        off = C.sizeof(_lib.Snap2DHeader)
        rows = blob[off:off + 80 * n_groups].view(np.float32).reshape(n_groups, 20)
        self._groups = [(int(r[16:17].view(np.int32)[0]), r[:16].copy()) for r in rows]
        self._staged = []
        self._n_added = max(self._n_added, int(blob[40:48].view(np.int64)[0]))

    def write_partio(self, file_name):
        """MPM<2>::write_partio (src/visualize.cpp:17-100): the same Houdini .bgeo as the 3D simulation writes, z = 0
        (include/mpmhip.h: mpmhip2d_write_bgeo)"""
        self._ensure_ctx()
        self._check(self._L.mpmhip2d_write_bgeo(self._ctx, os.fsencode(file_name), int(self.verbose_bgeo)))

    def bgeo_bytes(self, verbose=None):
        """the .bgeo image of the current state as bytes (no file)"""
        self._ensure_ctx()
        verbose = int(self.verbose_bgeo if verbose is None else verbose)
        n = C.c_size_t()
        self._check(self._L.mpmhip2d_bgeo_size(self._ctx, verbose, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        w = C.c_size_t()
        self._check(self._L.mpmhip2d_bgeo_encode(self._ctx, verbose, buf.ctypes.data_as(C.c_void_p), n.value, C.byref(w)))
        return buf[:w.value].tobytes()

    def visualize(self):
        """MPM<2>::visualize -> write_bgeo (src/visualize.cpp:156-159, src/mpm.h:333-343): the next `frame_directory/%04d.bgeo`
        (frame numbers start at 1) and every rigid body's outline next to it"""
        if not self.frame_directory:
            raise MPMError("visualize() needs the config key 'frame_directory'")
        self.frame_count += 1
        os.makedirs(self.frame_directory, exist_ok=True)
        path = os.path.join(self.frame_directory, "%04d.bgeo" % self.frame_count)
        self.write_partio(path)
        for rid in range(1, len(self._rigids) + 1):
            self.write_rigid_body(rid, os.path.join(self.frame_directory, "rigid_%03d_%04d" % (rid, self.frame_count)))
        return path
