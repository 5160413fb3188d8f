"""ctypes binding of oracle/_ref/libmpm_ref.so (TEST INFRASTRUCTURE ONLY).

libmpm_ref.so is the REFERENCE's own solver: /root/reference/src/{mpm,transfer,visualize,particles}.cpp compiled from the
sources where they lie (oracle/Makefile: ref_mpm) against oracle/taichi_shim, our stand-in for the un-vendored legacy
taichi core.  What that pins and what it cannot (svd / polar_decomp, the sampled level set) is stated in the header
of oracle/taichi_shim/taichi/common/util.h and in DESIGN.md §2.

The binding mirrors oracle/oracle.py (same State, same matrix layout: row-major float[9]) so a test can run the same
scene through the restated oracle, the reference and the HIP library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_ref", "libmpm_ref.so")
_lib = None

P_F = C.POINTER(C.c_float)
P_I = C.POINTER(C.c_int32)


def available():
    return os.path.exists(_LIB)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref_mpm"])


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libmpm_ref.so is missing: run `make -C oracle ref_mpm` where /root/reference exists")
        L = C.CDLL(_LIB)
        L.ref_last_error.restype = C.c_char_p
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_int, C.c_char_p]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_num_particles.restype = C.c_int64
        L.ref_num_particles.argtypes = [C.c_void_p]
        L.ref_download.restype = C.c_int64
        L.ref_download.argtypes = [C.c_void_p, P_F, P_F, P_F, P_F, P_F, P_I]
        L.ref_add_particles.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, P_F, P_F, P_F, P_F, P_F]
        L.ref_add_particles_cfg.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_set_levelset.argtypes = [C.c_void_p, C.c_int, P_F, C.c_int, P_F, C.c_float, C.c_float, C.c_float]
        L.ref_download_grid.argtypes = [C.c_void_p, P_F]
        L.ref_upload_grid.argtypes = [C.c_void_p, P_F]
        L.ref_substep.argtypes = [C.c_void_p, C.c_int]
        L.ref_step.argtypes = [C.c_void_p, C.c_float]
        L.ref_phase.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_time.restype = C.c_double
        L.ref_time.argtypes = [C.c_void_p]
        L.ref_set_time.argtypes = [C.c_void_p, C.c_double]
        L.ref_calculate_energy.restype = C.c_double
        L.ref_calculate_energy.argtypes = [C.c_void_p]
        L.ref_write_bgeo.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_general_action.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.ref_profile.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.ref_calculate_force.argtypes = [C.c_int, C.c_char_p, C.c_int64, P_F, P_F, P_F]
        L.ref_plasticity.argtypes = [C.c_int, C.c_char_p, C.c_int64, P_F, P_F, P_F]
        L.ref_particle_scalars.argtypes = [C.c_char_p, C.c_int64, P_F, P_F, P_F, C.c_float, P_F, P_F]
        L.ref_kernel3.argtypes = [C.c_int, P_F, C.c_float, P_F]
        L.ref_kernel2.argtypes = [P_F, C.c_float, P_F]
        L.ref_stencil_start.argtypes = [C.c_float]
        L.ref_friction_project.argtypes = [P_F, P_F, P_F, C.c_float, P_F]
        L.ref_shim_svd3.argtypes = [P_F, P_F, P_F, P_F]
        P_L = C.POINTER(C.c_int64)
        L.ref_async_update_dt_limits.argtypes = [C.c_void_p]
        L.ref_async_blocks.restype = C.c_int64
        L.ref_async_blocks.argtypes = [C.c_void_p, C.c_int64, P_I, P_L, P_L, P_L, P_L, P_L]
        L.ref_async_download.restype = C.c_int64
        L.ref_async_download.argtypes = [C.c_void_p, C.c_int64, P_F, P_F, P_F, P_F, P_F, P_I, P_L]
        L.ref_async_geometry.restype = C.c_int64
        L.ref_async_geometry.argtypes = [C.c_void_p, C.c_int64, P_I, P_I, P_I]
        L.ref_async_num_particles.restype = C.c_int64
        L.ref_async_num_particles.argtypes = [C.c_void_p]
        L.ref_async_time_int.restype = C.c_int64
        L.ref_async_time_int.argtypes = [C.c_void_p]
        L.ref_async_update_counter.restype = C.c_int64
        L.ref_async_update_counter.argtypes = [C.c_void_p]
        P_U = C.POINTER(C.c_uint32)
        L.ref_add_rigid.argtypes = [C.c_void_p, C.c_char_p, C.c_int, P_F, P_F]
        L.ref_rigid_state.argtypes = [C.c_void_p, C.c_int, P_F]
        L.ref_rigid_set_velocity.argtypes = [C.c_void_p, C.c_int, P_F, P_F]
        L.ref_rigid_samples.restype = C.c_int64
        L.ref_rigid_samples.argtypes = [C.c_void_p, C.c_int, C.c_int64, P_F, P_F, P_F, P_I]
        L.ref_download_cdf.argtypes = [C.c_void_p, P_U, P_F]
        L.ref_particle_cdf.restype = C.c_int64
        L.ref_particle_cdf.argtypes = [C.c_void_p, P_U, P_F, P_F, P_I, P_U]
        L.ref2_add_rigid.argtypes = [C.c_void_p, C.c_char_p, C.c_int, P_F, P_F]
        L.ref2_rigid_state.argtypes = [C.c_void_p, C.c_int, P_F]
        L.ref2_rigid_samples.restype = C.c_int64
        L.ref2_rigid_samples.argtypes = [C.c_void_p, C.c_int, C.c_int64, P_F]
        L.ref2_download_cdf.argtypes = [C.c_void_p, P_U, P_F]
        L.ref2_particle_cdf.restype = C.c_int64
        L.ref2_particle_cdf.argtypes = [C.c_void_p, P_U, P_F, P_F, P_I]
        if hasattr(L, "ref88_advance"):  # /root/reference/mls-mpm88.cpp compiled in place (oracle/ref_mpm88_driver.cpp)
            L.ref88_dt.restype = C.c_double
            L.ref88_num_particles.restype = C.c_int64
            L.ref88_reset.argtypes = [C.c_int32]
            L.ref88_add.argtypes = [C.c_int64, P_F, P_F, P_F, P_F, P_F]
            L.ref88_advance.argtypes = [C.c_int32]
            L.ref88_get.argtypes = [P_F, P_F, P_F, P_F, P_F]
            L.ref88_get_grid.argtypes = [P_F]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("reference: " + lib().ref_last_error().decode())


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(P_F)


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (tuple, list, np.ndarray)):
        return "(" + ",".join(_fmt(x) for x in v) + ")"
    if isinstance(v, (float, np.floating)):
        return "%.9g" % float(v)
    return str(v)


def cfg_string(**kw):
    return ";".join("%s=%s" % (k, _fmt(v)) for k, v in kw.items()).encode()


def rigid_script(position=None, velocity=(0, 0, 0), amplitude=(0, 0, 0), omega=0.0, rotation=None, rotation_rate=(0, 0, 0)):
    """the parametric scripts of ref_add_rigid: scripted_position(t) = position + velocity t + amplitude sin(omega t),
    scripted_rotation(t) = rotation + rotation_rate t (Euler angles in degrees); None = not scripted"""
    s = np.zeros(18, np.float32)
    if position is not None:
        s[0] = 1
        s[1:4], s[4:7], s[7:10], s[10] = position, velocity, amplitude, omega
    if rotation is not None:
        s[11] = 1
        s[12:15], s[15:18] = rotation, rotation_rate
    return s


def set_threads(n):
    return lib().ref_set_threads(int(n))


def type_config(type_name, mass, vol, **mat_kw):
    """the material's own Config keys (MPMParticle subclasses' initialize(), src/particles.cpp) + mass / vol"""
    return cfg_string(type=type_name, mass=mass, vol=vol, **mat_kw)


class Sim:
    """MPM<dim> of the reference (create_simulation3('mpm') / create_simulation2('mpm'))"""

    def __init__(self, res, dx, dt, dim=3, gravity=None, shapes=(), friction=1.0, **cfg):
        self.dim = dim
        if np.isscalar(res):
            res = (int(res),) * dim
        self.res = tuple(int(r) for r in res)
        if gravity is None:
            gravity = (0, -10, 0)[:dim]
        s = cfg_string(res=self.res, delta_x=dx, base_delta_t=dt, gravity=tuple(gravity), **cfg)
        self.h = lib().ref_create(dim, s)
        if not self.h:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        self.set_levelset(shapes, friction)

    def close(self):
        if self.h:
            lib().ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_levelset(self, shapes=(), friction=1.0, shapes1=None, t0=0.0, t1=1.0):
        """shapes: rows (type, inside_out, p0..p5) in WORLD units: type 0 plane (n, d), 1 sphere (c, r), 2 cuboid
        (lo, hi).  shapes1 (with t0 < t1): the level set at time t1 — DynamicLevelSet(t0, t1, shapes, shapes1), blended
        linearly in time like the reference's (scripts/async/async_mpm.py:119-127)."""
        def rows(sh):
            r = np.zeros((len(sh), 8), np.float32)
            for i, row in enumerate(sh):
                row = [float(v) for v in row]
                r[i, :len(row)] = row
            return r
        r0 = rows(shapes)
        r1 = rows(shapes1) if shapes1 is not None else np.zeros((0, 8), np.float32)
        _chk(lib().ref_set_levelset(self.h, len(r0), r0.ctypes.data_as(P_F), len(r1) if shapes1 is not None else -1,
                                    r1.ctypes.data_as(P_F), C.c_float(t0), C.c_float(t1), C.c_float(friction)))

    def add_particles(self, type_name, mass, vol, x, v=None, F=None, B=None, aux=None, **mat_kw):
        d = self.dim
        x, xp = _f(np.asarray(x).reshape(-1, d))
        n = len(x)

        def opt(a, w):
            if a is None:
                return None, None
            a, p = _f(np.asarray(a).reshape(n, w) if w > 1 else np.asarray(a).reshape(n))
            return a, p
        v, vp = opt(v, d); F, Fp = opt(F, d * d); B, Bp = opt(B, d * d); aux, ap = opt(aux, 1)
        _chk(lib().ref_add_particles(self.h, type_config(type_name, mass, vol, **mat_kw), n, xp, vp, Fp, Bp, ap))

    def add_benchmark(self, type_name, benchmark, **mat_kw):
        """the reference's own lattice generator, src/mpm.cpp:149-186 (benchmark = 125 or 8000)"""
        _chk(lib().ref_add_particles_cfg(self.h, cfg_string(type=type_name, benchmark=benchmark, **mat_kw)))

    def num_particles(self):
        return int(lib().ref_num_particles(self.h))

    def download(self, by_id=True):
        n, d = self.num_particles(), self.dim
        out = dict(x=np.zeros((n, d), np.float32), v=np.zeros((n, d), np.float32), F=np.zeros((n, d * d), np.float32),
                   B=np.zeros((n, d * d), np.float32), aux=np.zeros(n, np.float32), id=np.zeros(n, np.int32))
        m = lib().ref_download(self.h, *(out[k].ctypes.data_as(P_F) for k in ("x", "v", "F", "B", "aux")),
                               out["id"].ctypes.data_as(P_I))
        if m != n:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        if by_id:
            o = np.argsort(out["id"], kind="stable")
            out = {k: a[o] for k, a in out.items()}
        return out

    def grid_shape(self):
        return tuple(r + 1 for r in self.res) + (self.dim + 1,)

    def download_grid(self):
        g = np.zeros(self.grid_shape(), np.float32)
        _chk(lib().ref_download_grid(self.h, g.ctypes.data_as(P_F)))
        return g

    def upload_grid(self, g):
        g, gp = _f(np.asarray(g).reshape(self.grid_shape()))
        _chk(lib().ref_upload_grid(self.h, gp))

    def substep(self, n=1):
        _chk(lib().ref_substep(self.h, int(n)))

    def step(self, dt):
        _chk(lib().ref_step(self.h, C.c_float(dt)))

    def sort(self):
        _chk(lib().ref_phase(self.h, 0, 1))

    def p2g(self, optimized=True):
        _chk(lib().ref_phase(self.h, 1, int(optimized)))

    def grid_update(self):
        _chk(lib().ref_phase(self.h, 2, 1))

    def g2p(self, optimized=True):
        _chk(lib().ref_phase(self.h, 3, int(optimized)))

    def clear_boundary_particles(self):
        _chk(lib().ref_phase(self.h, 4, 1))

    def particle_collision(self):
        _chk(lib().ref_phase(self.h, 5, 1))

    # ---- CPIC rigid coupling (3D): src/rigid_transfer.cpp, src/mpm_rigid_body.cpp, rigid branches of src/transfer.cpp
    def add_rigid(self, triangles, script=None, **cfg):
        """add_particles(type='rigid', ...): triangles (n, 3, 3) in mesh space; cfg = the reference's keys (codimensional,
        density, friction, scale, initial_position, initial_rotation, ...); script = rigid_script(...) or None.
        Returns the body's index (>= 1)."""
        tri, tp = _f(np.asarray(triangles, np.float32).reshape(-1, 9))
        sp = None
        if script is not None:
            script, sp = _f(np.asarray(script, np.float32).reshape(18))
        cfg.setdefault("codimensional", True)
        rid = lib().ref_add_rigid(self.h, cfg_string(**cfg), len(tri), tp, sp)
        if rid < 0:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        return rid

    def rigid_state(self, rid):
        o = np.zeros(33, np.float32)
        _chk(lib().ref_rigid_state(self.h, int(rid), o.ctypes.data_as(P_F)))
        return dict(position=o[0:3], rotation=o[3:7], velocity=o[7:10], angular_velocity=o[10:13], mass=float(o[13]),
                    inv_mass=float(o[14]), inertia=o[15:24].reshape(3, 3), inv_inertia=o[24:33].reshape(3, 3))

    def rigid_set_velocity(self, rid, v=None, w=None):
        vp = _f(v)[1] if v is not None else None
        wp = _f(w)[1] if w is not None else None
        _chk(lib().ref_rigid_set_velocity(self.h, int(rid), vp, wp))

    def rigid_samples(self, rid=-1):
        """boundary particles: world position, body-frame offset, untransformed triangle, body index"""
        n = lib().ref_rigid_samples(self.h, int(rid), 0, None, None, None, None)
        pos, off, el, body = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3, 3), np.float32), np.zeros(n, np.int32)
        m = lib().ref_rigid_samples(self.h, int(rid), n, pos.ctypes.data_as(P_F), off.ctypes.data_as(P_F), el.ctypes.data_as(P_F),
                                    body.ctypes.data_as(P_I))
        assert m == n
        return dict(pos=pos, offset=off, element=el, body=body)

    # the same in 2D (MPM<2>)
    def add_rigid2(self, segments, script=None, **cfg):
        """segments (n, 2, 2); script = [has_pos, p0(2), vel(2), has_rot, a0 deg, rate deg/s] or None"""
        seg, sp_ = _f(np.asarray(segments, np.float32).reshape(-1, 4))
        sp = None
        if script is not None:
            script, sp = _f(np.asarray(script, np.float32).reshape(8))
        cfg.setdefault("codimensional", True)
        rid = lib().ref2_add_rigid(self.h, cfg_string(**cfg), len(seg), sp_, sp)
        if rid < 0:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        return rid

    def rigid_state2(self, rid):
        o = np.zeros(10, np.float32)
        _chk(lib().ref2_rigid_state(self.h, int(rid), o.ctypes.data_as(P_F)))
        return o

    def rigid_samples2(self, rid=-1):
        n = lib().ref2_rigid_samples(self.h, int(rid), 0, None)
        pos = np.zeros((n, 2), np.float32)
        assert lib().ref2_rigid_samples(self.h, int(rid), n, pos.ctypes.data_as(P_F)) == n
        return pos

    def download_cdf2(self):
        shp = tuple(r + 1 for r in self.res)
        st, d = np.zeros(shp, np.uint32), np.zeros(shp, np.float32)
        _chk(lib().ref2_download_cdf(self.h, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(P_F)))
        return st, d

    def particle_cdf2(self):
        n = self.num_particles()
        st, d, nr, near = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.int32)
        assert lib().ref2_particle_cdf(self.h, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(P_F), nr.ctypes.data_as(P_F),
                                       near.ctypes.data_as(P_I)) == n
        return dict(states=st, distance=d, normal=nr, near=near)

    def rasterize_rigid_boundary(self):
        _chk(lib().ref_phase(self.h, 7, 1))

    def gather_cdf(self):
        _chk(lib().ref_phase(self.h, 8, 1))

    def advect_rigid_bodies(self):
        _chk(lib().ref_phase(self.h, 9, 1))

    def download_cdf(self):
        """(states, distance) of every grid node, dense (res+1)^3; states = 24 colour-tag bits | (rigid id + 1) << 24"""
        shp = tuple(r + 1 for r in self.res)
        st, d = np.zeros(shp, np.uint32), np.zeros(shp, np.float32)
        _chk(lib().ref_download_cdf(self.h, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(P_F)))
        return st, d

    def particle_cdf(self, upload_states=None):
        """per material particle in the order of download(by_id=False): states, boundary_distance, boundary_normal, near_boundary"""
        n = self.num_particles()
        st, d, nr, near = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        up = None
        if upload_states is not None:
            upload_states = np.ascontiguousarray(upload_states, np.uint32)
            up = upload_states.ctypes.data_as(C.POINTER(C.c_uint32))
        m = lib().ref_particle_cdf(self.h, st.ctypes.data_as(C.POINTER(C.c_uint32)), d.ctypes.data_as(P_F), nr.ctypes.data_as(P_F),
                                   near.ctypes.data_as(P_I), up)
        assert m == n
        return dict(states=st, distance=d, normal=nr, near=near)

    def time(self):
        return lib().ref_time(self.h)

    def set_time(self, t):
        lib().ref_set_time(self.h, float(t))

    def calculate_energy(self):
        return lib().ref_calculate_energy(self.h)

    def write_bgeo(self, path):
        _chk(lib().ref_write_bgeo(self.h, os.fsencode(path)))

    def general_action(self, **kw):
        buf = C.create_string_buffer(4096)
        _chk(lib().ref_general_action(self.h, cfg_string(**kw), buf, 4096))
        return buf.value.decode()


class AsyncSim(Sim):
    """AsyncMPM<dim> of the reference (create_simulation3('async_mpm') / create_simulation2('async_mpm'),
    src/async/async_mpm.{h,cpp}): block-local time steps.  Config keys: unit_delta_t, max_units, cfl_dt_mul, strength_dt_mul
    (src/async/async_mpm.cpp:24-27), left_boundary (:43-53)."""

    def __init__(self, res, dx, dt=1e-4, dim=3, **cfg):
        super().__init__(res, dx, dt, dim=dim, **{"async": True, **cfg})

    def geometry(self):
        """the scheduler's block table in the order the pools are walked in: corner node (n,3), cached_neighbours (n,26;
        -1 terminated), left_boundary flag (n)"""
        n = int(lib().ref_async_geometry(self.h, 0, None, None, None))
        coord, neigh, bnd = np.zeros((n, 3), np.int32), np.zeros((n, 26), np.int32), np.zeros(n, np.int32)
        if lib().ref_async_geometry(self.h, n, coord.ctypes.data_as(P_I), neigh.ctypes.data_as(P_I), bnd.ctypes.data_as(P_I)) != n:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        return coord, neigh, bnd

    def update_dt_limits(self):
        _chk(lib().ref_async_update_dt_limits(self.h))

    def blocks(self):
        """non-empty scheduler blocks: dict(coord (n,3) node coordinates of the block corner, strength, cfl, continuous,
        count) + (min_delta_t_int, max_delta_t_int)"""
        cap = 1 << 20
        coord = np.zeros((cap, 3), np.int32)
        arrs = [np.zeros(cap, np.int64) for _ in range(4)]
        mm = np.zeros(2, np.int64)
        P_L = C.POINTER(C.c_int64)
        n = lib().ref_async_blocks(self.h, cap, coord.ctypes.data_as(P_I), *(a.ctypes.data_as(P_L) for a in arrs), mm.ctypes.data_as(P_L))
        if n < 0:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        return dict(coord=coord[:n], strength=arrs[0][:n], cfl=arrs[1][:n], continuous=arrs[2][:n], count=arrs[3][:n]), tuple(mm)

    def num_particles(self):
        return int(lib().ref_async_num_particles(self.h))

    def download(self, by_id=True):
        n = self.num_particles()
        d = self.dim
        out = dict(x=np.zeros((n, d), np.float32), v=np.zeros((n, d), np.float32), F=np.zeros((n, d * d), np.float32),
                   B=np.zeros((n, d * d), np.float32), aux=np.zeros(n, np.float32), id=np.zeros(n, np.int32), limits=np.zeros((n, 4), np.int64))
        m = lib().ref_async_download(self.h, n, *(out[k].ctypes.data_as(P_F) for k in ("x", "v", "F", "B", "aux")),
                                     out["id"].ctypes.data_as(P_I), out["limits"].ctypes.data_as(C.POINTER(C.c_int64)))
        if m != n:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())
        if by_id:
            o = np.argsort(out["id"], kind="stable")
            out = {k: a[o] for k, a in out.items()}
        return out

    def time_int(self):
        return int(lib().ref_async_time_int(self.h))

    def update_counter(self):
        return int(lib().ref_async_update_counter(self.h))


def profile(reset=False):
    """seconds per TC_PROFILE name of the reference ('P2G optimized', 'G2P optimized', 'parallel_sort', ...)"""
    buf = C.create_string_buffer(1 << 16)
    lib().ref_profile(buf, 1 << 16, int(reset))
    out = {}
    for kv in buf.value.decode().split(";"):
        if "=" in kv:
            k, v = kv.rsplit("=", 1)
            out[k] = float(v)
    return out


# ----------------------------------------------------------------------------- single-particle entry points
def calculate_force(type_name, mass, vol, F, aux=None, dim=3, **mat_kw):
    F, fp = _f(np.asarray(F).reshape(-1, dim * dim))
    n = len(F)
    aux, ap = _f(np.zeros(n) if aux is None else np.broadcast_to(np.asarray(aux, np.float32), (n,)))
    out = np.zeros_like(F)
    _chk(lib().ref_calculate_force(dim, type_config(type_name, mass, vol, **mat_kw), n, fp, ap, out.ctypes.data_as(P_F)))
    return out


def plasticity(type_name, mass, vol, cdg, F, aux=None, dim=3, **mat_kw):
    F = np.array(F, np.float32).reshape(-1, dim * dim).copy()
    n = len(F)
    cdg, cp = _f(np.asarray(cdg).reshape(n, dim * dim))
    aux = np.array(np.zeros(n) if aux is None else np.broadcast_to(np.asarray(aux, np.float32), (n,)), np.float32).copy()
    _chk(lib().ref_plasticity(dim, type_config(type_name, mass, vol, **mat_kw), n, cp, F.ctypes.data_as(P_F), aux.ctypes.data_as(P_F)))
    return F, aux


def particle_scalars(type_name, mass, vol, F, aux, v, dx, **mat_kw):
    """(get_allowed_dt(dx), potential_energy()) per particle state; NaN where the type has no potential_energy()"""
    F, fp = _f(np.asarray(F).reshape(-1, 9))
    n = len(F)
    aux, ap = _f(np.broadcast_to(np.asarray(aux, np.float32), (n,)))
    v, vp = _f(np.asarray(v).reshape(n, 3))
    adt = np.zeros(n, np.float32); pot = np.zeros(n, np.float32)
    _chk(lib().ref_particle_scalars(type_config(type_name, mass, vol, **mat_kw), n, fp, ap, vp, C.c_float(dx),
                                    adt.ctypes.data_as(P_F), pot.ctypes.data_as(P_F)))
    return adt, pot


def kernel3_dw_w(pos, inv_dx=1.0, fast=False):
    p, pp = _f(pos)
    out = np.zeros((27, 4), np.float32)
    _chk(lib().ref_kernel3(int(fast), pp, C.c_float(inv_dx), out.ctypes.data_as(P_F)))
    return out


def kernel2_dw_w(pos, inv_dx=1.0):
    p, pp = _f(pos)
    out = np.zeros((9, 3), np.float32)
    _chk(lib().ref_kernel2(pp, C.c_float(inv_dx), out.ctypes.data_as(P_F)))
    return out


def stencil_start(x):
    return int(lib().ref_stencil_start(C.c_float(x)))


def friction_project(v, vb, n, mu):
    v, vp = _f(v); vb, vbp = _f(vb); n, np_ = _f(n)
    out = np.zeros(3, np.float32)
    _chk(lib().ref_friction_project(vp, vbp, np_, C.c_float(mu), out.ctypes.data_as(P_F)))
    return out


def shim_svd3(F):
    F, fp = _f(np.asarray(F).reshape(9))
    U = np.zeros(9, np.float32); S = np.zeros(3, np.float32); V = np.zeros(9, np.float32)
    lib().ref_shim_svd3(fp, U.ctypes.data_as(P_F), S.ctypes.data_as(P_F), V.ctypes.data_as(P_F))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def mpm88_available():
    return available() and hasattr(lib(), "ref88_advance")


def mpm88_advance(x, v, F, Cm, Jp, steps=1, plastic=True):
    """the reference's 2D demo itself: /root/reference/mls-mpm88.cpp:16-69 advance(dt) with its own constants (n = 80,
    dt = 1e-4) run `steps` times on the given particles (x, v: n x 2; F, C: n x 4 row-major; Jp: n), updated IN PLACE
    like oracle.mpm88_advance.  Returns the grid (v.x, v.y, m) of the last step, shape (81, 81, 3).  The state lives
    in the file's globals: one scene at a time."""
    L = lib()
    n = len(x)
    L.ref88_reset(1 if plastic else 0)
    L.ref88_add(n, _f(x)[1], _f(v)[1], _f(F)[1], _f(Cm)[1], _f(Jp)[1])
    L.ref88_advance(int(steps))
    for a in (x, v, F, Cm, Jp):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    L.ref88_get(x.ctypes.data_as(P_F), v.ctypes.data_as(P_F), F.ctypes.data_as(P_F), Cm.ctypes.data_as(P_F), Jp.ctypes.data_as(P_F))
    ng = L.ref88_grid_cells()
    g = np.zeros((ng + 1, ng + 1, 3), np.float32)
    L.ref88_get_grid(g.ctypes.data_as(P_F))
    return g
