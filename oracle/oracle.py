"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

PARITY STATUS: "parity unpinned" except for the B-spline kernel weights — see
oracle/mpm_oracle.h.  Each wrapped function cites the reference file:line in the C++ source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

VISCO, SNOW, LINEAR, JELLY, WATER, SAND, VON_MISES, ELASTIC = 1, 2, 3, 4, 5, 6, 7, 8
NPARAM = 16
TYPE_IDS = {"visco": VISCO, "snow": SNOW, "linear": LINEAR, "jelly": JELLY, "water": WATER,
            "sand": SAND, "von_mises": VON_MISES, "elastic": ELASTIC}


class Shape(C.Structure):
    _fields_ = [("type", C.c_int32), ("inside_out", C.c_int32), ("p", C.c_float * 6)]


class Config(C.Structure):
    _fields_ = [("res", C.c_int32 * 3), ("dx", C.c_float), ("dt", C.c_float),
                ("gravity", C.c_float * 3), ("particle_gravity", C.c_int32),
                ("apic_damping", C.c_float), ("rpic_damping", C.c_float),
                ("clean_boundary", C.c_int32), ("n_planes", C.c_int32),
                ("planes", (C.c_float * 4) * 8), ("friction", C.c_float), ("n_shapes", C.c_int32),
                ("shapes", Shape * 8), ("particle_collision", C.c_int32),
                ("dynamic", C.c_int32), ("t0", C.c_float), ("t1", C.c_float), ("t", C.c_float),
                ("n_planes1", C.c_int32), ("planes1", (C.c_float * 4) * 8), ("n_shapes1", C.c_int32), ("shapes1", Shape * 8)]


def build():
    """(Re)build liboracle.so with the Makefile next to this file."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_opt_run.restype = C.c_double
        _lib.orc_substep.restype = C.c_int64
        _lib.orc_clear_boundary.restype = C.c_int64
    return _lib


def make_config(res, dx, dt, gravity=(0, -10, 0), particle_gravity=True, apic_damping=0.0,
                rpic_damping=0.0, clean_boundary=True, planes=(), friction=-1.0, shapes=(), particle_collision=False,
                planes1=None, shapes1=None, t0=0.0, t1=1.0, t=0.0):
    """shapes: [(type, inside_out, params...)] with type 1 = sphere (cx, cy, cz, r), 2 = cuboid (lo xyz, hi xyz).
    planes1 / shapes1 (either not None): the level set at time t1 — a DynamicLevelSet(t0, t1, ...) blended linearly in
    time; `t` is the current time, advanced by substep()."""
    c = Config()
    if np.isscalar(res):
        res = (res,) * 3
    c.res[:] = [int(r) for r in res]
    c.dx = dx
    c.dt = dt
    c.gravity[:] = [float(g) for g in gravity]
    c.particle_gravity = int(bool(particle_gravity))
    c.apic_damping = apic_damping
    c.rpic_damping = rpic_damping
    c.clean_boundary = int(bool(clean_boundary))
    c.n_planes = len(planes)
    for i, p in enumerate(planes):
        c.planes[i][:] = [float(v) for v in p]
    c.friction = friction
    c.n_shapes = len(shapes)
    for i, sh in enumerate(shapes):
        c.shapes[i].type, c.shapes[i].inside_out = int(sh[0]), int(bool(sh[1]))
        vals = [float(v) for v in sh[2:]]
        c.shapes[i].p[:] = vals + [0.0] * (6 - len(vals))
    c.particle_collision = int(bool(particle_collision))
    c.dynamic = int(planes1 is not None or shapes1 is not None)
    c.t0, c.t1, c.t = t0, t1, t
    c.n_planes1 = len(planes1 or ())
    for i, p in enumerate(planes1 or ()):
        c.planes1[i][:] = [float(v) for v in p]
    c.n_shapes1 = len(shapes1 or ())
    for i, sh in enumerate(shapes1 or ()):
        c.shapes1[i].type, c.shapes1[i].inside_out = int(sh[0]), int(bool(sh[1]))
        vals = [float(v) for v in sh[2:]]
        c.shapes1[i].p[:] = vals + [0.0] * (6 - len(vals))
    return c


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def group_params(type_name, mass, vol, **kw):
    """float[16] parameter row (layout: mpm_oracle.h) with the reference's defaults
    (src/particles.cpp initialize() of each material)."""
    p = np.zeros(NPARAM, np.float32)
    p[0], p[1] = mass, vol
    t = TYPE_IDS[type_name]

    def lame(E, nu):
        return E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    if t == SNOW:  # particles.cpp:192-205
        E, nu = kw.get("youngs_modulus", 1.4e5), kw.get("poisson_ratio", 0.2)
        mu, lam = lame(E, nu)
        p[2], p[3] = kw.get("mu_0", mu), kw.get("lambda_0", lam)
        p[4] = kw.get("hardening", 10.0)
        p[5], p[6] = kw.get("theta_c", 2.5e-2), kw.get("theta_s", 7.5e-3)
        p[7], p[8] = kw.get("min_Jp", 0.6), kw.get("max_Jp", 20.0)
    elif t in (LINEAR, JELLY):  # :315-321, :383-389
        p[2], p[3] = lame(kw.get("E", 1e5), kw.get("nu", 0.3))
    elif t == WATER:  # :448-461
        p[2], p[3] = kw.get("k", 10000.0), kw.get("gamma", 7.0)
    elif t == SAND:  # :570-597
        p[2], p[3] = kw.get("mu_0", 136038.0), kw.get("lambda_0", 204057.0)
        fa = kw.get("friction_angle", 30.0)
        sin_phi = np.sin(np.float32(fa) / np.float32(180.0) * np.float32(3.141592653))
        p[4] = np.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi)
        p[5], p[6] = kw.get("cohesion", 0.0), kw.get("beta", 1.0)
    elif t == VON_MISES:  # :691-699
        p[2], p[3] = lame(kw.get("youngs_modulus", 5e3), kw.get("poisson_ratio", 0.4))
        p[4] = kw.get("yield_stress", 1.0)
    elif t == ELASTIC:  # :777-783
        p[2], p[3] = lame(kw.get("E", 5e3), kw.get("nu", 0.4))
        p[4] = kw.get("E", 5e3)  # `E` member, read back only by get_debug_info() (:838-840)
    elif t == VISCO:  # :57-70
        p[2], p[3] = lame(kw.get("youngs_modulus", 4e4), kw.get("poisson_ratio", 0.4))
        p[4], p[5], p[6] = kw.get("nu", 10000.0), kw.get("kappa", 0.0), kw.get("base_delta_t", 1e-4)
    return p, t


def initial_aux(type_name, **kw):
    t = TYPE_IDS[type_name]
    if t == SNOW:
        return kw.get("Jp", 1.0)
    if t == WATER:
        return 1.0
    if t == VISCO:
        return kw.get("tau", 1000.0)
    return 0.0


# ----------------------------------------------------------------------------- kernels
def kernel3_dw_w(pos, inv_dx=1.0, slow=False):
    p, pp = _f(pos)
    out = np.zeros((27, 4), np.float32)
    fn = lib().orc_kernel3_dw_w_slow if slow else lib().orc_kernel3_dw_w
    fn(pp, C.c_float(inv_dx), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def mls_kernel3_w(rel_pos):
    p, pp = _f(rel_pos)
    out = np.zeros(27, np.float32)
    lib().orc_mls_kernel3_w(pp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def kernel2_dw_w(pos, inv_dx=1.0):
    p, pp = _f(pos)
    out = np.zeros((9, 3), np.float32)
    lib().orc_kernel2_dw_w(pp, C.c_float(inv_dx), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def kernel_cubic_dw_w(pos, inv_dx=1.0):
    p, pp = _f(pos)
    dim = len(p)
    out = np.zeros((4 ** dim, dim + 1), np.float32)
    lib().orc_kernel_cubic_dw_w(dim, pp, C.c_float(inv_dx), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


# ----------------------------------------------------------------------------- svd / polar
def svd3(F):
    F, fp = _f(np.asarray(F).reshape(9))
    U = np.zeros(9, np.float32); S = np.zeros(3, np.float32); V = np.zeros(9, np.float32)
    P = C.POINTER(C.c_float)
    lib().orc_svd3(fp, U.ctypes.data_as(P), S.ctypes.data_as(P), V.ctypes.data_as(P))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def polar3(F):
    F, fp = _f(np.asarray(F).reshape(9))
    R = np.zeros(9, np.float32); S = np.zeros(9, np.float32)
    P = C.POINTER(C.c_float)
    lib().orc_polar3(fp, R.ctypes.data_as(P), S.ctypes.data_as(P))
    return R.reshape(3, 3), S.reshape(3, 3)


def svd2(F):
    F, fp = _f(np.asarray(F).reshape(4))
    U = np.zeros(4, np.float32); S = np.zeros(2, np.float32); V = np.zeros(4, np.float32)
    P = C.POINTER(C.c_float)
    lib().orc_svd2(fp, U.ctypes.data_as(P), S.ctypes.data_as(P), V.ctypes.data_as(P))
    return U.reshape(2, 2), S, V.reshape(2, 2)


def polar2(F):
    F, fp = _f(np.asarray(F).reshape(4))
    R = np.zeros(4, np.float32); S = np.zeros(4, np.float32)
    P = C.POINTER(C.c_float)
    lib().orc_polar2(fp, R.ctypes.data_as(P), S.ctypes.data_as(P))
    return R.reshape(2, 2), S.reshape(2, 2)


# ----------------------------------------------------------------------------- materials
def calculate_force(type_id, gp, F, aux=0.0):
    gp, gpp = _f(gp)
    F, fp = _f(np.asarray(F).reshape(9))
    out = np.zeros(9, np.float32)
    lib().orc_calculate_force(int(type_id), gpp, fp, C.c_float(aux), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(3, 3)


def plasticity(type_id, gp, cdg, F, aux=0.0):
    gp, gpp = _f(gp)
    cdg, cp = _f(np.asarray(cdg).reshape(9))
    F = np.array(F, dtype=np.float32).reshape(9).copy()
    a = C.c_float(aux)
    lib().orc_plasticity(int(type_id), gpp, cp, F.ctypes.data_as(C.POINTER(C.c_float)), C.byref(a))
    return F.reshape(3, 3), a.value


def friction_project(v, vb, n, mu):
    v, vp = _f(v); vb, vbp = _f(vb); n, np_ = _f(n)
    out = np.zeros(3, np.float32)
    lib().orc_friction_project(vp, vbp, np_, C.c_float(mu), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


# ----------------------------------------------------------------------------- phases
class State:
    """SoA particle state (numpy fp32) + group table, the layout every oracle phase takes."""

    def __init__(self, x, v=None, B=None, F=None, aux=None, gid=None, gparams=None, gtype=None, ids=None):
        n = len(x)
        self.x = np.ascontiguousarray(x, np.float32).reshape(n, 3).copy()
        self.v = np.zeros((n, 3), np.float32) if v is None else np.ascontiguousarray(v, np.float32).reshape(n, 3).copy()
        self.B = np.zeros((n, 9), np.float32) if B is None else np.ascontiguousarray(B, np.float32).reshape(n, 9).copy()
        if F is None:
            self.F = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1))
        else:
            self.F = np.ascontiguousarray(F, np.float32).reshape(n, 9).copy()
        self.aux = np.zeros(n, np.float32) if aux is None else np.ascontiguousarray(aux, np.float32).copy()
        self.gid = np.zeros(n, np.int32) if gid is None else np.ascontiguousarray(gid, np.int32).copy()
        self.gparams = np.ascontiguousarray(gparams, np.float32).reshape(-1, NPARAM).copy()
        self.gtype = np.ascontiguousarray(gtype, np.int32).copy()
        self.ids = np.arange(n, dtype=np.int32) if ids is None else np.ascontiguousarray(ids, np.int32).copy()

    @property
    def n(self):
        return len(self.x)

    def copy(self):
        return State(self.x, self.v, self.B, self.F, self.aux, self.gid, self.gparams, self.gtype, self.ids)

    def select(self, mask):
        """the particles where mask is True, as a new State (same group table)"""
        m = np.asarray(mask, bool)
        return State(self.x[m], self.v[m], self.B[m], self.F[m], self.aux[m], self.gid[m], self.gparams, self.gtype, self.ids[m])

    def truncate(self, n):
        for k in ("x", "v", "B", "F", "aux", "gid", "ids"):
            setattr(self, k, getattr(self, k)[:n].copy())


def grid_shape(cfg):
    return (cfg.res[0] + 1, cfg.res[1] + 1, cfg.res[2] + 1, 4)


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def p2g(cfg, s):
    grid = np.zeros(grid_shape(cfg), np.float32)
    lib().orc_p2g(C.byref(cfg), C.c_int64(s.n), _pf(s.x), _pf(s.v), _pf(s.B), _pf(s.F), _pf(s.aux),
                  _pi(s.gid), _pf(s.gparams), _pi(s.gtype), _pf(grid))
    return grid


def grid_update(cfg, grid):
    lib().orc_grid_update(C.byref(cfg), _pf(grid))
    return grid


def g2p(cfg, s, grid):
    grid = np.ascontiguousarray(grid, np.float32)
    lib().orc_g2p(C.byref(cfg), C.c_int64(s.n), _pf(s.x), _pf(s.v), _pf(s.B), _pf(s.F), _pf(s.aux),
                  _pi(s.gid), _pf(s.gparams), _pi(s.gtype), _pf(grid))


def particle_collision(cfg, s):
    lib().orc_particle_collision(C.byref(cfg), C.c_int64(s.n), _pf(s.x), _pf(s.v))


def delete_inside_levelset(cfg, s):
    """-> keep mask (bool) of general_action 'delete_particles_inside_level_set' (src/mpm.cpp:962-974)"""
    keep = np.zeros(s.n, np.uint8)
    lib().orc_delete_inside_levelset.restype = C.c_int64
    lib().orc_delete_inside_levelset(C.byref(cfg), C.c_int64(s.n), _pf(s.x), keep.ctypes.data_as(C.POINTER(C.c_uint8)))
    return keep.astype(bool)


def clear_boundary(cfg, s):
    keep = np.zeros(s.n, np.uint8)
    lib().orc_clear_boundary(C.byref(cfg), C.c_int64(s.n), _pf(s.x), _pf(s.v),
                             keep.ctypes.data_as(C.POINTER(C.c_uint8)))
    return keep.astype(bool)


def substep(cfg, s, grid=None):
    if grid is None:
        grid = np.zeros(grid_shape(cfg), np.float32)
    n = lib().orc_substep(C.byref(cfg), C.c_int64(s.n), _pf(s.x), _pf(s.v), _pf(s.B), _pf(s.F), _pf(s.aux),
                          _pi(s.gid), _pi(s.ids), _pf(s.gparams), _pi(s.gtype), _pf(grid))
    if n != s.n:
        s.truncate(n)
    cfg.t = float(np.float32(cfg.t) + np.float32(cfg.dt))  # this->current_t += delta_t, src/mpm.cpp:573
    return grid


def opt_run(cfg, s, steps, threads=0):
    """Timed block-sorted CPU path (cpu_baseline). Returns (seconds, phase_seconds[4])."""
    ph = (C.c_double * 4)()
    t = lib().orc_opt_run(C.byref(cfg), C.c_int64(s.n), _pf(s.x), _pf(s.v), _pf(s.B), _pf(s.F), _pf(s.aux),
                          _pi(s.gid), _pf(s.gparams), _pi(s.gtype), int(steps), int(threads), ph)
    return t, list(ph)


def mpm88_advance(n_grid, dt, x, v, F, Cm, Jp, plastic=True):
    grid = np.zeros(((n_grid + 1), (n_grid + 1), 3), np.float32)
    lib().orc_mpm88_advance(int(n_grid), C.c_float(dt), C.c_int64(len(x)), _pf(x), _pf(v), _pf(F), _pf(Cm),
                            _pf(Jp), _pf(grid), int(plastic))
    return grid
