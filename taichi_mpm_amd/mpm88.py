"""The 2D dense-grid demo (BASELINE configs[0]) on the device: mirror of `mls-mpm88.cpp` (16-77).

    sim = MPM88()                         # n = 80, dt = 1e-4, plastic (snow) — the file's constants (:5-10)
    sim.add_object((0.55, 0.45))          # add_object(center, color): 1000 particles in a 0.16 x 0.16 square (:70-73)
    sim.advance(10)                       # advance(dt) x 10 (:16-69)
    x, v, F, C, Jp = sim.particles()

All arithmetic runs in the HIP kernels of libmpmhip (csrc/k_mpm88.h); there is no CPU path."""
import ctypes as C

import numpy as np

from . import _lib


class MPM88:
    def __init__(self, n=80, dt=1e-4, plastic=True, device=0, seed=88):
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.mpmhip_mpm88_create(int(n), float(dt), int(bool(plastic)), int(device), C.byref(self._h))
        if rc < 0:
            raise RuntimeError("mpmhip_mpm88_create: %s" % self._L.mpmhip_mpm88_last_error(None).decode())
        self.n, self.dt = int(n), float(dt)
        self._rng = np.random.default_rng(seed)

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError("libmpmhip (mpm88) error %d: %s" % (rc, self._L.mpmhip_mpm88_last_error(self._h).decode()))
        return rc

    def close(self):
        if self._h:
            self._L.mpmhip_mpm88_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_particles(self, x, v=None, F=None, C_=None, Jp=None):
        """explicit state (any of v, F, C, Jp may be None: the demo's constructor values)"""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
        n = len(x)

        def ptr(a, w):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32).reshape(n, w) if w > 1 else np.ascontiguousarray(a, np.float32).reshape(n)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_float))
        keep = []
        self._check(self._L.mpmhip_mpm88_add(self._h, n, x.ctypes.data_as(C.POINTER(C.c_float)), ptr(v, 2), ptr(F, 4),
                                             ptr(C_, 4), ptr(Jp, 1)))

    def add_object(self, center, count=1000):
        """add_object(center, c): `count` particles uniformly in center +- 0.08 (mls-mpm88.cpp:70-73)"""
        x = (self._rng.random((count, 2)) * 2 - 1) * 0.08 + np.asarray(center, np.float64)
        self.add_particles(x.astype(np.float32))

    def num_particles(self):
        return int(self._check(self._L.mpmhip_mpm88_num_particles(self._h)))

    def advance(self, steps=1):
        self._check(self._L.mpmhip_mpm88_advance(self._h, int(steps)))

    def particles(self):
        n = self.num_particles()
        x, v = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
        F, Cm, Jp = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self._L.mpmhip_mpm88_download(self._h, x.ctypes.data_as(fp), v.ctypes.data_as(fp), F.ctypes.data_as(fp),
                                                  Cm.ctypes.data_as(fp), Jp.ctypes.data_as(fp)))
        return x, v, F, Cm, Jp

    def grid(self):
        g = np.zeros((self.n + 1, self.n + 1, 3), np.float32)
        self._check(self._L.mpmhip_mpm88_download_grid(self._h, g.ctypes.data_as(C.POINTER(C.c_float))))
        return g
