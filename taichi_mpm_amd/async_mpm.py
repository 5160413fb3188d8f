"""AsyncMPM — the reference's asynchronous (block-local time step) stepper, `create_simulation3('async_mpm')` and
`create_simulation2('async_mpm')` (class AsyncMPM<dim>, src/async/async_mpm.{h,cpp}; TC_IMPLEMENTATION(Simulation3D,
AsyncMPM3D, "async_mpm") / (Simulation2D, AsyncMPM2D, "async_mpm"), :423-427).

Everything of it runs inside libmpmhip (csrc/async_sched.h: the block scheduler of both dimensions; csrc/async_api.h,
csrc/k_async.h: 3D; csrc/async2d_api.h, csrc/k_async2d.h: 2D; C ABI: include/mpmhip.h "AsyncMPM, second half", "AsyncMPM<2>"):
the particle pools and backup pools of every scheduler block are a device-resident store, `advance(limit)` gathers its
working set with index kernels, runs one ordinary substep with dt = unit_delta_t * limit and files the results back; the
block tables and the walk over the power-of-two levels are C++ host code, as in the reference.  This module is the Python
mirror of the class: it forwards.  No particle data crosses the host boundary during `step()`.

The methods of the base class that look at "the particles" — frame output, downloads, particle counts, energy, snapshots —
first make every container of every particle pool the ctx's records (`mpmhip_async_load_pools`), which is what
AsyncMPM<dim>::visualize lists (src/async/async_visualize.cpp:86-96): after a step() the records would otherwise hold
only the last working set.
"""
import ctypes as C

import numpy as np

from . import _lib
from .mpm import MPMError, Simulation3D
from .mpm2d import Simulation2D

_LP = C.POINTER(C.c_int64)


class AsyncSimulation3D(Simulation3D):
    """AsyncMPM<3>.  Config keys on top of MPM's: unit_delta_t (1e-6), max_units (8192), cfl_dt_mul, strength_dt_mul
    (src/async/async_mpm.cpp:24-27), left_boundary (:43-53)."""

    def initialize(self, config):
        cfg = dict(config)
        cfg["keep_apic_b"] = True  # the P2G matrix depends on each advance's dt: it is rebuilt from apic_b (include/mpmhip.h)
        super().initialize(cfg)
        self.unit_delta_t = float(cfg.get("unit_delta_t", 1e-6))
        self.max_units = int(cfg.get("max_units", 8192))
        self.cfl_dt_mul = float(cfg.get("cfl_dt_mul", 1.0))
        self.strength_dt_mul = float(cfg.get("strength_dt_mul", 1.0))
        self.left_boundary = bool(cfg.get("left_boundary", False))  # src/async/async_mpm.cpp:43-53
        self.nb = ((self.res[0] >> 2) + 1, (self.res[1] >> 2) + 1, (self.res[2] >> 3) + 1)
        self._begun = False
        return self

    # ------------------------------------------------------------------------------------------- plumbing
    def _create(self, capacity):
        super()._create(capacity)
        a = _lib.AsyncConfig(self.unit_delta_t, self.max_units, self.cfl_dt_mul, self.strength_dt_mul, int(self.left_boundary))
        self._check(self._L.mpmhip_async_begin(self._ctx, C.byref(a)))
        self._begun = True

    def _state(self):
        self._ensure_ctx()
        o = (C.c_int64 * 8)()
        self._check(self._L.mpmhip_async_state(self._ctx, o))
        return list(o)

    def _pool(self):
        """particles waiting in the ctx's records (just added) -> their blocks' pools"""
        self._ensure_ctx()
        self._check(self._L.mpmhip_async_pool_particles(self._ctx))

    def _load_pools(self):
        self._ensure_ctx()
        self._check(self._L.mpmhip_async_load_pools(self._ctx))

    # ------------------------------------------------------------------------------------------- the AsyncMPM interface
    def add_particles(self, config):
        """AsyncMPM<dim>::add_particles (src/async/async_mpm.cpp:57-75): the new particles go to their blocks' pools"""
        if dict(config).get("type") == "rigid":
            raise MPMError("rigid bodies cannot be combined with asynchronous stepping")
        self._ensure_ctx()
        ret = super().add_particles(config)
        self._pool()
        return ret

    def step(self, dt):
        """AsyncMPM<dim>::step (src/async/async_mpm.cpp:380-421)"""
        self._ensure_ctx()
        if dt < 0:
            raise MPMError("AsyncMPM.step(dt < 0) is the synchronous substep of the base class: use create_simulation3('mpm')")
        self._check(self._L.mpmhip_async_step(self._ctx, C.c_float(dt)))

    def substep(self):
        raise MPMError("AsyncMPM steps with step(dt); the synchronous substep belongs to create_simulation3('mpm')")

    run_substeps = None

    @property
    def current_t_int(self):
        return self._state()[0]

    @property
    def update_counter(self):
        return self._state()[1]

    @property
    def min_delta_t_int(self):
        return self._state()[2]

    @property
    def max_delta_t_int(self):
        return self._state()[3]

    def get_current_time(self):
        self._ensure_ctx()
        return float(self._L.mpmhip_async_current_time(self._ctx))

    def get_num_pool_particles(self):
        """containers in all particle pools (an id can sit in more than one, as in the reference)"""
        self._ensure_ctx()
        return int(self._check(self._L.mpmhip_async_download_pools(self._ctx, 0, None, None)))

    def block_table(self):
        """dense block table (block b = (bx nb[1] + by) nb[2] + bz): continuous / strength / cfl limits, pool sizes,
        particle_t, backup_t, local_min_dt_limit"""
        self._ensure_ctx()
        nb = (C.c_int32 * 3)()
        n = int(self._check(self._L.mpmhip_async_table(self._ctx, nb, 0, None, None, None, None)))
        arr = {k: np.zeros(n, np.int64) for k in ("strength", "cfl", "continuous", "count", "particle_t", "backup_t", "local_min")}
        p = {k: a.ctypes.data_as(_LP) for k, a in arr.items()}
        self._check(self._L.mpmhip_async_table(self._ctx, nb, n, p["strength"], p["cfl"], p["continuous"], p["count"]))
        self._check(self._L.mpmhip_async_block_times(self._ctx, n, p["particle_t"], p["backup_t"], p["local_min"]))
        arr["nb"] = tuple(nb)
        return arr

    def get_pool_particles(self):
        """every container of every particle pool (each at its block's particle_t) + its block's limits, sorted by id"""
        n = self.get_num_pool_particles()
        rows = np.zeros((max(n, 1), 27), np.float32)
        blk = np.zeros(max(n, 1), np.int32)
        got = int(self._check(self._L.mpmhip_async_download_pools(self._ctx, n, rows.ctypes.data_as(C.POINTER(C.c_float)),
                                                                 blk.ctypes.data_as(C.POINTER(C.c_int32)))))
        rows, blk = rows[:got], blk[:got]
        tab = self.block_table()
        out = dict(x=rows[:, 0:3].copy(), v=rows[:, 3:6].copy(), F=rows[:, 6:15].copy(), B=rows[:, 15:24].copy(), aux=rows[:, 24].copy(),
                   gid=rows[:, 25].copy().view(np.int32), id=rows[:, 26].copy().view(np.int32), block=blk.astype(np.int64),
                   continuous=tab["continuous"][blk], particle_t=tab["particle_t"][blk])
        o = np.lexsort((out["block"], out["id"]))
        return {k: v[o] for k, v in out.items()}

    def host_particle_bytes(self):
        self._ensure_ctx()
        return int(self._L.mpmhip_host_particle_bytes(self._ctx))

    # ------------------------------------------------------------------------------------------- views of the whole state
    def _whole_state(self):
        """AsyncMPM<dim>::visualize's particle list: all containers of all particle pools become the ctx's records"""
        self._load_pools()

    def get_num_particles(self):
        if self._ctx is None:
            return self._n_added
        return self.get_num_pool_particles()

    def get_particles(self, sort_by_id=True):
        self._whole_state()
        return super().get_particles(sort_by_id)

    def write_partio(self, file_name):
        self._whole_state()
        return super().write_partio(file_name)

    def bgeo_bytes(self, verbose=None):
        self._whole_state()
        return super().bgeo_bytes(verbose)

    def calculate_energy(self):
        self._whole_state()
        return super().calculate_energy()

    def save_snapshot(self, path):
        """every pool and backup container, the block table and the clocks (include/mpmhip.h: mpmhip_async_snapshot_save; the
        reference serialises the same: src/async/async_mpm.h:120-172)"""
        self._ensure_ctx()
        n = int(self._check(self._L.mpmhip_async_snapshot_size(self._ctx)))
        buf = np.empty(n, np.uint8)
        self._check(self._L.mpmhip_async_snapshot_save(self._ctx, buf.ctypes.data_as(C.c_void_p), n))
        with open(path, "wb") as f:
            f.write(np.array([self.frame], np.int64).tobytes())
            f.write(buf.tobytes())

    def load_snapshot(self, path):
        """into a simulation initialised with the same grid and unit_delta_t; replaces all particles and groups"""
        raw = np.fromfile(path, np.uint8)
        self.frame = int(raw[:8].view(np.int64)[0])
        blob = np.ascontiguousarray(raw[8:])
        n_groups = int(blob[12:16].view(np.uint32)[0])
        self._ensure_ctx()
        self._check(self._L.mpmhip_async_snapshot_load(self._ctx, blob.ctypes.data_as(C.c_void_p), len(blob)))
        off = 8 + 4 + 4 + 24 + 8 + 7 * 8 + 8 + 8  # sizeof(SnapAsync)
        rows = blob[off:off + 80 * n_groups].view(np.float32).reshape(n_groups, 20)
        self._groups = [(int(r[16:17].view(np.int32)[0]), r[:16].copy()) for r in rows]
        self._n_added = max(self._n_added, self.get_num_pool_particles())


class AsyncSimulation2D(Simulation2D):
    """AsyncMPM<2> (create_simulation2('async_mpm')).  Config keys on top of MPM<2>'s: unit_delta_t (1e-6), max_units (8192),
    cfl_dt_mul, strength_dt_mul (src/async/async_mpm.cpp:24-27), left_boundary (:43-53).  A scheduler block is 8 x 16 nodes."""

    def initialize(self, config):
        cfg = dict(config)
        super().initialize(cfg)
        self.unit_delta_t = float(cfg.get("unit_delta_t", 1e-6))
        self.max_units = int(cfg.get("max_units", 8192))
        self.cfl_dt_mul = float(cfg.get("cfl_dt_mul", 1.0))
        self.strength_dt_mul = float(cfg.get("strength_dt_mul", 1.0))
        self.left_boundary = bool(cfg.get("left_boundary", False))
        self.nb = ((self.res[0] >> 3) + 1, (self.res[1] >> 4) + 1)
        return self

    # ------------------------------------------------------------------------------------------- plumbing
    def _resident_particles(self):
        return 0  # (every batch moves to the pools right away: the arrays only ever hold one batch or one working set)

    def _ensure_ctx(self, extra=0):
        fresh = self._ctx is None
        super()._ensure_ctx(extra)
        if fresh:
            a = _lib.AsyncConfig(self.unit_delta_t, self.max_units, self.cfl_dt_mul, self.strength_dt_mul, int(self.left_boundary))
            self._check(self._L.mpmhip2d_async_begin(self._ctx, C.byref(a)))
            self._capacity = 1 << 62  # (the library grows the arrays of a resident stepper: they hold one batch or one working set)
            self._check(self._L.mpmhip2d_async_pool_particles(self._ctx))  # (particles staged before the ctx existed)

    def _state(self):
        self._ensure_ctx()
        o = (C.c_int64 * 8)()
        self._check(self._L.mpmhip2d_async_state(self._ctx, o))
        return list(o)

    # ------------------------------------------------------------------------------------------- the AsyncMPM interface
    def add_particles(self, config):
        """AsyncMPM<dim>::add_particles (src/async/async_mpm.cpp:57-75): the new particles go to their blocks' pools"""
        if dict(config).get("type") == "rigid":
            raise MPMError("rigid bodies cannot be combined with asynchronous stepping")
        self._ensure_ctx()
        ret = super().add_particles(config)
        self._check(self._L.mpmhip2d_async_pool_particles(self._ctx))
        return ret

    def add_rigid_body(self, cfg):
        raise MPMError("rigid bodies cannot be combined with asynchronous stepping")

    def step(self, dt):
        """AsyncMPM<dim>::step (src/async/async_mpm.cpp:380-421)"""
        self._ensure_ctx()
        if dt < 0:
            raise MPMError("AsyncMPM.step(dt < 0) is the synchronous substep of the base class: use create_simulation2('mpm')")
        self._check(self._L.mpmhip2d_async_step(self._ctx, C.c_float(dt)))

    def substep(self):
        raise MPMError("AsyncMPM steps with step(dt); the synchronous substep belongs to create_simulation2('mpm')")

    run_substeps = None

    @property
    def current_t_int(self):
        return self._state()[0]

    @property
    def update_counter(self):
        return self._state()[1]

    @property
    def min_delta_t_int(self):
        return self._state()[2]

    @property
    def max_delta_t_int(self):
        return self._state()[3]

    def get_current_time(self):
        self._ensure_ctx()
        return float(self._L.mpmhip2d_async_current_time(self._ctx))

    def get_name(self):
        return "async_mpm"

    def block_table(self):
        """dense block table (block b = bx nb[1] + by): continuous / strength / cfl limits, pool sizes, particle_t, backup_t,
        local_min_dt_limit"""
        self._ensure_ctx()
        nb = (C.c_int32 * 2)()
        n = int(self._check(self._L.mpmhip2d_async_table(self._ctx, nb, 0, *([None] * 7))))
        keys = ("strength", "cfl", "continuous", "count", "particle_t", "backup_t", "local_min")
        arr = {k: np.zeros(n, np.int64) for k in keys}
        self._check(self._L.mpmhip2d_async_table(self._ctx, nb, n, *(arr[k].ctypes.data_as(_LP) for k in keys)))
        arr["nb"] = tuple(nb)
        return arr

    def _load_pools(self):
        self._ensure_ctx()
        return int(self._check(self._L.mpmhip2d_async_load_pools(self._ctx)))

    def get_num_pool_particles(self):
        """containers in all particle pools (an id can sit in more than one, as in the reference)"""
        return self._load_pools()

    def get_num_particles(self):
        if self._ctx is None:
            return self._n_added
        return self._load_pools()

    def get_particles(self, sort_by_id=True):
        """AsyncMPM<dim>::visualize's particle list: every container of every particle pool, each at its block's time"""
        self._load_pools()
        return super().get_particles(sort_by_id)

    def get_pool_particles(self):
        """every container of every particle pool + its block and that block's limits, sorted by (id, block)"""
        n = self._load_pools()
        out = Simulation2D.get_particles(self, sort_by_id=False)
        blk = np.zeros(max(n, 1), np.int32)
        got = int(self._check(self._L.mpmhip2d_async_view_blocks(self._ctx, n, blk.ctypes.data_as(C.POINTER(C.c_int32)))))
        assert got == n == len(out["id"])
        blk = blk[:n].astype(np.int64)
        tab = self.block_table()
        out.update(block=blk, continuous=tab["continuous"][blk], particle_t=tab["particle_t"][blk])
        o = np.lexsort((out["block"], out["id"]))
        return {k: v[o] for k, v in out.items()}
