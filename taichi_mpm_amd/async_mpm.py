"""AsyncMPM — the reference's asynchronous (block-local time step) stepper, `create_simulation3('async_mpm')`
(class AsyncMPM<dim>, src/async/async_mpm.{h,cpp}).

The reference's AsyncMPM is host orchestration AROUND MPM<dim>::substep(): per scheduler block (SPGrid block of 4x4x8 nodes)
a pool of particles at the block's own time and a backup copy at an earlier time; `advance(limit)` gathers the blocks
that step with `limit` units together with frozen copies of their neighbours (pools of smaller-step neighbours, backups
of larger-step ones), runs ONE ordinary substep with dt = unit_delta_t * limit on that working set, and files the
results back into the pools (src/async/async_mpm.cpp:255-378).  This module keeps that structure: the bookkeeping
(pools, block times, neighbour lists) runs on the host exactly as the reference does it, every substep of a working
set and every per-block limit reduction runs on the device (mpmhip_substep, mpmhip_async_update_dt_limits).  Working
sets travel host <-> device per advance: correct first, not yet fast — the resident version (pools as device record
ranges, gathers as index kernels) is the follow-up.
"""
import ctypes as C

import numpy as np

from .mpm import F_ID, MPMError, Simulation3D

FIELDS = (("x", 3), ("v", 3), ("F", 9), ("B", 9), ("aux", 1), ("gid", 1), ("id", 1))


def _spread3(v):
    v = np.asarray(v, np.uint64) & np.uint64(0x3ff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x030000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x0300f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x030c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
    return v


def scheduler_offset(bx, by, bz):
    """the reference's block number: the page bits of SparseMask::Linear_Offset — per level z is the most significant
    bit, then x, then y (page_zmask / page_xmask / page_ymask, external/SPGrid/Core/SPGrid_Mask.h:29-35); the order the
    reference walks its pools in"""
    return (_spread3(bz) << np.uint64(2)) | (_spread3(bx) << np.uint64(1)) | _spread3(by)


class _Pool:
    """particle containers of one block: dict of arrays"""

    @staticmethod
    def empty():
        return {k: np.zeros((0, w) if w > 1 else (0,), np.int32 if k in ("gid", "id") else np.float32) for k, w in FIELDS}

    @staticmethod
    def take(p, idx):
        return {k: p[k][idx] for k, _ in FIELDS}

    @staticmethod
    def cat(parts):
        parts = [p for p in parts if len(p["id"])]
        if not parts:
            return _Pool.empty()
        return {k: np.concatenate([p[k] for p in parts]) for k, _ in FIELDS}


class AsyncSimulation3D(Simulation3D):
    """AsyncMPM<3>.  Config keys on top of MPM's: unit_delta_t (1e-6), max_units (8192), cfl_dt_mul, strength_dt_mul
    (src/async/async_mpm.cpp:24-27)."""

    def initialize(self, config):
        cfg = dict(config)
        cfg["keep_apic_b"] = True  # working sets are re-uploaded with their apic_b every advance
        super().initialize(cfg)
        self.unit_delta_t = float(cfg.get("unit_delta_t", 1e-6))
        self.max_units = int(cfg.get("max_units", 8192))
        self.cfl_dt_mul = float(cfg.get("cfl_dt_mul", 1.0))
        self.strength_dt_mul = float(cfg.get("strength_dt_mul", 1.0))
        self.nb = ((self.res[0] >> 2) + 1, (self.res[1] >> 2) + 1, (self.res[2] >> 3) + 1)
        nblk = self.nb[0] * self.nb[1] * self.nb[2]
        bx, by, bz = np.meshgrid(np.arange(self.nb[0]), np.arange(self.nb[1]), np.arange(self.nb[2]), indexing="ij")
        self._offset = scheduler_offset(bx, by, bz).reshape(-1)  # block b (dense index) -> the reference's offset
        self._order = np.argsort(self._offset, kind="stable")     # dense indices in ascending offset order
        self._coords = np.stack([bx, by, bz], -1).reshape(-1, 3)
        self.particle_t = np.zeros(nblk, np.int64)
        self.backup_t = np.zeros(nblk, np.int64)
        self.local_min = np.ones(nblk, np.int64)
        self.continuous = np.ones(nblk, np.int64)
        self.pool = {}    # dense block index -> containers at particle_t
        self.backup = {}  # ... at backup_t
        self.current_t_int = 0
        self.request_t = 0.0
        self.current_t = 0.0
        self.update_counter = 0
        self._async_on = False
        self._next_id = 0
        # neighbour table (cached_neighbours, src/async/async_mpm.h:248-301): the 26 blocks around a block
        nb = self.nb
        self._neigh = []
        for b in range(nblk):
            c = self._coords[b]
            lst = []
            for i in (-1, 0, 1):
                for j in (-1, 0, 1):
                    for k in (-1, 0, 1):
                        q = c + (i, j, k)
                        if (i or j or k) and (q >= 0).all() and (q < nb).all():
                            lst.append(int((q[0] * nb[1] + q[1]) * nb[2] + q[2]))
            self._neigh.append(np.asarray(lst, np.int64))
        return self

    # ------------------------------------------------------------------------------------------- pools
    def _block_of(self, x):
        b = (np.asarray(x, np.float32) * np.float32(1.0 / self.delta_x) - np.float32(0.5)).astype(np.int32)
        return ((b[:, 0] >> 2) * self.nb[1] + (b[:, 1] >> 2)) * self.nb[2] + (b[:, 2] >> 3)

    def add_particles(self, config):
        """AsyncMPM<dim>::add_particles (src/async/async_mpm.cpp:57-75): the new particles go to their blocks' pools"""
        self._ensure_ctx()
        self._check(self._L.mpmhip_clear_particles(self._ctx))
        super().add_particles(config)  # registers the group, applies the near-boundary filter, uploads
        st = self.get_particles(sort_by_id=True)
        part = {k: st[k] for k, _ in FIELDS}
        part["id"] = (part["id"] + self._next_id).astype(np.int32)  # creation ids keep counting (particle_counter)
        self._next_id += len(part["id"])
        blk = self._block_of(part["x"])
        for b in np.unique(blk):
            self.pool[int(b)] = _Pool.cat([self.pool.get(int(b), _Pool.empty()), _Pool.take(part, blk == b)])
        self._check(self._L.mpmhip_clear_particles(self._ctx))
        return ""

    def _load(self, part):
        """working set -> device (slot order = the order given)"""
        self._check(self._L.mpmhip_clear_particles(self._ctx))
        n = len(part["id"])
        if not n:
            return
        order = np.argsort(part["gid"], kind="stable")
        part = _Pool.take(part, order)
        for gi in np.unique(part["gid"]):
            m = part["gid"] == gi
            self._upload_new(int(gi), part["x"][m], part["v"][m], part["F"][m], part["B"][m], part["aux"][m])
        ids = np.ascontiguousarray(part["id"], np.int32)
        self._check(self._L.mpmhip_upload(self._ctx, F_ID, ids.ctypes.data_as(C.c_void_p), len(ids)))

    def _all_pool_particles(self):
        return _Pool.cat([self.pool[b] for b in self._order if b in self.pool])

    # ------------------------------------------------------------------------------------------- limits
    def update_dt_limits(self):
        """AsyncMPM<dim>::update_dt_limits (src/async/async_mpm.cpp:90-253)"""
        from . import _lib
        if not self._async_on:
            a = _lib.AsyncConfig(self.unit_delta_t, self.max_units, self.cfl_dt_mul, self.strength_dt_mul)
            self._check(self._L.mpmhip_async_enable(self._ctx, C.byref(a)))
            self._async_on = True
        self._load(self._all_pool_particles())  # the per-block reductions run on the device over every pool particle
        self._check(self._L.mpmhip_async_set_time_int(self._ctx, self.current_t_int))
        self._check(self._L.mpmhip_async_update_dt_limits(self._ctx))
        nblk = len(self.continuous)
        nb = (C.c_int32 * 3)()
        cont, cnt = np.zeros(nblk, np.int64), np.zeros(nblk, np.int64)
        lp = C.POINTER(C.c_int64)
        got = self._check(self._L.mpmhip_async_table(self._ctx, nb, nblk, None, None, cont.ctypes.data_as(lp), cnt.ctypes.data_as(lp)))
        assert got == nblk and tuple(nb) == self.nb
        self.continuous = cont
        nonempty = cnt > 0
        self.min_delta_t_int = int(cont.min())  # update_dt_limit_boundary(false), :153
        self.max_delta_t_int = int(cont.max())
        # local_min_dt_limit (:165-182)
        for b in range(nblk):
            if cont[b] == self.min_delta_t_int:
                continue
            nbs = self._neigh[b]
            self.local_min[b] = int((self.particle_t[nbs] + cont[nbs]).min()) if len(nbs) else (1 << 31)
        # larger / smaller neighbour lists per log2(limit) (:183-247), each sorted by the reference's block number
        self._larger, self._smaller = {}, {}
        for b in range(nblk):
            for q in self._neigh[b]:
                if cont[b] < cont[q]:
                    self._larger.setdefault(int(cont[b]), set()).add(int(q))
                    self._smaller.setdefault(int(cont[q]), set()).add(b)
        key = lambda b: int(self._offset[b])  # noqa: E731
        self._larger = {k: sorted(v, key=key) for k, v in self._larger.items()}
        self._smaller = {k: sorted(v, key=key) for k, v in self._smaller.items()}
        del nonempty

    # ------------------------------------------------------------------------------------------- advance
    def advance(self, limit):
        """AsyncMPM<dim>::advance (src/async/async_mpm.cpp:255-378)"""
        t = self.current_t_int
        cont = self.continuous
        nblk = len(cont)
        has_copied = np.zeros(nblk, bool)
        seen, parts = set(), []

        def gather(p):  # gather_from_pool: the first copy of an id wins (particles_cnt[id] != global_cnt), :191-218
            if p is None or not len(p["id"]):
                return
            keep = np.array([int(i) not in seen for i in p["id"]])
            # duplicates inside one pool: keep the first
            ids = p["id"]
            first = np.zeros(len(ids), bool)
            for j, i in enumerate(ids):
                if keep[j] and int(i) not in seen:
                    seen.add(int(i))
                    first[j] = True
            if first.any():
                parts.append(_Pool.take(p, first))
        for b in self._smaller.get(limit, []):  # pools of smaller-step neighbours (at the current time)
            assert self.particle_t[b] == t, "particle_pool broken 2"
            has_copied[b] = True
            gather(self.pool.get(b))
        for b in self._order:  # blocks that step with `limit`
            if cont[b] == limit:
                assert self.particle_t[b] == t, "particle_pool broken 1"
                gather(self.pool.get(int(b)))
        for b in self._larger.get(limit, []):  # backups of larger-step neighbours
            assert self.backup_t[b] == t, "backup_pool broken"
            has_copied[b] = True
            gather(self.backup.get(b))
        for b in range(nblk):  # backup_current_dt_limit (:317-325)
            if cont[b] == limit and self.particle_t[b] == t:
                self.backup[b] = self.pool.pop(b, _Pool.empty())
                self.backup_t[b] = t
        work = _Pool.cat(parts)
        self.update_counter += len(work["id"])
        # ONE ordinary substep of the working set with this level's dt
        self._load(work)
        self._check(self._L.mpmhip_set_dt(self._ctx, np.float32(self.unit_delta_t) * np.float32(limit)))
        self._check(self._L.mpmhip_set_time(self._ctx, float(np.float32(self.unit_delta_t) * np.float32(t))))
        if len(work["id"]):
            self._check(self._L.mpmhip_substep(self._ctx))
            out = self.get_particles(sort_by_id=False)
        else:
            out = _Pool.empty()
        for b in range(nblk):  # update backup_t and particle_t (:331-343)
            if cont[b] == limit:
                self.particle_t[b] = t + limit
            elif has_copied[b] and cont[b] > limit and self.local_min[b] == t + limit:
                self.backup[b] = _Pool.empty()
                self.backup_t[b] = t + limit
        if len(out["id"]):  # file the results back (:345-375)
            blk = self._block_of(out["x"])
            for b in np.unique(blk):
                b = int(b)
                sel = _Pool.take({k: out[k] for k, _ in FIELDS}, blk == b)
                if cont[b] == limit:
                    self.pool[b] = _Pool.cat([self.pool.get(b, _Pool.empty()), sel])
                elif has_copied[b] and cont[b] > limit and self.local_min[b] == t + limit:
                    self.backup[b] = _Pool.cat([self.backup.get(b, _Pool.empty()), sel])

    # ------------------------------------------------------------------------------------------- step
    def step(self, dt):
        """AsyncMPM<dim>::step (src/async/async_mpm.cpp:380-421)"""
        self._ensure_ctx()
        if dt < 0:
            raise MPMError("AsyncMPM.step(dt < 0) is the synchronous substep of the base class: use create_simulation3('mpm')")
        self.request_t = float(np.float32(self.request_t) + np.float32(dt))
        while True:
            self.update_dt_limits()
            d = self.max_delta_t_int
            while d >= self.min_delta_t_int:
                if self.current_t_int % d == 0:
                    self.advance(d)
                d >>= 1
            self.current_t_int += self.min_delta_t_int - self.current_t_int % self.min_delta_t_int
            self.current_t = float(np.float32(self.unit_delta_t) * np.float32(self.current_t_int))
            if not self.current_t < self.request_t:
                break

    def get_current_time(self):
        return self.current_t

    def get_num_pool_particles(self):
        return sum(len(p["id"]) for p in self.pool.values())

    def get_pool_particles(self):
        """every container of every pool (each at its block's particle_t) + its block's limits, sorted by id"""
        blocks = [b for b in self._order if b in self.pool and len(self.pool[b]["id"])]
        out = _Pool.cat([self.pool[b] for b in blocks])
        out["continuous"] = np.concatenate([np.full(len(self.pool[b]["id"]), self.continuous[b]) for b in blocks]) if blocks else np.zeros(0, np.int64)
        out["particle_t"] = np.concatenate([np.full(len(self.pool[b]["id"]), self.particle_t[b]) for b in blocks]) if blocks else np.zeros(0, np.int64)
        o = np.argsort(out["id"], kind="stable")
        return {k: v[o] for k, v in out.items()}
