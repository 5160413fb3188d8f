"""The grid tiled across the GPUs of one node: one process per GPU, each with its own libmpmhip ctx holding the
particles of one brick.  The reference has no multi-process code; the decomposition
follows from its stencil support (base..base+2 nodes, src/kernel.h:119-121, src/transfer.cpp:59-63).

Per substep:   begin (sort, P2G, pack partial sums on the halo boxes)
               ONE all_to_all of the halo boxes (RCCL over xGMI; box R∩S is the same on both sides)
               end (grid: contributors summed in rank order => bit-identical on every holder; G2P)
Every `migrate_interval` substeps: particles whose base cell left the brick move to their new owner.

The module is engine-agnostic: `HipEngine` drives a libmpmhip ctx; the CPU tests plug a checker engine in
(tests/fake_engine.py) to run the exchange/migration logic under gloo with world_size 2.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

MIGRATE_FLOATS = _lib.MIGRATE_FLOATS


def brick_dims(world):
    """2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2; otherwise the most cubic factorisation."""
    dims = [1, 1, 1]
    n, a = world, 0
    f = 2
    factors = []
    while n > 1:
        while n % f == 0:
            factors.append(f)
            n //= f
        f += 1
    for f in sorted(factors, reverse=True):
        a = int(np.argmin(dims))
        dims[a] *= f
    return tuple(sorted(dims, reverse=True))


def base_cells(x, dx):
    """base cell of the quadratic stencil, int(x/dx - 0.5) (src/kernel.h:119-121), in fp32 like the device"""
    X = np.asarray(x, np.float32) * np.float32(1.0 / dx)
    return (X - np.float32(0.5)).astype(np.int32)


def balanced_cuts(cells, res, parts, hist=None):
    """cell cut planes [0, c1, .., res] along one axis so that every part holds about the same number of
    particles (marginal histogram of base cells; `hist` = that histogram when it was accumulated by the caller)."""
    cuts = [0]
    if parts > 1:
        if hist is None:
            hist = np.bincount(np.clip(cells, 0, res - 1), minlength=res)
        cum = np.cumsum(np.asarray(hist, np.int64))
        total = int(cum[-1])
        for k in range(1, parts):
            c = int(np.searchsorted(cum, total * k / parts, side="left")) + 1 if total else (res * k) // parts
            if total and c < res and hist[c] == 0:
                # the balance point lies at the edge of an EMPTY stretch (clusters that do not touch): cut in its middle, not along
                # the face of the lower cluster — there every particle that moves up a cell would cross into the next brick
                nz = np.nonzero(np.asarray(hist[c:]) > 0)[0]
                if len(nz):
                    c = c + int(nz[0]) // 2
            c = max(c, cuts[-1] + 1)
            c = min(c, res - (parts - k))
            cuts.append(c)
    cuts.append(int(res))
    return cuts


class Partition:
    """`dims` bricks with cell cut planes `cuts[a]`; rank = (px*dims[1] + py)*dims[2] + pz."""

    def __init__(self, res, dims, cuts, margin):
        self.res = tuple(int(r) for r in res)
        self.dims = tuple(int(d) for d in dims)
        self.cuts = [list(map(int, c)) for c in cuts]
        self.margin = int(margin)
        self.world = self.dims[0] * self.dims[1] * self.dims[2]
        # halo boxes are clipped to this node box (the occupied part of the grid + slack, the same on all ranks)
        self.clip = ([0, 0, 0], [r + 1 for r in self.res])
        for a in range(3):
            assert len(self.cuts[a]) == self.dims[a] + 1 and self.cuts[a][0] == 0 and self.cuts[a][-1] >= self.res[a]
            assert all(self.cuts[a][k] < self.cuts[a][k + 1] for k in range(self.dims[a]))
            assert self.dims[a] <= _lib.MAX_PARTS

    @classmethod
    def balanced(cls, res, world, x, dx, margin, dims=None, clip=True):
        dims = dims or brick_dims(world)
        b = base_cells(x, dx)
        part = cls(res, dims, [balanced_cuts(b[:, a], res[a], dims[a]) for a in range(3)], margin)
        if clip and len(b):
            part.set_clip_from_bounds(b.min(0), b.max(0) + 1)
        return part

    @classmethod
    def from_histograms(cls, res, world, hists, cell_lo, cell_hi, margin, dims=None):
        """as `balanced`, from the per-axis histograms of the base cells and their bounds (a scene made of several
        clusters is histogrammed cluster by cluster, without ever holding all positions at once)"""
        dims = dims or brick_dims(world)
        part = cls(res, dims, [balanced_cuts(None, res[a], dims[a], hist=hists[a]) for a in range(3)], margin)
        part.set_clip_from_bounds(cell_lo, cell_hi)
        return part

    def coords(self, rank):
        return (rank // (self.dims[1] * self.dims[2]), (rank // self.dims[2]) % self.dims[1], rank % self.dims[2])

    def brick(self, rank):
        pc = self.coords(rank)
        return ([self.cuts[a][pc[a]] for a in range(3)], [self.cuts[a][pc[a] + 1] for a in range(3)])

    def node_box(self, rank):
        """nodes the rank's particles can touch: base cells in [lo-margin, hi+margin), stencil base..base+2"""
        lo, hi = self.brick(rank)
        return ([max(self.clip[0][a], lo[a] - self.margin) for a in range(3)],
                [min(self.clip[1][a], hi[a] + self.margin + 2) for a in range(3)])

    def clip_slack(self):
        """slack of a freshly cut clip box: the room clip_covers asks for and as much again (csrc/tiled_api.h: tn_clip_slack)"""
        return max(8, 4 * self.margin + 4)

    def clip_covers(self, cell_lo, cell_hi):
        """does the current clip box hold every node the particles (base cells in [cell_lo, cell_hi)) can touch
        before the next check, with room for a travel of 2 margin cells (the schedule plans for margin)?"""
        room = 2 * self.margin
        return all(cell_lo[a] - room - 1 >= self.clip[0][a] or self.clip[0][a] == 0 for a in range(3)) and \
            all(cell_hi[a] + room + 3 <= self.clip[1][a] or self.clip[1][a] == self.res[a] + 1 for a in range(3))

    def clip_holds(self, cell_lo, cell_hi):
        """did the clip box hold every node the particles touch NOW (base cell b: nodes b .. b + 2)?  If not, halo sums of the
        substeps since the last check may have missed mass outside the clipped boxes (csrc/tiled_api.h: tn_mig_c)"""
        # (a side where the clip box ends at the grid itself holds everything there is: with clean_boundary off, or on the clamped
        # generic path, base cells reach res - 1 and the stencil's last node res + 1 does not exist)
        return all(cell_lo[a] >= self.clip[0][a] and (cell_hi[a] + 2 <= self.clip[1][a] or self.clip[1][a] == self.res[a] + 1)
                   for a in range(3))

    def set_clip_from_bounds(self, cell_lo, cell_hi):
        s = self.clip_slack()
        self.clip = ([max(0, int(cell_lo[a]) - s) for a in range(3)],
                     [min(self.res[a] + 1, int(cell_hi[a]) + s + 2) for a in range(3)])

    def overlap(self, r, s):
        (alo, ahi), (blo, bhi) = self.node_box(r), self.node_box(s)
        lo = [max(alo[a], blo[a]) for a in range(3)]
        hi = [min(ahi[a], bhi[a]) for a in range(3)]
        return (lo, hi) if all(lo[a] < hi[a] for a in range(3)) else None

    def boxes(self, rank):
        """[(peer, lo, hi)] sorted by peer: the halo boxes of `rank`"""
        out = []
        for s in range(self.world):
            if s != rank:
                o = self.overlap(rank, s)
                if o:
                    out.append((s, o[0], o[1]))
        return out

    def rank_of_cells(self, b):
        idx = [np.searchsorted(np.asarray(self.cuts[a][1:-1]), b[:, a], side="right") for a in range(3)]
        return (idx[0] * self.dims[1] + idx[1]) * self.dims[2] + idx[2]


class HaloPlan:
    """send/recv buffers of one rank, laid out by peer rank so ONE all_to_all moves every box."""

    def __init__(self, part, rank, alloc):
        self.boxes = part.boxes(rank)
        if len(self.boxes) > _lib.MAX_HALO_BOXES:
            raise ValueError("too many halo boxes (%d)" % len(self.boxes))
        self.vol = [int(np.prod([hi[a] - lo[a] for a in range(3)])) for _, lo, hi in self.boxes]
        self.splits = [0] * part.world  # floats per peer; symmetric (box R∩S is the same on both sides)
        self.offsets = []
        off = 0
        for (peer, _, _), v in zip(self.boxes, self.vol):
            self.offsets.append(off)
            self.splits[peer] = 4 * v
            off += 4 * v
        self.total = off
        self.send = alloc(max(off, 4))
        self.recv = alloc(max(off, 4))
        self.send.zero_()
        self.recv.zero_()


# ---------------------------------------------------------------------------------------------------- engines
class HipEngine:
    """drives one libmpmhip ctx (the product path)"""

    def __init__(self, sim, device):
        import torch
        self.torch = torch
        self.sim = sim
        self.device = torch.device("cuda", device)
        sim._ensure_ctx()
        self.L, self.ctx = sim._L, sim._ctx
        # everything (kernels, RCCL collectives, buffer copies) on ONE stream: ordering by construction.  The legacy
        # default stream has handle 0, which the C ABI reads as "the ctx's own stream" — and a non-blocking stream
        # does not order against it — so make a real stream torch's current one first.
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == 0:
            cur = torch.cuda.Stream(self.device)
            torch.cuda.set_stream(cur)
        self.stream = cur
        sim.set_stream(cur.cuda_stream)

    def alloc(self, nfloats):
        return self.torch.empty(int(nfloats), dtype=self.torch.float32, device=self.device)

    def configure(self, part, rank, plan):
        ip = C.POINTER(C.c_int32)
        arrs = [np.asarray(c, np.int32) for c in part.cuts]
        dims = np.asarray(part.dims, np.int32)
        self.sim._check(self.L.mpmhip_set_partition(self.ctx, rank, dims.ctypes.data_as(ip), arrs[0].ctypes.data_as(ip),
                                                    arrs[1].ctypes.data_as(ip), arrs[2].ctypes.data_as(ip), part.margin))
        n = len(plan.boxes)
        hb = (_lib.HaloBox * max(n, 1))()
        for i, ((peer, lo, hi), off) in enumerate(zip(plan.boxes, plan.offsets)):
            hb[i].lo[:] = lo
            hb[i].hi[:] = hi
            hb[i].peer = peer
            hb[i].send = plan.send.data_ptr() + 4 * off
            hb[i].recv = plan.recv.data_ptr() + 4 * off
        self.sim._check(self.L.mpmhip_set_halo(self.ctx, n, hb))
        self.world = part.world

    def begin(self):
        self.sim._check(self.L.mpmhip_substep_begin(self.ctx))

    def interior(self):
        self.sim._check(self.L.mpmhip_substep_interior(self.ctx))

    def end(self):
        self.sim._check(self.L.mpmhip_substep_end(self.ctx))

    def set_overlap(self, on):
        """split substeps into boundary / interior work so that the halo exchange overlaps the interior part"""
        self.sim._check(self.L.mpmhip_set_overlap(self.ctx, int(bool(on))))

    def run_native(self, n, start, wait):
        """n substeps with the loop in the library (mpmhip_tiled_run): begin / interior / end never return to Python, only
        the transport does — start() launches the exchange of the halo boxes, wait() makes the ctx's stream wait for it.
        Exceptions of the callbacks are re-raised here."""
        err = []

        def cb(_user, phase):
            try:
                (start if phase == 0 else wait)()
                return 0
            except BaseException as e:  # noqa: BLE001 — must not unwind through the C frames
                err.append(e)
                return 1
        fn = _lib.EXCHANGE_FN(cb)
        rc = self.L.mpmhip_tiled_run(self.ctx, int(n), 0, fn, None)
        if err:
            raise err[0]
        return self.sim._check(rc)

    def migration_scan(self):
        """(leavers per destination rank, base-cell bounds lo, hi, fastest particle in cells per substep) — one pass,
        one synchronisation"""
        out, lo, hi = np.zeros(self.world, np.int64), np.zeros(3, np.int32), np.zeros(3, np.int32)
        ip, speed = C.POINTER(C.c_int32), C.c_float()
        self.sim._check(self.L.mpmhip_migration_scan(self.ctx, self.world, out.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     lo.ctypes.data_as(ip), hi.ctypes.data_as(ip), C.byref(speed)))
        return out, lo, hi, float(speed.value)

    def export_leavers(self, counts, buf):
        counts = np.ascontiguousarray(counts, np.int64)
        self.sim._check(self.L.mpmhip_export_leavers(self.ctx, self.world, counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     C.c_void_p(buf.data_ptr())))

    def import_particles(self, buf, n):
        if n:
            self.sim._check(self.L.mpmhip_import_particles(self.ctx, int(n), C.c_void_p(buf.data_ptr())))
        cap, slots = self.sim._capacity, int(self.L.mpmhip_num_slots(self.ctx))
        if slots > 0.85 * cap:  # dead slots (leavers) pile up: compact at the next sort
            self.sim._check(self.L.mpmhip_request_compaction(self.ctx))

    def num_particles(self):
        return self.sim.get_num_particles()

    def synchronize(self):
        self.sim.synchronize()


# ---------------------------------------------------------------------------------------------------- comms
class DistComm:
    """torch.distributed: backend nccl (= RCCL over xGMI) with device tensors, gloo with CPU tensors in tests"""

    def __init__(self, dist, device, group=None):
        import torch
        self.torch, self.dist, self.device, self.group = torch, dist, device, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_to_all(self, out, inp, out_splits, in_splits):
        out_splits, in_splits = list(map(int, out_splits)), list(map(int, in_splits))
        self.dist.all_to_all_single(out[:sum(out_splits)], inp[:sum(in_splits)], out_splits, in_splits,
                                    group=self.group)

    def all_to_all_async(self, out, inp, out_splits, in_splits):
        """start the exchange behind everything enqueued so far and return a handle; handle.wait() makes the
        CURRENT stream wait for it (no host block with nccl), so kernels launched in between overlap the wire"""
        out_splits, in_splits = list(map(int, out_splits)), list(map(int, in_splits))
        if self.device.type != "cuda":
            return self.dist.all_to_all_single(out[:sum(out_splits)], inp[:sum(in_splits)], out_splits, in_splits,
                                               group=self.group, async_op=True)
        tc = self.torch.cuda
        if getattr(self, "_side", None) is None:
            self._side, self._ev = tc.Stream(self.device), tc.Event()
        self._ev.record(tc.current_stream(self.device))
        with tc.stream(self._side):
            self._side.wait_event(self._ev)
            return self.dist.all_to_all_single(out[:sum(out_splits)], inp[:sum(in_splits)], out_splits, in_splits,
                                               group=self.group, async_op=True)

    def all_gather_ints(self, row):
        """every rank's int64 row -> (world, len(row)) on every rank: ONE small collective per migration"""
        t = self.torch.as_tensor(np.asarray(row, np.int64)).to(self.device)
        o = self.torch.empty(self.world * t.numel(), dtype=t.dtype, device=self.device)
        self.dist.all_gather_into_tensor(o, t, group=self.group)
        return o.cpu().numpy().reshape(self.world, -1)


class StagedDistComm(DistComm):
    """device buffers moved by a CPU-only backend (gloo): device -> host, collective, host -> device.  Lets several
    ranks share ONE GPU (RCCL refuses that), which is how the multi-process path is tested on a single-GPU box."""

    def __init__(self, dist, group=None):
        import torch
        super().__init__(dist, torch.device("cpu"), group)

    def all_to_all(self, out, inp, out_splits, in_splits):
        out_splits, in_splits = list(map(int, out_splits)), list(map(int, in_splits))
        h_in = inp[:sum(in_splits)].cpu()  # synchronises the stream the ctx runs on
        h_out = self.torch.empty(sum(out_splits), dtype=inp.dtype)
        self.dist.all_to_all_single(h_out, h_in, out_splits, in_splits, group=self.group)
        out[:sum(out_splits)].copy_(h_out)

    def all_to_all_async(self, out, inp, out_splits, in_splits):
        self.all_to_all(out, inp, out_splits, in_splits)  # staged transport: nothing to overlap with

        class _Done:
            def wait(self):
                return True
        return _Done()


# ---------------------------------------------------------------------------------------------------- one rank
class TiledRank:
    """one rank's substep / migration, split into phases so that a distributed job (one rank per process) and a
    virtual job (several ranks in one process, tests) run the very same code between the exchanges"""

    def __init__(self, engine, part, rank, migrate_interval=None):
        self.e, self.part, self.rank = engine, part, rank
        self.plan = HaloPlan(part, rank, engine.alloc)
        engine.configure(part, rank, self.plan)
        # CFL: a particle moves < 1 cell per substep, so it stays inside the margin for `margin` substeps.  That is
        # the worst case; the scan also measures the fastest particle, and when it is slow the next migration is
        # scheduled later (see schedule()).  An explicit migrate_interval fixes the schedule (tests).
        self.migrate_interval = int(migrate_interval or part.margin)
        assert 1 <= self.migrate_interval <= part.margin
        self.adaptive_cap = 0 if migrate_interval else int(os.environ.get("MPMHIP_TILE_MIGRATE_CAP", 64))
        self.k = 0
        self.next_migration = self.migrate_interval
        self.migrated_out = 0

    def schedule(self, speed):
        """next migration, from the globally fastest particle (cells per substep, identical on every rank): after a
        migration every particle is inside its brick and needs margin / speed substeps to cross the margin; half of
        that leaves room for the speed to double in between (gravity adds g dt^2 / dx ~ 1e-5 cells per substep^2),
        never sooner than the CFL schedule and never later than the cap"""
        n = self.migrate_interval
        if self.adaptive_cap > n and np.isfinite(speed):
            n = max(n, min(self.adaptive_cap, int(0.5 * self.part.margin / max(speed, 1e-9))))
        self.next_migration = self.k + n

    # --- migration phases
    def mig_scan(self):
        """this rank's row of the migration table: [leavers per destination | -lo | hi | speed] of its particles
        (speed = fastest particle in 2^-20 cells per substep, rounded up)"""
        self._counts, lo, hi, speed = self.e.migration_scan()
        if not (lo <= hi).all():  # no particles here
            lo, hi = np.full(3, 1 << 30), np.full(3, -(1 << 30))
        q = int(np.ceil(min(max(speed, 0.0), 1e6) * (1 << 20))) if np.isfinite(speed) else (1 << 40)
        return np.concatenate([self._counts, -lo.astype(np.int64), hi.astype(np.int64), [q]])

    def mig_export(self, incoming):
        self._incoming = np.asarray(incoming, np.int64)
        ns, nr = int(self._counts.sum()), int(self._incoming.sum())
        self._sendrec = self.e.alloc(max(ns, 1) * MIGRATE_FLOATS)
        self._recvrec = self.e.alloc(max(nr, 1) * MIGRATE_FLOATS)
        self.e.export_leavers(self._counts, self._sendrec)
        self.migrated_out += ns
        return self._sendrec, self._recvrec

    def mig_import(self):
        self.e.import_particles(self._recvrec, int(self._incoming.sum()))
        self._sendrec = self._recvrec = None

    def replan(self):
        """the partition's clip box changed: new halo boxes and buffers"""
        self.plan = HaloPlan(self.part, self.rank, self.e.alloc)
        self.e.configure(self.part, self.rank, self.plan)
        self.replans = getattr(self, "replans", 0) + 1


def _finish_migration(table, world, ranks, exchange):
    """second half of a migration, identical on every rank because it only looks at the all-gathered table
    [counts r -> s | -lo | hi | speed]: move the records (skipped when nobody moves), keep the halo boxes wrapped around
    the particles.  `ranks` are the TiledRank objects of this process (one, or all of them for a virtual job)."""
    counts = table[:, :world]  # counts[r][s]: r -> s
    if counts.sum() > 0:
        bufs = [r.mig_export(counts[:, r.rank]) for r in ranks]
        exchange([b[1] for b in bufs], [b[0] for b in bufs], counts)
        for r in ranks:
            r.mig_import()
    lo, hi = -table[:, world:world + 3].max(0), table[:, world + 3:world + 6].max(0)
    part = ranks[0].part
    if (lo <= hi).all() and not part.clip_holds(lo, hi):
        raise RuntimeError("tiled run: particles left the clipped halo region between two migrations (base cells %s .. %s, halo boxes "
                           "cut to nodes %s .. %s): their top speed more than doubled since the last check; lower the migration cap "
                           "(MPMHIP_TILE_MIGRATE_CAP) or pass migrate_interval" % (lo.tolist(), hi.tolist(), part.clip[0], part.clip[1]))
    if (lo <= hi).all() and not part.clip_covers(lo, hi):
        part.set_clip_from_bounds(lo, hi)  # (a virtual job shares ONE Partition object between its ranks)
        for r in ranks:
            if r.part is not part:
                r.part.set_clip_from_bounds(lo, hi)
            r.replan()
    speed = float(table[:, world + 6].max()) / (1 << 20)
    for r in ranks:
        r.schedule(speed)


class TiledJob:
    """bench.py job: this process's rank of the tiled run (torch.distributed)"""
    scaling = "strong"

    def __init__(self, engine, part, comm, migrate_interval=None, overlap=True):
        self.r = TiledRank(engine, part, comm.rank, migrate_interval)
        self.comm, self.e = comm, engine
        self.overlap = bool(overlap) and hasattr(engine, "set_overlap")
        if self.overlap:
            engine.set_overlap(True)
        self.parallelism = "%dx%dx%d bricks, one rank per GPU, halo all-sum + migration over RCCL" % part.dims

    def set_overlap(self, flag):
        """boundary / interior split of the substep (the exchange overlaps the interior part) on or off; between substeps"""
        self.overlap = bool(flag) and hasattr(self.e, "set_overlap")
        if hasattr(self.e, "set_overlap"):
            self.e.set_overlap(self.overlap)

    def substep(self):
        r, p = self.r, self.r.plan
        r.e.begin()  # sort, P2G (only the blocks touching a halo box when overlapping), halo pack
        if p.total and self.overlap:
            try:
                work = self.comm.all_to_all_async(p.recv, p.send, p.splits, p.splits)
            except Exception as exc:  # a transport without async collectives: fall back to the serial exchange
                import sys
                print("tiled: async exchange unavailable (%r); continuing without overlap" % (exc,), file=sys.stderr)
                self.overlap, self._unsplit_after = False, True  # the substep in flight stays split:
                self.comm.all_to_all(p.recv, p.send, p.splits, p.splits)  # substep_end runs its interior part
                work = None
            if work is not None:
                r.e.interior()  # everything that cannot touch a halo node runs while the boxes are on the wire
                work.wait()
        elif p.total:
            self.comm.all_to_all(p.recv, p.send, p.splits, p.splits)
        r.e.end()
        if getattr(self, "_unsplit_after", False):
            self._unsplit_after = False
            self.e.set_overlap(False)
        r.k += 1
        if r.k >= r.next_migration:
            self.migrate()

    def migrate(self):
        r, w = self.r, self.comm.world
        table = self.comm.all_gather_ints(r.mig_scan())  # one pass over the particles, one small collective
        _finish_migration(table, w, [r], lambda recvs, sends, cnt: self.comm.all_to_all(
            recvs[0], sends[0], cnt[:, r.rank] * MIGRATE_FLOATS, cnt[r.rank] * MIGRATE_FLOATS))

    substeps = 0  # substeps run so far (bench.py reports it)

    def run(self, n):
        """n substeps.  With the HIP engine the loop between two migrations runs inside the library (mpmhip_tiled_run) and only
        the exchange calls back; other engines (the CPU checker of the gloo tests) take the Python loop."""
        left = n
        while left > 0:
            r, p = self.r, self.r.plan
            m = min(left, max(int(r.next_migration - r.k), 1))
            if hasattr(self.e, "run_native") and not getattr(self, "_python_loop", False):
                hold = {}

                def start():
                    if not p.total:
                        return
                    if self.overlap:
                        try:
                            hold["w"] = self.comm.all_to_all_async(p.recv, p.send, p.splits, p.splits)
                            return
                        except Exception as exc:  # a transport without async collectives: the serial exchange from here on
                            import sys
                            print("tiled: async exchange unavailable (%r); continuing without overlap" % (exc,), file=sys.stderr)
                            self.overlap, self._unsplit_after = False, True  # (the substep in flight stays split)
                    self.comm.all_to_all(p.recv, p.send, p.splits, p.splits)

                def wait():
                    w = hold.pop("w", None)
                    if w is not None:
                        w.wait()
                self.e.run_native(m, start, wait)
                if getattr(self, "_unsplit_after", False):
                    self._unsplit_after = False
                    self.e.set_overlap(False)
                r.k += m
                if r.k >= r.next_migration:
                    self.migrate()
            else:
                for _ in range(m):
                    self.substep()
            left -= m
        self.substeps += n

    def num_particles(self):
        return self.e.num_particles()

    def synchronize(self):
        self.e.synchronize()

    def set_profiling(self, level, every=1):
        self.e.sim.set_profiling(level, every)
        self.e.sim.profile(reset=True)

    def profile(self):
        return self.e.sim.profile()


class VirtualTiledJob:
    """all ranks of a partition in ONE process (several ctx on one GPU, or checker engines on the CPU): the
    exchanges become local copies.  Used by the tests to check K-tile == 1-tile on a single device."""

    def __init__(self, engines, part, migrate_interval=None, overlap=False):
        assert len(engines) == part.world
        self.part = part
        self.ranks = [TiledRank(e, part, i, migrate_interval) for i, e in enumerate(engines)]
        self.overlap = bool(overlap)
        if self.overlap:
            for e in engines:
                e.set_overlap(True)

    @staticmethod
    def _a2a(recvs, sends, splits):
        """recvs[r] gets, for every peer s, the slice sends[s] addressed to r.  splits[r][s] = floats r sends to s."""
        w = len(sends)
        soff = [np.concatenate([[0], np.cumsum(splits[r])]) for r in range(w)]
        for r in range(w):
            o = 0
            for s in range(w):
                n = int(splits[s][r])
                if n:
                    recvs[r][o:o + n].copy_(sends[s][int(soff[s][r]):int(soff[s][r]) + n])
                o += n

    def substep(self):
        for r in self.ranks:
            r.e.begin()
        self._a2a([r.plan.recv for r in self.ranks], [r.plan.send for r in self.ranks], [r.plan.splits for r in self.ranks])
        if self.overlap:
            for r in self.ranks:
                r.e.interior()
        for r in self.ranks:
            r.e.end()
            r.k += 1
        if self.ranks[0].k >= self.ranks[0].next_migration:
            self.migrate()

    def migrate(self):
        table = np.stack([r.mig_scan() for r in self.ranks])
        _finish_migration(table, len(self.ranks), self.ranks,
                          lambda recvs, sends, cnt: self._a2a(recvs, sends, cnt * MIGRATE_FLOATS))

    substeps = 0

    def run(self, n):
        for _ in range(n):
            self.substep()
        self.substeps += n


# ---------------------------------------------------------------------------------------------------- native data plane
def native_setup(engine, part, rank, wire, migrate_interval=None, overlap=False, inbox_records=0, migrate_cap=None):
    """mpmhip_tiled_setup: the library derives the halo boxes from the partition, owns the buffers and runs loop, exchange
    and migration itself (include/mpmhip.h: "the native data plane").  `part` only supplies dims / cuts / margin / clip."""
    L, sim = engine.L, engine.sim
    cfg = _lib.TiledConfig()
    cfg.rank, cfg.world, cfg.margin = int(rank), int(part.world), int(part.margin)
    cfg.dims[:] = part.dims
    cfg.clip_lo[:] = part.clip[0]
    cfg.clip_hi[:] = part.clip[1]
    cfg.migrate_interval = int(migrate_interval or 0)
    cfg.migrate_cap = int(migrate_cap if migrate_cap is not None else os.environ.get("MPMHIP_TILE_MIGRATE_CAP", 64))
    cfg.wire, cfg.overlap, cfg.inbox_records = int(wire), int(bool(overlap)), int(inbox_records)
    ip = C.POINTER(C.c_int32)
    arrs = [np.asarray(c, np.int32) for c in part.cuts]
    sim._check(L.mpmhip_tiled_setup(engine.ctx, C.byref(cfg), arrs[0].ctypes.data_as(ip), arrs[1].ctypes.data_as(ip),
                                    arrs[2].ctypes.data_as(ip)))
    engine.world = part.world


def native_state(engine):
    o = (C.c_int64 * 8)()
    engine.sim._check(engine.L.mpmhip_tiled_state(engine.ctx, o))
    keys = ("substeps", "next_migration", "migrated_out", "migrations", "replans", "halo_boxes", "halo_nodes", "wire")
    return dict(zip(keys, (int(v) for v in o)))


def native_plan(engine):
    """[(peer, lo, hi, offset of the box in the peer's buffers)] of the library's current plan"""
    hb = (_lib.HaloBox * _lib.MAX_HALO_BOXES)()
    n = int(engine.sim._check(engine.L.mpmhip_tiled_plan(engine.ctx, _lib.MAX_HALO_BOXES, hb)))
    return [(int(hb[i].peer), list(hb[i].lo), list(hb[i].hi), int(hb[i].reserved)) for i in range(n)]


class _NativeJobBase:
    scaling = "strong"
    substeps = 0

    def set_overlap(self, flag):
        for e in self.engines:
            e.set_overlap(flag)
        self.overlap = bool(flag)

    def num_particles(self):
        return sum(e.num_particles() for e in self.engines)

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def set_profiling(self, level, every=1):
        for e in self.engines:
            e.sim.set_profiling(level, every)
            e.sim.profile(reset=True)

    def profile(self):
        return self.engines[0].sim.profile()

    def state(self):
        return [native_state(e) for e in self.engines]


class NativeTiledJob(_NativeJobBase):
    """this process's rank of a tiled run on the library's own data plane: Python carries 128 bytes once (the ncclUniqueId,
    or the ranks' IPC handles) over the control group `dist` (gloo) and then calls mpmhip_tiled_advance(n).
    wire: "rccl" | "ipc" (peer writes into IPC-mapped receive buffers; with use_comm the handles travel through RCCL)."""

    def __init__(self, engine, part, rank, world, wire="rccl", dist=None, migrate_interval=None, overlap=True, inbox_records=0,
                 use_comm=None):
        self.engines, self.e, self.part, self.rank, self.world = [engine], engine, part, rank, world
        self._dist, self._setup_args, self._have_comm = dist, (migrate_interval, inbox_records), False
        self._connect(wire, overlap, use_comm)

    def rewire(self, wire, use_comm=None):
        """the same job on the other wire (bench.py --wire both): the clip box is wrapped around the particles as they are now, plan
        and arena are built again; particles, clocks and the RCCL communicator stay.  Collective."""
        import torch
        e = self.e
        e.synchronize()
        if self.world > 1:
            self._dist.barrier()  # (nobody frees or unmaps an arena a peer may still write to)
        lo, hi = (C.c_int32 * 3)(), (C.c_int32 * 3)()
        e.sim._check(e.L.mpmhip_sort(e.ctx))  # (the bounds are those of the last sort: make it the current positions')
        e.sim._check(e.L.mpmhip_active_bounds(e.ctx, lo, hi))
        t = torch.tensor([-lo[0], -lo[1], -lo[2], hi[0], hi[1], hi[2]], dtype=torch.int64)
        if any(l > h for l, h in zip(lo, hi)):  # no particles on this rank (mpmhip_active_bounds: lo = 2^30, hi = -1)
            t = torch.full((6,), -(1 << 30), dtype=torch.int64)
        if self.world > 1:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        self.part.set_clip_from_bounds([-int(t[0]), -int(t[1]), -int(t[2])], [int(t[3]), int(t[4]), int(t[5])])
        self._connect(wire, self.overlap, use_comm)

    def _connect(self, wire, overlap, use_comm):
        import torch
        engine, part, rank, world, dist = self.e, self.part, self.rank, self.world, self._dist
        migrate_interval, inbox_records = self._setup_args
        self.wire = wire
        L, sim = engine.L, engine.sim
        need_comm = (wire == "rccl" or (wire == "ipc" and use_comm)) and not self._have_comm
        if need_comm:
            self._have_comm = True
            ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
                if L.mpmhip_comm_unique_id(buf) != 0:
                    raise RuntimeError("mpmhip_comm_unique_id: " + L.mpmhip_last_error(None).decode())
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            if world > 1:
                dist.broadcast(ident, src=0)
            buf = (C.c_uint8 * _lib.COMM_ID_BYTES)(*ident.tolist())
            sim._check(L.mpmhip_comm_init(engine.ctx, buf, rank, world))
        native_setup(engine, part, rank, _lib.WIRE_RCCL if wire == "rccl" else _lib.WIRE_IPC, migrate_interval, overlap, inbox_records)
        if wire == "ipc" and world > 1:
            if use_comm:
                sim._check(L.mpmhip_tiled_ipc_connect(engine.ctx, None))
            else:
                mine = (C.c_uint8 * _lib.IPC_HANDLE_BYTES)()
                sim._check(L.mpmhip_tiled_ipc_handle(engine.ctx, mine))
                rows = [torch.zeros(_lib.IPC_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(rows, torch.tensor(list(mine), dtype=torch.uint8))
                flat = (C.c_uint8 * (_lib.IPC_HANDLE_BYTES * world))(*[int(v) for r in rows for v in r.tolist()])
                sim._check(L.mpmhip_tiled_ipc_connect(engine.ctx, flat))
                dist.barrier()  # every rank has mapped every arena before anybody writes
        self.overlap = bool(overlap)
        self.parallelism = "%dx%dx%d bricks, one rank per GPU, halo + migration on the library's own data plane (%s)" % (
            part.dims + ({"rccl": "ncclSend / ncclRecv groups over xGMI", "ipc": "peer writes into IPC-mapped buffers"}[wire],))

    def run(self, n):
        self.e.sim._check(self.e.L.mpmhip_tiled_advance(self.e.ctx, int(n)))
        self.substeps += n

    def num_particles(self):
        return self.e.num_particles()

    # --- scalars of the WHOLE job, reduced inside the library (mpmhip_tiled_reduce: ncclAllReduce, or rows + epochs over the IPC wire)
    def reduce(self, values, op="sum"):
        """in-place all-reduce of up to 16 doubles over the ranks (collective); returns the reduced list"""
        v = (C.c_double * len(values))(*[float(x) for x in values])
        self.e.sim._check(self.e.L.mpmhip_tiled_reduce(self.e.ctx, v, len(values), {"sum": 0, "max": 1, "min": 2}[op]))
        return list(v)

    def calculate_energy(self):
        """(kinetic, potential) of the whole job (src/mpm.cpp:1078-1110): collective, every rank gets the same numbers"""
        return self.e.sim.calculate_energy()

    def totals(self):
        """{live particles, active blocks, sticky error word, migrated particles} of the whole job (collective)"""
        out = (C.c_int64 * 4)()
        self.e.sim._check(self.e.L.mpmhip_tiled_totals(self.e.ctx, out))
        return {"particles": int(out[0]), "active_blocks": int(out[1]), "error": int(out[2]), "migrated": int(out[3])}


class NativeVirtualJob(_NativeJobBase):
    """all ranks of a partition as ctx of ONE process on one device, on the library's data plane (MPMHIP_WIRE_LOCAL: the
    peer-write wire with plain pointers): the K-rank == 1-ctx tests and bench.py --virtual"""

    def __init__(self, engines, part, migrate_interval=None, overlap=False, inbox_records=0, halo_by_rccl=False):
        """halo_by_rccl: MPMHIP_WIRE_LOCAL_RCCL — the one-GPU pre-flight of the RCCL exchange: every rank gets its own one-rank
        communicator and its halo boxes travel as RCCL self-sends aimed at the peer ctx's receive buffer (include/mpmhip.h)"""
        assert len(engines) == part.world
        self.engines, self.part = list(engines), part
        if halo_by_rccl:
            for e in engines:
                ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
                e.sim._check(e.L.mpmhip_comm_unique_id(ident))
                e.sim._check(e.L.mpmhip_comm_init(e.ctx, ident, 0, 1))
        for r, e in enumerate(engines):
            native_setup(e, part, r, _lib.WIRE_LOCAL_RCCL if halo_by_rccl else _lib.WIRE_LOCAL, migrate_interval, overlap, inbox_records)
        self._arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        L = engines[0].L
        self._group_check(L.mpmhip_tiled_connect_local(self._arr, len(engines)), "mpmhip_tiled_connect_local")
        self.overlap = bool(overlap)

    def _group_check(self, rc, what):
        if rc < 0:  # (the failing ctx holds the message; the others may hold older ones)
            msgs = ["rank %d: %s" % (r, e.L.mpmhip_last_error(e.ctx).decode()) for r, e in enumerate(self.engines)
                    if e.L.mpmhip_last_error(e.ctx)]
            from .mpm import MPMError
            raise MPMError("%s failed (%d): %s" % (what, rc, "; ".join(msgs)))
        return rc

    def run(self, n):
        L = self.engines[0].L
        self._group_check(int(L.mpmhip_tiled_advance_group(self._arr, len(self.engines), int(n))), "mpmhip_tiled_advance_group")
        self.substeps += n

    def reduce(self, rows, op="sum"):
        """rows[r] = rank r's values (<= 16 doubles): all-reduce over the local wire; returns every rank's copy of the result"""
        K, n = len(self.engines), len(rows[0])
        v = (C.c_double * (K * n))(*[float(x) for row in rows for x in row])
        L = self.engines[0].L
        self._group_check(L.mpmhip_tiled_reduce_group(self._arr, K, v, n, {"sum": 0, "max": 1, "min": 2}[op]), "mpmhip_tiled_reduce_group")
        return [list(v[r * n:(r + 1) * n]) for r in range(K)]

    def calculate_energy(self):
        """(kinetic, potential) of the whole job (src/mpm.cpp:1078-1110) — mpmhip_calculate_energy_group"""
        k, p = C.c_double(), C.c_double()
        L = self.engines[0].L
        self._group_check(L.mpmhip_calculate_energy_group(self._arr, len(self.engines), C.byref(k), C.byref(p)), "mpmhip_calculate_energy_group")
        return k.value, p.value

    def totals(self):
        out = (C.c_int64 * 4)()
        L = self.engines[0].L
        self._group_check(L.mpmhip_tiled_totals_group(self._arr, len(self.engines), out), "mpmhip_tiled_totals_group")
        return {"particles": int(out[0]), "active_blocks": int(out[1]), "error": int(out[2]), "migrated": int(out[3])}


# ---------------------------------------------------------------------------------------------------- bench glue
def scene_groups(cfg):
    """the synthetic workload of a bench config as [(particle type, lower corner in cells (3,), cube edge in cells)]:
    one cube (C2 / C3), or the 2x2x2 arrangement of clusters with alternating materials (C5).  Every group is one
    `add_particles` call; bench.py's single-GPU builder and the tiled builder below both go through this list."""
    res, cells = cfg["res"], cfg["cells"]
    if "clusters" in cfg:
        out, k = [], 0
        for ox in cfg["clusters"]:
            for oy in cfg["clusters"]:
                for oz in cfg["clusters"]:
                    out.append(("water" if k % 2 == 0 else "elastic", (ox, oy, oz), cells))
                    k += 1
        return out
    lo = res // 2 - cells // 2
    return [(cfg["material"], (lo, lo, lo), cells)]


def group_positions(group, dx):
    from .mpm import lattice_cube
    _, lo, cells = group
    return lattice_cube(0, cells, dx) + (np.asarray(lo, np.float64) * dx).astype(np.float32)


def scene_partition(cfg, world, margin, dims=None):
    """balanced bricks over ALL groups of the scene (union of the clusters' histograms)"""
    res = cfg["res"]
    dx = 1.0 / res
    hists = [np.zeros(res, np.int64) for _ in range(3)]
    lo, hi = np.full(3, 1 << 30), np.full(3, -1)
    for g in scene_groups(cfg):
        b = base_cells(group_positions(g, dx), dx)
        for a in range(3):
            hists[a] += np.bincount(np.clip(b[:, a], 0, res - 1), minlength=res)
        lo, hi = np.minimum(lo, b.min(0)), np.maximum(hi, b.max(0) + 1)
    return Partition.from_histograms((res,) * 3, world, hists, lo, hi, margin, dims=dims)


def build_rank_sim(tm, cfg, part, rank, device, extra_cfg=None):
    """the ctx of one rank: EVERY rank registers EVERY group in the same order (group ids travel with migrating
    particles), holds the particles whose base cell lies in its brick, with global creation ids"""
    from .mpm import F_ID
    res = cfg["res"]
    dx = 1.0 / res
    groups = scene_groups(cfg)
    mine, ids, offset = [], [], 0
    for g in groups:
        x = group_positions(g, dx)
        m = np.nonzero(part.rank_of_cells(base_cells(x, dx)) == rank)[0]
        mine.append(x[m])
        ids.append((offset + m).astype(np.int32))
        offset += len(x)
    n_mine = sum(len(x) for x in mine)
    sim = tm.create_simulation3("mpm").initialize(dict(
        res=(res,) * 3, delta_x=dx, base_delta_t=cfg.get("dt", 1e-4), gravity=(0, -10, 0), device=device,
        max_particles=int(n_mine * 1.5) + (1 << 16), **(extra_cfg or {})))
    sim.set_levelset(tm.mpm.LevelSet(friction=-1.0).add_plane((0, 1, 0), d=-0.1))
    for g, x in zip(groups, mine):
        sim.add_particles(dict(type=g[0], positions=x))
    sim._ensure_ctx()
    if n_mine:
        sim.upload(F_ID, np.concatenate(ids))  # creation ids are global
    return sim, offset


def make_tiled_job(tm, cfg, rank, world, local_rank, margin=4, migrate_interval=None, comm=None):
    """bench.py, N > 1: every rank generates the same synthetic scene (one cube, or the C5 clusters with their two
    materials), keeps its brick's particles."""
    import os

    import torch
    margin = int(os.environ.get("MPMHIP_TILE_MARGIN", margin))
    import torch.distributed as dist

    part = scene_partition(cfg, world, margin)
    sim, _ = build_rank_sim(tm, cfg, part, rank, local_rank)
    engine = HipEngine(sim, local_rank)
    overlap = os.environ.get("MPMHIP_TILE_OVERLAP", "1") != "0"  # ("auto": bench.py times both and keeps the faster)
    comm = comm or DistComm(dist, torch.device("cuda", local_rank))
    return TiledJob(engine, part, comm, migrate_interval, overlap=overlap)


def make_native_job(tm, cfg, rank, world, local_rank, wire="rccl", dist=None, margin=4, migrate_interval=None):
    """bench.py, N > 1 on the library's own data plane: the same scene and partition as make_tiled_job; Python hands over
    the partition and 64 / 128 bytes of wire set-up, then only calls mpmhip_tiled_advance"""
    margin = int(os.environ.get("MPMHIP_TILE_MARGIN", margin))
    part = scene_partition(cfg, world, margin)
    sim, _ = build_rank_sim(tm, cfg, part, rank, local_rank)
    engine = HipEngine(sim, local_rank)
    overlap = os.environ.get("MPMHIP_TILE_OVERLAP", "1") != "0"  # ("auto": bench.py times both and keeps the faster)
    return NativeTiledJob(engine, part, rank, world, wire=wire, dist=dist, migrate_interval=migrate_interval, overlap=overlap)
