"""TEST INFRASTRUCTURE: a CPU engine for taichi_mpm_amd.tiled built on the oracle, so the partition / halo-plan /
exchange / migration host logic can run under gloo (world_size 2) and in-process without a GPU.  It implements the
engine interface of tiled.HipEngine with dense numpy grids; it is never imported by the package."""
import numpy as np
import torch

from oracle import oracle as orc
from taichi_mpm_amd import tiled

NF = tiled.MIGRATE_FLOATS


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self, cfg, state, dx):
        self.cfg, self.s, self.dx = cfg, state, dx

    def alloc(self, nfloats):
        return torch.empty(int(nfloats), dtype=torch.float32)

    def configure(self, part, rank, plan):
        self.part, self.rank, self.plan, self.world = part, rank, plan, part.world

    def _box_view(self, grid, lo, hi):
        return grid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]

    def begin(self):
        self.own = orc.p2g(self.cfg, self.s)
        send = self.plan.send.numpy()
        for (peer, lo, hi), off, v in zip(self.plan.boxes, self.plan.offsets, self.plan.vol):
            send[off:off + 4 * v] = self._box_view(self.own, lo, hi).reshape(-1)

    def end(self):
        recv = self.plan.recv.numpy()
        tot = np.zeros_like(self.own)
        own_added = False
        for (peer, lo, hi), off, v in zip(self.plan.boxes, self.plan.offsets, self.plan.vol):  # ascending peer
            if not own_added and peer > self.rank:
                tot += self.own
                own_added = True
            view = self._box_view(tot, lo, hi)
            view += recv[off:off + 4 * v].reshape(view.shape)
        if not own_added:
            tot += self.own
        self.total = tot
        grid = orc.grid_update(self.cfg, tot.copy())
        orc.g2p(self.cfg, self.s, grid)
        keep = orc.clear_boundary(self.cfg, self.s)
        if not keep.all():
            self._select(keep)

    def _select(self, keep):
        s = self.s
        for k in ("x", "v", "B", "F", "aux", "gid", "ids"):
            setattr(s, k, getattr(s, k)[keep].copy())

    def _dest(self):
        return self.part.rank_of_cells(tiled.base_cells(self.s.x, self.dx))

    def migration_scan(self):
        lo, hi = self.active_bounds()
        speed = float(np.abs(self.s.v).max() * self.cfg.dt / self.dx) if self.s.n else 0.0
        return self.leaver_counts(), lo, hi, speed

    def leaver_counts(self):
        d = self._dest()
        b = tiled.base_cells(self.s.x, self.dx)
        lo, hi = self.part.brick(self.rank)
        assert (b >= np.array(lo) - self.part.margin).all() and (b < np.array(hi) + self.part.margin).all(), \
            "a particle left the margin"
        c = np.bincount(d, minlength=self.world).astype(np.int64)
        c[self.rank] = 0
        return c

    def export_leavers(self, counts, buf):
        s, d = self.s, self._dest()
        leave = d != self.rank
        order = np.argsort(d[leave], kind="stable")
        idx = np.nonzero(leave)[0][order]
        rec = np.zeros((len(idx), NF), np.float32)
        rec[:, 0:3] = s.x[idx]; rec[:, 3:6] = s.v[idx]; rec[:, 6:15] = s.B[idx]; rec[:, 15:24] = s.F[idx]
        rec[:, 24] = s.aux[idx]
        rec[:, 25] = s.gid[idx].view(np.float32)
        rec[:, 26] = s.ids[idx].view(np.float32)
        assert np.array_equal(np.bincount(d[idx], minlength=self.world), counts)
        buf.numpy()[:rec.size] = rec.reshape(-1)
        self._select(~leave)

    def import_particles(self, buf, n):
        if not n:
            return
        rec = buf.numpy()[:n * NF].reshape(n, NF)
        s = self.s
        s.x = np.concatenate([s.x, rec[:, 0:3]]); s.v = np.concatenate([s.v, rec[:, 3:6]])
        s.B = np.concatenate([s.B, rec[:, 6:15]]); s.F = np.concatenate([s.F, rec[:, 15:24]])
        s.aux = np.concatenate([s.aux, rec[:, 24]])
        s.gid = np.concatenate([s.gid, rec[:, 25].copy().view(np.int32)])
        s.ids = np.concatenate([s.ids, rec[:, 26].copy().view(np.int32)])
        for k in ("x", "v", "B", "F", "aux", "gid", "ids"):
            setattr(s, k, np.ascontiguousarray(getattr(s, k)))

    def active_bounds(self):
        if self.s.n == 0:
            return np.full(3, 1 << 30, np.int32), np.full(3, -1, np.int32)
        b = tiled.base_cells(self.s.x, self.dx)
        return b.min(0).astype(np.int32), (b.max(0) + 1).astype(np.int32)

    def num_particles(self):
        return self.s.n

    def synchronize(self):
        pass


def subset(state, sel):
    return orc.State(state.x[sel], state.v[sel], state.B[sel], state.F[sel], state.aux[sel], state.gid[sel],
                     state.gparams, state.gtype, state.ids[sel])
