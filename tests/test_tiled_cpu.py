"""CPU tests of the multi-GPU host logic (taichi_mpm_amd/tiled.py): partition, halo plan, the all_to_all exchange
and the migration protocol, with the oracle as the per-rank engine (tests/fake_engine.py).  The distributed case
runs under torch.distributed `gloo` with world_size 2 — the same TiledJob code path bench.py drives over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from taichi_mpm_amd import tiled
from tests.common import lattice_cube, make_state, rel_l2
from tests.fake_engine import OracleEngine, subset

RES, DX, DT = 24, 1.0 / 24, 1e-4
PLANES = [(0.0, 1.0, 0.0, -0.3)]


def _state():
    x = lattice_cube(RES, 8, 15, DX, jitter=0.2, seed=41)
    return make_state(x, "snow", DX, perturb_F=0.02, seed=42, vel_scale=20.0)


def _cfg(orc):
    return orc.make_config(RES, DX, DT, planes=PLANES, friction=0.4)


def test_brick_dims_and_balanced_cuts():
    assert tiled.brick_dims(1) == (1, 1, 1)
    assert tiled.brick_dims(2) == (2, 1, 1)
    assert tiled.brick_dims(4) == (2, 2, 1)
    assert tiled.brick_dims(8) == (2, 2, 2)
    assert tiled.brick_dims(6) == (3, 2, 1)
    # BASELINE configs[3]: 100^3 cells x 8 in a 256^3 grid on 8 GPUs -> 1M particles per brick, +-2 %
    from taichi_mpm_amd.mpm import lattice_cube as lc
    x = lc(128 - 50, 128 + 50, 1.0 / 256)[::8]  # one lattice site in 8 is enough for the histogram
    part = tiled.Partition.balanced((256,) * 3, 8, x, 1.0 / 256, margin=4)
    counts = np.bincount(part.rank_of_cells(tiled.base_cells(x, 1.0 / 256)), minlength=8)
    assert counts.sum() == len(x) and counts.max() / counts.min() < 1.07, counts
    # slabs: 4 ranks along one axis, cell-granular cuts
    part = tiled.Partition.balanced((256,) * 3, 4, x, 1.0 / 256, margin=4, dims=(4, 1, 1))
    counts = np.bincount(part.rank_of_cells(tiled.base_cells(x, 1.0 / 256)), minlength=4)
    assert counts.max() / counts.min() < 1.05, counts


def test_scene_partition_balances_the_c5_clusters_over_8_ranks():
    """BASELINE configs[4] on 8 GPUs: the scene is a list of groups (8 clusters, two materials); the bricks are balanced
    over the union of the clusters' histograms and every particle has exactly one owner (no GPU needed)"""
    cfg = dict(res=128, cells=20, material="water+elastic", clusters=(20, 84))
    groups = tiled.scene_groups(cfg)
    assert [g[0] for g in groups] == ["water", "elastic"] * 4 and len({g[1] for g in groups}) == 8
    part = tiled.scene_partition(cfg, 8, margin=4)
    assert part.dims == (2, 2, 2)
    counts = np.zeros(8, np.int64)
    for g in groups:
        x = tiled.group_positions(g, 1.0 / 128)
        assert len(x) == 20 ** 3 * 8
        counts += np.bincount(part.rank_of_cells(tiled.base_cells(x, 1.0 / 128)), minlength=8)
    assert counts.sum() == 8 * 20 ** 3 * 8 and counts.min() == counts.max()  # one cluster per brick
    # clusters that do not touch: the cut lies in the MIDDLE of the empty stretch between them (base cells 19..39 | 83..103 -> 61), not
    # along the face of the lower cluster, where every particle that moves up a cell would cross into the next brick
    assert all(part.cuts[a] == [0, 40 + (83 - 40) // 2, 128] for a in range(3)), part.cuts
    assert tiled.balanced_cuts(None, 16, 2, hist=np.array([0, 0, 5, 5, 0, 0, 0, 0, 0, 0, 5, 5, 0, 0, 0, 0])) == [0, 7, 16]
    assert tiled.balanced_cuts(None, 8, 2, hist=np.array([0, 3, 3, 3, 3, 3, 3, 0])) == [0, 4, 8]  # (no gap at the balance point: as before)
    # a single cube (C3) goes through the same builders
    c3 = dict(res=256, cells=100, material="sand")
    assert tiled.scene_groups(c3) == [("sand", (78, 78, 78), 100)]
    # the clip box wraps the occupied part of the grid
    lo, hi = part.clip
    assert all(lo[a] <= 20 - 4 and hi[a] >= 84 + 20 + 4 for a in range(3))


def test_halo_boxes_are_symmetric_and_cover_every_shared_node():
    x = _state().x
    for world, dims in ((2, None), (8, None), (4, (4, 1, 1)), (6, None)):
        part = tiled.Partition.balanced((RES,) * 3, world, x, DX, margin=2, dims=dims)
        shape = (RES + 1,) * 3
        holders = np.zeros(shape, np.int32)
        for r in range(world):
            lo, hi = part.node_box(r)
            holders[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] += 1
        for r in range(world):
            plan = tiled.HaloPlan(part, r, lambda n: torch.empty(int(n)))
            peers = [b[0] for b in plan.boxes]
            assert peers == sorted(peers) and r not in peers
            cover = np.zeros(shape, np.int32)
            for (peer, lo, hi), v in zip(plan.boxes, plan.vol):
                assert part.overlap(peer, r) == (lo, hi)  # same box seen from the other side
                assert plan.splits[peer] == 4 * v
                cover[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] += 1
            lo, hi = part.node_box(r)
            mine = np.zeros(shape, bool)
            mine[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
            # every node this rank holds is covered once per OTHER holder
            assert np.array_equal(cover[mine], holders[mine] - 1)


def _reference_run(orc, s, steps):
    ref = s.copy()
    cfg = _cfg(orc)
    for _ in range(steps):
        orc.substep(cfg, ref)
    order = np.argsort(ref.ids)
    return {k: getattr(ref, k)[order] for k in ("x", "v", "F", "B", "aux", "ids")}


def _collect(states):
    out = {k: np.concatenate([getattr(s, k) for s in states]) for k in ("x", "v", "F", "B", "aux", "ids")}
    order = np.argsort(out["ids"])
    return {k: v[order] for k, v in out.items()}


def _compare(got, ref):
    assert np.array_equal(got["ids"], ref["ids"])
    assert np.abs(got["x"] - ref["x"]).max() <= 1e-6
    assert rel_l2(got["v"], ref["v"]) <= 1e-4
    assert rel_l2(got["F"], ref["F"]) <= 1e-4
    assert rel_l2(got["aux"], ref["aux"]) <= 1e-5


@pytest.mark.parametrize("world,dims", [(2, None), (4, None), (3, (1, 1, 3))])
def test_virtual_ranks_match_single_oracle(orc, world, dims):
    s = _state()
    steps = 6
    ref = _reference_run(orc, s, steps)
    part = tiled.Partition.balanced((RES,) * 3, world, s.x, DX, margin=2, dims=dims)
    owner = part.rank_of_cells(tiled.base_cells(s.x, DX))
    engines = [OracleEngine(_cfg(orc), subset(s, owner == r), DX) for r in range(world)]
    job = tiled.VirtualTiledJob(engines, part, migrate_interval=2)
    job.run(steps)
    assert sum(r.migrated_out for r in job.ranks) > 0
    # holders of a node agree bit-for-bit on its total (rank-ordered sums)
    for a in range(world):
        for b in range(a + 1, world):
            o = part.overlap(a, b)
            if o:
                lo, hi = o
                sl = (slice(lo[0], hi[0]), slice(lo[1], hi[1]), slice(lo[2], hi[2]))
                ta, tb = engines[a].total[sl], engines[b].total[sl]
                both = (engines[a].own[sl][..., 3] > 0) & (engines[b].own[sl][..., 3] > 0)
                assert np.array_equal(ta[both], tb[both])
    _compare(_collect([e.s for e in engines]), ref)


def test_adaptive_migration_schedule_follows_the_fastest_particle(orc):
    """without an explicit interval the next migration is scheduled from the measured top speed (cells per substep):
    half the time the fastest particle needs to cross the margin, never sooner than the CFL schedule (= margin
    substeps), never later than the cap — and the run still equals the single-domain run (the engines assert that
    no particle ever leaves the margin)"""
    s = _state()
    steps = 80
    ref = _reference_run(orc, s, steps)
    part = tiled.Partition.balanced((RES,) * 3, 2, s.x, DX, margin=2)
    owner = part.rank_of_cells(tiled.base_cells(s.x, DX))
    engines = [OracleEngine(_cfg(orc), subset(s, owner == r), DX) for r in range(2)]
    job = tiled.VirtualTiledJob(engines, part)
    seen = {}

    def counting(e):
        plain = e.migration_scan

        def scan():
            out = plain()
            k = job.ranks[0].k
            seen[k] = max(seen.get(k, 0.0), out[3])  # the schedule uses the fastest particle of ALL ranks
            return out
        return scan
    for e in engines:
        e.migration_scan = counting(e)
    job.run(steps)
    scans = sorted(seen.items())
    ks = [k for k, _ in scans]
    assert ks[0] == 2 and len(ks) < steps // 4, ks  # first scan on the CFL schedule, then far fewer than every 2
    for (k0, sp), k1 in zip(scans, ks[1:]):  # every gap = what the speed measured at its start allows
        q = np.ceil(sp * (1 << 20)) / (1 << 20)
        assert k1 - k0 == max(2, min(64, int(0.5 * 2 / q))), (k0, k1, sp)
    assert sum(r.migrated_out for r in job.ranks) > 0
    _compare(_collect([e.s for e in engines]), ref)
    # fast particles: back to the CFL schedule; an explicit interval or cap 0 switches the adaptation off
    r = job.ranks[0]
    r.k = 100
    r.schedule(0.7)
    assert r.next_migration == 102
    r.schedule(1e-9)
    assert r.next_migration == 164
    fixed = tiled.VirtualTiledJob([OracleEngine(_cfg(orc), subset(s, owner == q), DX) for q in range(2)], part,
                                  migrate_interval=2).ranks[0]
    fixed.schedule(1e-9)
    assert fixed.next_migration == 2


def test_halo_boxes_follow_the_particles(orc):
    """halo boxes are clipped to the occupied part of the grid; when the particles approach the clip box the
    plan is rebuilt (same decision on all ranks) and the result is unchanged"""
    s = _state()
    steps = 6
    ref = _reference_run(orc, s, steps)
    part = tiled.Partition.balanced((RES,) * 3, 2, s.x, DX, margin=2)
    b = tiled.base_cells(s.x, DX)
    full = tiled.Partition((RES,) * 3, part.dims, part.cuts, 2)
    part.clip = (list(map(int, b.min(0))), list(map(int, b.max(0) + 3)))  # no slack at all: must replan at once
    assert sum(tiled.HaloPlan(part, 0, lambda n: torch.empty(int(n))).vol) < \
        sum(tiled.HaloPlan(full, 0, lambda n: torch.empty(int(n))).vol)  # clipped boxes are smaller
    owner = part.rank_of_cells(b)
    engines = [OracleEngine(_cfg(orc), subset(s, owner == r), DX) for r in range(2)]
    job = tiled.VirtualTiledJob(engines, part, migrate_interval=1)
    job.run(steps)
    assert all(getattr(r, "replans", 0) >= 1 for r in job.ranks)
    _compare(_collect([e.s for e in engines]), ref)


def test_particles_outside_the_clipped_halo_region_are_an_error_not_a_silent_loss(orc):
    """the halo boxes are cut to the clip box; a migration that finds particles whose stencils reach beyond it (their speed
    more than doubled since the last check) must say so: the sums of the substeps in between may have missed mass
    (the native data plane makes the same test in csrc/tiled_api.h: tn_mig_c)"""
    s = _state()
    part = tiled.Partition.balanced((RES,) * 3, 2, s.x, DX, margin=2)
    b = tiled.base_cells(s.x, DX)
    lo, hi = b.min(0), b.max(0) + 1
    assert part.clip_holds(lo, hi) and part.clip_covers(lo, hi)  # a fresh clip box has room for 2 margin + 1 cells of travel
    assert part.clip_holds(lo - 2 * part.margin, hi + 2 * part.margin)
    part.clip = (list(map(int, lo + 1)), list(map(int, hi + 2)))  # one layer of particles already outside on the low side
    assert not part.clip_holds(lo, hi)
    owner = part.rank_of_cells(b)
    engines = [OracleEngine(_cfg(orc), subset(s, owner == r), DX) for r in range(2)]
    job = tiled.VirtualTiledJob(engines, part, migrate_interval=1)
    with pytest.raises(RuntimeError, match="left the clipped halo region"):
        job.run(2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        s = _state()
        part = tiled.Partition.balanced((RES,) * 3, world, s.x, DX, margin=2)
        owner = part.rank_of_cells(tiled.base_cells(s.x, DX))
        eng = OracleEngine(_cfg(orc), subset(s, owner == rank), DX)
        job = tiled.TiledJob(eng, part, tiled.DistComm(dist, torch.device("cpu")), migrate_interval=2)
        job.run(steps)
        n = torch.tensor([job.num_particles()])
        dist.all_reduce(n)
        st = eng.s
        q.put((rank, int(n.item()), job.r.migrated_out,
               {k: getattr(st, k) for k in ("x", "v", "F", "B", "aux", "ids")}))
    finally:
        dist.destroy_process_group()


def test_distributed_gloo_world2_matches_single_oracle(orc):
    """TiledJob over torch.distributed (gloo, 2 processes): halo all_to_all every substep, migration every 2."""
    s = _state()
    steps = 6
    ref = _reference_run(orc, s, steps)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    assert res[0][1] == res[1][1] == s.n  # all_reduce'd live count
    assert res[0][2] + res[1][2] > 0  # migration happened
    got = {k: np.concatenate([r[3][k] for r in res]) for k in res[0][3]}
    order = np.argsort(got["ids"])
    _compare({k: v[order] for k, v in got.items()}, ref)
