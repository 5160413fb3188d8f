"""The deterministic mode (include/mpmhip.h: mpmhip_config.deterministic; env MPMHIP_DETERMINISTIC): behind every sort the particles
of a cell are put in ascending creation id, so the one nondeterministic step of a substep — the in-cell ranks the counting sort
hands out with atomics — no longer decides a summation order.  What must then hold, BIT FOR BIT:
  * two runs of one scene;
  * a tiled job run twice, and the same job over another wire: K virtual ranks on the local wire (one process) against K PROCESSES on
    the IPC wire — the only way a wire-ordering bug will ever be told from rounding noise;
(the sort's forms, the two G2P walks, the physical reorder and a grown ctx: tests/test_gpu_parity.py) and to the regrouping of <= 8
block tiles per halo node: a K-rank job against the one-ctx run.  The reference's scalar path is deterministic for the same reason its
sort key is unique ((offset >> 5) << 25 | i, src/mpm.cpp:785-795)."""
import numpy as np
import pytest

from tests.common import lattice_cube, make_state, rel_l2
from tests.test_gpu_tiled import DT, DX, PLANES, RES, _gather, _two_material_state

pytestmark = pytest.mark.gpu
FIELDS = ("x", "v", "F", "aux")


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def _scene():
    rng = np.random.default_rng(61)
    dense = lattice_cube(RES, 8, 16, DX, jitter=0.25, seed=60)
    dense = np.concatenate([dense, dense + np.float32(2e-3)])  # 16 per cell: runs of the sort that cross lanes
    spray = (rng.uniform(7.0, 25.0, (4000, 3)) * DX).astype(np.float32)
    x = np.concatenate([dense, spray])
    s = make_state(x[rng.permutation(len(x))], "sand", DX, perturb_F=0.03, seed=62, vel_scale=6.0)
    s.v[-50:] = (0.0, 0.0, 500.0)  # these leave through the wall: dead slots, compaction
    return s


def _run(tm, s, steps, **cfg):
    sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, **cfg))
    ls = tm.mpm.LevelSet(friction=0.4)
    for p in PLANES:
        ls.add_plane(p[:3], d=p[3])
    sim.set_levelset(ls)
    sim.add_particles(dict(type="sand", positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
    sim.run_substeps(steps)
    out = sim.get_particles()
    sim.close()
    return out


def test_two_runs_of_one_scene_agree_bit_for_bit(tm):
    """30 substeps of a sand scene with crowded cells, spray and leavers, slots in shuffled order (the LDS-hash path of the ranks after
    the first sort): three runs in the deterministic mode give the same bits; the default mode stays within the run-to-run spread of
    DESIGN.md section 2 of them"""
    s = _scene()
    runs = [_run(tm, s, 30, deterministic=True) for _ in range(3)]
    assert len(runs[0]["id"]) < s.n
    for r in runs[1:]:
        assert np.array_equal(r["id"], runs[0]["id"])
        for f in FIELDS:
            assert np.array_equal(r[f], runs[0][f]), (f, float(np.abs(r[f] - runs[0][f]).max()))
    loose = _run(tm, s, 30)
    assert np.array_equal(loose["id"], runs[0]["id"])
    assert np.abs(loose["x"] - runs[0]["x"]).max() <= 2e-6 and rel_l2(loose["v"], runs[0]["v"]) <= 1e-3 and rel_l2(loose["F"], runs[0]["F"]) <= 1e-3



def _crowded_scene():
    """cells of 1, <= 16, 17..64 and > 64 particles; one block of more than 704 entries (k_sort.h: CO_CAP — the per-lane walk of
    k_cell_order_blocks); slots shuffled"""
    rng = np.random.default_rng(71)
    cell = lambda c, n: ((np.asarray(c, np.float32) + rng.uniform(0.05, 0.95, (n, 3))) * DX).astype(np.float32)
    parts = [cell((12, 14, 12), 150), cell((13, 14, 12), 40), cell((12, 15, 13), 70), cell((14, 14, 14), 17)]
    for c in np.ndindex(4, 4, 4):  # block (5, 5, 5): 64 cells x 14 = 896 entries
        parts.append(cell((20 + c[0], 20 + c[1], 20 + c[2]), 14))
    parts.append((rng.uniform(6.0, 26.0, (3000, 3)) * DX).astype(np.float32))  # singles and small cells
    x = np.concatenate(parts)
    return make_state(x[rng.permutation(len(x))], "jelly", DX, perturb_F=0.01, seed=72, vel_scale=2.0)


def test_every_path_of_the_ordering_launch_gives_the_same_bits(tm, monkeypatch):
    """the wave-per-block ordering launch (cells ordered in LDS; crowded cells and an over-full block on their own paths), the
    lane-per-cell one (MPMHIP_CELL_ORDER=0), and ids gathered from the records instead of the compact array (the mode switched on
    again in mid-run: the next sort finds keys a G2P wrote before the switch): one result, bit for bit"""
    s = _crowded_scene()

    def run(toggle=False, **cfg):
        sim = tm.create_simulation3("mpm").initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, deterministic=True, **cfg))
        sim.add_particles(dict(type="jelly", positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
        for k in range(4):
            sim.run_substeps(3)
            if toggle:
                sim.set_deterministic(True)
        out = sim.get_particles()
        sim.close()
        return out

    ref = run()
    assert len(ref["id"]) > 3000 and np.isfinite(ref["x"]).all()  # (spray near the walls is refused at creation, as in the reference)
    again, toggled = run(), run(toggle=True)
    monkeypatch.setenv("MPMHIP_CELL_ORDER", "0")
    lanes = run()
    monkeypatch.delenv("MPMHIP_CELL_ORDER")
    for name, r in (("again", again), ("records' ids", toggled), ("lane per cell", lanes)):
        assert np.array_equal(r["id"], ref["id"]), name
        for f in FIELDS:
            assert np.array_equal(r[f], ref[f]), (name, f, float(np.abs(r[f] - ref[f]).max()))


def _det_sim(tm, s, sel, ids, cap, device=0):
    from taichi_mpm_amd.mpm import F_ID
    sim = tm.create_simulation3("mpm")
    sim.initialize(dict(res=(RES,) * 3, delta_x=DX, base_delta_t=DT, max_particles=cap, reorder_interval=0, deterministic=True, device=device))
    ls = tm.mpm.LevelSet(friction=0.4)
    for p in PLANES:
        ls.add_plane(p[:3], d=p[3])
    sim.set_levelset(ls)
    names = {v: k for k, v in tm.MATERIAL_IDS.items()}
    for gi in range(len(s.gtype)):
        m = sel & (s.gid == gi)
        sim.add_particles(dict(type=names[int(s.gtype[gi])], positions=s.x[m], velocities=s.v[m], F=s.F[m], B=s.B[m], aux=s.aux[m],
                               params=s.gparams[gi]))
    order = np.concatenate([np.nonzero(sel & (s.gid == gi))[0] for gi in range(len(s.gtype))])
    if len(order):
        sim.upload(F_ID, ids[order].astype(np.int32))
    else:
        sim._ensure_ctx()
    return sim


def _virtual(tm, world, overlap, steps, halo_by_rccl=False):
    from taichi_mpm_amd import tiled
    s = _two_material_state()
    ids = np.arange(s.n)
    part = tiled.Partition.balanced((RES,) * 3, world, s.x, DX, margin=2)
    owner = part.rank_of_cells(tiled.base_cells(s.x, DX))
    sims = [_det_sim(tm, s, owner == r, ids, s.n + 1024) for r in range(world)]
    job = tiled.NativeVirtualJob([tiled.HipEngine(sim, 0) for sim in sims], part, migrate_interval=2, overlap=overlap, halo_by_rccl=halo_by_rccl)
    job.run(steps)
    got, st = _gather(sims), job.state()
    assert all(t["wire"] == (4 if halo_by_rccl else 3) for t in st)
    for sim in sims:
        sim.close()
    assert sum(t["migrated_out"] for t in st) > 0, "the scene must exercise migration"
    return got


@pytest.mark.parametrize("overlap", [False, True], ids=["serial", "overlap_split"])
@pytest.mark.parametrize("world", [2, 8])
def test_k_virtual_ranks_run_twice_agree_bit_for_bit_and_with_one_ctx_to_the_tile_regrouping(tm, world, overlap):
    """the library's own data plane (halo boxes written into the peers' buffers, epochs, migration every 2 substeps) twice: the same
    bits (arrivals of a migration land in slots handed out by atomics — the creation-id order makes that irrelevant); against the
    one-ctx run: a halo node's total is the rank-ordered sum of rank partials there and the q-ordered sum of <= 8 block tiles here, so
    the results agree to a regrouping of those few terms, not to the bit"""
    steps = 12
    a, b = _virtual(tm, world, overlap, steps), _virtual(tm, world, overlap, steps)
    assert np.array_equal(a["id"], b["id"])
    for f in FIELDS + ("B",):
        assert np.array_equal(a[f], b[f]), (f, float(np.abs(a[f] - b[f]).max()))
    s = _two_material_state()
    one = _det_sim(tm, s, np.ones(s.n, bool), np.arange(s.n), s.n + 1024)
    one.run_substeps(steps)
    ref = one.get_particles()
    one.close()
    assert np.array_equal(a["id"], ref["id"])
    assert np.abs(a["x"] - ref["x"]).max() <= 2e-7
    assert rel_l2(a["v"], ref["v"]) <= 2e-5 and rel_l2(a["F"], ref["F"]) <= 2e-5


def _ipc_worker(rank, world, port, steps, overlap, q):
    import os

    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("MPMHIP_TILE_WAIT_S", "10")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import taichi_mpm_amd as tm
        from taichi_mpm_amd import tiled
        tm.load()
        s = _two_material_state()
        part = tiled.Partition.balanced((RES,) * 3, world, s.x, DX, margin=2)
        owner = part.rank_of_cells(tiled.base_cells(s.x, DX))
        sim = _det_sim(tm, s, owner == rank, np.arange(s.n), s.n + 1024)
        job = tiled.NativeTiledJob(tiled.HipEngine(sim, 0), part, rank, world, wire="ipc", dist=dist, migrate_interval=2, overlap=overlap)
        job.run(steps)
        job.synchronize()
        dist.barrier()  # nobody unmaps an arena a peer may still write to
        p = sim.get_particles(sort_by_id=False)
        q.put((rank, {k: p[k] for k in FIELDS + ("B", "id")}))
        sim.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["serial", "overlap_split"])
def test_two_processes_over_the_ipc_wire_give_the_bits_of_two_virtual_ranks(tm, overlap):
    """the SAME two-brick job over two wires: MPMHIP_WIRE_LOCAL (two ctx of this process, one stream: the kernels of the ranks can
    only run one after another) and MPMHIP_WIRE_IPC (two processes whose kernels really race: peer writes into mapped arenas, epoch
    flags, bounded waits).  In the deterministic mode every bit must agree: an exchange that read a box before its writer was done,
    or a parity mix-up of the two receive buffers, has nowhere to hide."""
    import socket

    import torch.multiprocessing as mp
    steps, world = 12, 2
    want = _virtual(tm, world, overlap, steps)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, steps, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {k: np.concatenate([r[1][k] for r in res]) for k in res[0][1]}
    order = np.argsort(got["id"], kind="stable")
    got = {k: v[order] for k, v in got.items()}
    assert np.array_equal(got["id"], want["id"])
    for f in FIELDS + ("B",):
        assert np.array_equal(got[f], want[f]), (f, float(np.abs(got[f] - want[f]).max()))


@pytest.mark.parametrize("overlap", [False, True], ids=["serial", "overlap_split"])
@pytest.mark.parametrize("world", [2, 8])
def test_rccl_exchange_path_preflight_on_one_gpu_gives_the_bits_of_the_local_wire(tm, world, overlap):
    """MPMHIP_WIRE_LOCAL_RCCL: the halo boxes of a K-rank job on ONE GPU through the code of the RCCL wire — ncclGroupStart, one
    ncclSend + ncclRecv per box, ncclGroupEnd, on the side stream fenced by two events when the substep is split, ONE receive buffer
    (the peer-write wires have two) — as self-sends on a one-rank communicator per rank whose receives land in the peer ctx's buffer.
    200 substeps with migrations every 2, real RCCL kernels between the substep's own: every bit must equal the local wire's.  What
    stays untested without a second GPU is the transport under RCCL, not the ordering of the exchange against the kernels."""
    steps = 200
    want = _virtual(tm, world, overlap, steps)
    got = _virtual(tm, world, overlap, steps, halo_by_rccl=True)
    assert np.array_equal(got["id"], want["id"]) and len(got["id"]) > 1000
    for f in FIELDS + ("B",):
        assert np.array_equal(got[f], want[f]), (f, float(np.abs(got[f] - want[f]).max()))
