#!/usr/bin/env python
"""bench.py — particle-steps/sec of the MLS-MPM substep (sort + P2G + grid + G2P) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched under torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one substep (src/mpm.cpp:452-575) of the workload
BASELINE.json's metric is quoted on: config C3 = 256^3 grid, 100^3 cells x 8 = 8 000 000 Drucker-Prager sand
particles, fp32, inputs resident in HBM before the timed region.

Sequence:  W warm-up substeps (untimed) | a short untimed pass with every phase bracketed by hipEvents (phase
table, picks the dominant kernel) | barrier | K timed substeps, only the dominant kernel bracketed (two event
records per substep; bracketing all phases would add ~20 us of idle GPU per substep) | barrier.

Extra objects on the line:
  roofline      dominant kernel (the slower of k_p2g / k_g2p): algorithmic bytes per launch / its average launch
                duration over the K timed substeps, hipEvents recorded on the ctx stream (mpmhip_set_profiling).
                Algorithmic bytes per particle: P2G 100 B particle read + 16 B per touched grid node
                written; G2P 52 B read + 100 B written + 16 B per touched node read.  `traffic` = HBM bytes per
                launch from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE x2 + WRITE_SIZE, see
                DESIGN.md §6) / the same launch duration, or null when no PMC summary matches the workload.
                `measured_copy_GBs` = a plain float4 copy kernel of the library on this box (1 GiB, best of 5, bytes
                read + written), run after the timed region: the practical ceiling next to the nominal `peak`.
  cpu_baseline  the block-sorted, 8-colour, OpenMP restatement of the reference's optimised CPU path
                (oracle/mpm_oracle_opt.cpp, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (res, cube cells, material, kwargs)
    "c3": dict(res=256, cells=100, material="sand", desc="256^3 grid, 100^3 cells x 8 = 8M Drucker-Prager sand particles (BASELINE configs[2])"),
    "c2": dict(res=128, cells=50, material="jelly", desc="128^3 grid, 50^3 cells x 8 = 1M fixed-corotated jelly particles (BASELINE configs[1])"),
    # single-GPU only (the whole 64 M-particle problem fits one MI355X: 12 GB); needs ~48 GB of host memory to stage
    "c5": dict(res=512, cells=100, material="water+elastic", cpu_material="water", clusters=(78, 334),
               desc="512^3 sparse blocked grid, 8 clusters of 100^3 cells x 8 = 64M particles, 4 water + 4 Hencky-elastic (BASELINE configs[4])"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def build_sim(tm, cfg, device):
    res, cells = cfg["res"], cfg["cells"]
    lo = res // 2 - cells // 2
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=1.0 / res, base_delta_t=1e-4,
                                                       gravity=(0, -10, 0), device=device))
    sim.set_levelset(tm.mpm.LevelSet(friction=-1.0).add_plane((0, 1, 0), d=-0.1))  # sticky floor y = 0.1
    if "clusters" in cfg:  # C5: one cube per corner of a 2x2x2 arrangement, materials alternating
        k = 0
        for ox in cfg["clusters"]:
            for oy in cfg["clusters"]:
                for oz in cfg["clusters"]:
                    sim.add_particles(dict(type="water" if k % 2 == 0 else "elastic", cube_lo=(ox, oy, oz), cube_cells=cells))
                    k += 1
        return sim
    sim.add_particles(dict(type=cfg["material"], cube=(lo, lo + cells)))
    return sim


def cpu_baseline(cfg, budget_s=20.0):
    """restated reference algorithm (CPU) on a bounded sample: same grid/material/ppc, a smaller cube.
    OpenMP thread count: the best of a short sweep (oversubscribing a 512k-particle sample with every hardware
    thread of a 256-thread host is several times SLOWER than 32 threads); `cores` is the count actually used."""
    from oracle import oracle as orc
    from taichi_mpm_amd.mpm import lattice_cube
    res = cfg["res"]
    dx = 1.0 / res
    cells = 48 if res >= 256 else 32
    lo = res // 2 - cells // 2
    x = lattice_cube(lo, lo + cells, dx)
    vol = dx ** 3 / 8
    gp, t = orc.group_params(cfg.get("cpu_material", cfg["material"]), 400.0 * vol, vol)
    aux = np.full(len(x), orc.initial_aux(cfg.get("cpu_material", cfg["material"])), np.float32)
    s = orc.State(x, None, None, None, aux, None, gp[None], np.array([t], np.int32))
    ocfg = orc.make_config(res, dx, 1e-4, planes=[(0, 1, 0, -0.1)], friction=-1.0)
    n = len(x)
    hw = os.cpu_count() or 1
    orc.opt_run(ocfg, s, 1, min(hw, 16))  # warm-up (page faults, first sort)
    sweep = {}
    for th in sorted({1, 8, 16, 32, 64, hw}):
        if th > hw:
            continue
        sec, _ = orc.opt_run(ocfg, s, 1, th)
        sweep[th] = n / sec
        if sec > budget_s / 4:
            break
    threads = max(sweep, key=sweep.get)
    steps, total, phases = 0, 0.0, np.zeros(4)
    t0 = time.time()
    while total < budget_s / 2 and steps < 40 and time.time() - t0 < 2 * budget_s:
        sec, ph = orc.opt_run(ocfg, s, 2, threads)
        total += sec; steps += 2; phases += np.array(ph)
    return {"value": n * steps / total, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": "%d^3 cells x 8 = %d %s particles on the %d^3 grid, %d substeps, %d OpenMP threads (best of a "
                      "sweep on a %d-thread host); block-sorted 8-colour restatement of rasterize_optimized/"
                      "resample_optimized, not the reference binary" % (cells, n, cfg.get("cpu_material", cfg["material"]), res, steps, threads, hw),
            "thread_sweep_particle_steps_per_s": {str(k): v for k, v in sweep.items()},
            "p2g_ns_per_particle": 1e9 * phases[1] / (n * steps), "g2p_ns_per_particle": 1e9 * phases[3] / (n * steps),
            "sort_ns_per_particle": 1e9 * phases[0] / (n * steps)}


def pmc_traffic(config_name, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this workload (profiles/), or None"""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % config_name)
    try:
        with open(path) as f:
            d = json.load(f)
        return float(d["kernels"][kernel]["hbm_bytes_per_launch"]), d.get("source")
    except Exception:
        return None, None


def virtual_run(tm, cfg, args):
    import torch

    from taichi_mpm_amd import tiled
    from taichi_mpm_amd.mpm import F_ID, lattice_cube
    K = args.virtual
    res, cells = cfg["res"], cfg["cells"]
    dx = 1.0 / res
    lo = res // 2 - cells // 2
    x = lattice_cube(lo, lo + cells, dx)
    part = tiled.Partition.balanced((res,) * 3, K, x, dx, margin=4)
    owner = part.rank_of_cells(tiled.base_cells(x, dx))
    engines = []
    for r in range(K):
        mine = np.nonzero(owner == r)[0]
        sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=1e-4, gravity=(0, -10, 0),
                                                           max_particles=int(len(mine) * 1.5) + (1 << 16)))
        sim.set_levelset(tm.mpm.LevelSet(friction=-1.0).add_plane((0, 1, 0), d=-0.1))
        sim.add_particles(dict(type=cfg["material"], positions=x[mine]))
        sim.upload(F_ID, mine.astype(np.int32))
        engines.append(tiled.HipEngine(sim, 0))
    job = tiled.VirtualTiledJob(engines, part, overlap=os.environ.get("MPMHIP_TILE_OVERLAP", "1") != "0")
    job.run(args.warmup)
    for e in engines:  # level 1 = full phase table (substeps then run unsplit); 2 = only G2P bracketed
        e.sim.set_profiling(int(os.environ.get("MPMHIP_VIRTUAL_PROFILE", "1")))
        e.sim.profile(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.run(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    profs = [e.sim.profile() for e in engines]
    per_rank = [{k: v / max(p["substeps"], 1) for k, v in p["phases"].items()} for p in profs]
    return ({"diagnostic": "virtual ranks on one GPU", "K": K, "dims": part.dims, "cuts": part.cuts,
                      "particles_per_rank": [p["particles"] for p in profs], "active_blocks": [p["active_blocks"] for p in profs],
                      "halo_floats_per_rank": [r.plan.total for r in job.ranks],
                      "ms_per_step_all_ranks_serial": 1e3 * el / args.steps,
                      "per_rank_compute_ms": [sum(v for k, v in pr.items() if k != "exchange") for pr in per_rank],
                      "rank0_phases_ms": per_rank[0], "migrated": [r.migrated_out for r in job.ranks]})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--virtual", type=int, default=0, metavar="K",
                    help="diagnostic, not the metric: run the K-brick tiled job as K ctx on ONE GPU (exchanges are local "
                         "copies) and print per-rank phase times = the per-GPU compute of a K-GPU run without the wire")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): native libraries print banners to fd 1 (RCCL prints its version
    # block when the first communicator is created), so fd 1 points at stderr until the result is ready
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist

    import taichi_mpm_amd as tm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # test hook: MPMHIP_BENCH_BACKEND=gloo runs the N > 1 path with all ranks sharing the visible GPU(s) and the
    # device buffers staged through gloo (RCCL refuses two ranks on one GPU) — everything but the wire
    staged = world > 1 and os.environ.get("MPMHIP_BENCH_BACKEND", "nccl") == "gloo"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    force_tiled = world == 1 and os.environ.get("MPMHIP_FORCE_TILED") == "1"  # test hook: TiledJob over RCCL, 1 rank
    data_group, wire = None, None
    if world > 1 or force_tiled:
        # control plane (barriers, the two scalar reductions below, agreement on the transport) = gloo, the default
        # group; data plane (halo all-sum, migration) = an RCCL group over xGMI with device buffers
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        wire = "gloo, staged through host memory (test hook)"
        if not staged:
            # probe the transport with the two collectives the job uses; every rank must see it work, else all ranks
            # stage the same buffers through gloo (slower wire, same kernels, same results) rather than abort
            ok, why = 1, ""
            try:
                data_group = dist.new_group(backend="nccl")
                dev = torch.device("cuda", local_rank)
                a = torch.full((world,), float(rank), device=dev)
                b = torch.empty_like(a)
                dist.all_to_all_single(b, a, [1] * world, [1] * world, group=data_group)
                g = torch.empty(world, device=dev)
                dist.all_gather_into_tensor(g, a[:1], group=data_group)
                torch.cuda.synchronize()
                ok = int(bool((b.cpu() == torch.arange(world, dtype=b.dtype)).all())
                         and bool((g.cpu() == torch.arange(world, dtype=g.dtype)).all()))
                why = "" if ok else "RCCL probe returned wrong data"
            except Exception as e:
                ok, why = 0, repr(e)
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                wire = "RCCL (nccl backend), device buffers"
            else:
                print("bench.py[rank %d]: RCCL transport unavailable (%s); staging the exchange through gloo" % (rank, why or "failed on another rank"),
                      file=sys.stderr)
                staged, data_group = True, None
                wire = "gloo, staged through host memory (RCCL probe failed)"

    cfg = CONFIGS[args.config]
    if "clusters" in cfg and (world > 1 or args.virtual > 1):
        raise SystemExit("--config %s is a single-GPU workload in this build" % args.config)
    from taichi_mpm_amd import tiling
    if args.virtual > 1:
        return emit(virtual_run(tm, cfg, args))
    if world > 1 or force_tiled:
        from taichi_mpm_amd import tiled
        comm = (tiled.StagedDistComm(dist) if staged else
                tiled.DistComm(dist, torch.device("cuda", local_rank), data_group))
        job = tiled.make_tiled_job(tm, cfg, rank, world, local_rank, comm=comm)
    else:
        job = tiling.make_job(tm, cfg, rank, world, local_rank, build_sim)
    n_local = job.num_particles()
    job.run(args.warmup)
    job.synchronize()
    # untimed pass, every phase bracketed: phase table + which transfer kernel dominates
    job.set_profiling(1)
    job.run(min(10, max(args.steps, 1)))
    phase_prof = job.profile()
    pms = {k: v / max(phase_prof["substeps"], 1) for k, v in phase_prof["phases"].items()}
    dom = "g2p" if pms["g2p"] >= pms["p2g"] else "p2g"
    job.set_profiling(2 if dom == "g2p" else 3)  # timed region: only the dominant kernel is bracketed

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    job.run(args.steps)
    job.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nt = torch.tensor([n_local], dtype=torch.float64)
        dist.all_reduce(nt, op=dist.ReduceOp.SUM)
        n_total = int(nt.item())
    else:
        n_total = n_local
    prof = job.profile()
    copy_gbs = None
    if world == 1 and not force_tiled:  # what a plain streaming copy reaches on THIS box, measured after the timed region
        try:
            copy_gbs = job.sim.copy_bandwidth(1 << 30, 5)
        except Exception as e:
            print("bench.py: copy bandwidth probe failed: %r" % (e,), file=sys.stderr)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms = dict(pms)  # phase table from the untimed pass ...
    ms[dom] = prof["phases"][dom] / max(prof["substeps"], 1)  # ... dominant kernel from the timed region itself
    n_per_gpu = prof["particles"]
    nodes = prof["active_blocks"] * 64.0  # touched 4^3 blocks x 64 nodes (upper bound of touched nodes)
    per_launch = {"p2g": n_per_gpu * 100.0 + nodes * 16.0, "g2p": n_per_gpu * 152.0 + nodes * 16.0}
    achieved = per_launch[dom] / (ms[dom] * 1e-3) / 1e9
    value = n_total * args.steps / elapsed
    whole_step_bytes = n_per_gpu * 252.0 + nodes * 16.0 * 5
    tbytes, tsrc = pmc_traffic(args.config, "k_" + dom) if world == 1 else (None, None)
    out = {
        "metric": "particle-steps/sec (P2G+grid+G2P), 256^3 grid 8M particles; %HBM roofline",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": job.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["desc"], "particles": n_total, "dt": 1e-4, "parallelism": job.parallelism, "wire": wire,
                   "timed": "full substep: sort+reorder, P2G, grid normalise+boundary, G2P, boundary cleanup"},
        "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "measured_copy_GBs": copy_gbs,  # plain float4 copy kernel on this box (read + written bytes)
                     "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                     "traffic": (tbytes / (ms[dom] * 1e-3) / 1e9) if tbytes else None,
                     "traffic_bytes_per_launch": tbytes, "traffic_source": tsrc,
                     "algorithmic_bytes_per_launch": per_launch[dom], "avg_launch_ms": ms[dom]},
        "phases_ms_per_step": ms,
        "p2g_plus_g2p_particle_steps_per_s": n_per_gpu / ((ms["p2g"] + ms["g2p"]) * 1e-3),
        "p2g_plus_g2p_hbm_frac_algorithmic": (per_launch["p2g"] + per_launch["g2p"]) / ((ms["p2g"] + ms["g2p"]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "whole_step_hbm_frac_algorithmic": whole_step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg)
        except Exception as e:  # the baseline is a reported extra: never lose the GPU line because of it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    emit(out)
    if world > 1 or force_tiled:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
