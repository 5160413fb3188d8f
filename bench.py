#!/usr/bin/env python
"""bench.py — particle-steps/sec of the MLS-MPM substep (sort + P2G + grid + G2P) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched under torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one substep (src/mpm.cpp:452-575) of the workload
BASELINE.json's metric is quoted on: config C3 = 256^3 grid, 100^3 cells x 8 = 8 000 000 Drucker-Prager sand
particles, fp32, inputs resident in HBM before the timed region.

Sequence:  W warm-up substeps (untimed) | a short untimed pass with every phase bracketed by hipEvents (phase
table, picks the dominant kernel) | barrier | K timed substeps, only the dominant kernel bracketed and only in every 4th of
them (an event record idles the GPU for ~5 us: bracketing all phases would add ~20 us per substep, the dominant kernel in
every substep 2 % of a C3 substep; roofline.launches_timed says how many launches the mean is over) | barrier.

N > 1 (bricks + halo exchange): before that sequence the boundary / interior split of the substep (it hides
the exchange behind the interior kernels at the price of three more launches) is timed on and off over a dozen untimed
substeps, max over ranks, and the faster way is kept: config.overlap_split (MPMHIP_TILE_OVERLAP=0|1 pins it).

Extra objects on the line:
  roofline      dominant kernel (the slower of k_p2g / k_g2p): algorithmic bytes per launch / its average launch
                duration over the K timed substeps, hipEvents recorded on the ctx stream (mpmhip_set_profiling).
                Algorithmic bytes per particle: P2G 100 B particle read + 16 B per touched grid node
                written; G2P 52 B read + 100 B written + 16 B per touched node read.  `traffic` = HBM bytes per
                launch from the committed rocprofv3 PMC passes (profiles/traffic_<config>[_evolved].json:
                FETCH_SIZE x2 + WRITE_SIZE, the factor calibrated for 64-byte gathers) / the same
                launch duration, or null when no PMC summary matches the workload.
                `measured_copy_GBs` = a plain float4 copy kernel of the library on this box (1 GiB, best of 5, bytes
                read + written), run after the timed region: the practical ceiling next to the nominal `peak`.
                `p2g_plus_g2p` (inside roofline): both transfer kernels together, the quantity the 40 % target of north_star is
                set on — algorithmic bytes of SURVEY section 8(d) (252 B per particle-step + the touched nodes) over the sum of the
                two kernels' launch times, and the PMC traffic of both from the same committed profile.
  evolved       (+ `cond_F`: census of cond(F) over the evolved state's particles — max, quantiles, the share the eigen-solve refines)
                the same measurement (ms_per_step, phases, roofline) on the same ctx after the seeded block has
                fallen onto the floor and EVOLVE_AFTER_IMPACT further substeps have run: uneven cells, active
                return map.  `value` stays the lattice the reference's benchmark seeds (config.state says so);
                --state evolved makes the evolved state the main measurement (profiling runs), --no-evolved skips it.
  deterministic the same ctx with mpmhip_config.deterministic on (every cell's particles in creation-id order behind the sort: bitwise
                reproducible runs): ms_per_step over the same K substeps and what the mode costs per substep.
  virtual       (N = 1, default workload) the multi-GPU evidence one GPU can give, timed by whoever runs this file: the K bricks of the
                workload as K ctx on this GPU for K = 8 (then 2 and 4 while the whole block stays below VIRTUAL_BUDGET_S), the library's
                own loop and halo exchange over the local wire — per-rank substep (what one rank of a K-GPU job computes per substep,
                before the wire), per-rank phase table, halo bytes, and `x_over_one_gpu` = ms_per_step of this line / per-rank substep:
                the upper bound of the strong-scaling factor at K GPUs.  --no-virtual skips it.
  c5            (N = 1, default workload) BASELINE configs[4] — 512^3 sparse grid, 64 M mixed particles in 8 clusters, the configuration
                the >= 6x at 8 GPUs is claimed for — on this GPU: the whole problem in one ctx (ms_per_step, value), then as 8 ctx of one
                cluster each over the local wire (`virtual8`: per-rank substep and x_over_one_gpu, the same proxy as `virtual`).
                --no-c5 skips it.
  cpu_baseline  kind "reference": the reference's own solver, compiled from its sources in place
                (oracle/_ref/libmpm_ref.so, built by `make -C oracle ref_mpm` where /root/reference exists; the
                built library travels), timed on this box's host cores: thread sweep + threads=1 row on the
                reference's benchmark=125 lattice (~1 M particles), then the full workload at the best thread count.
                Fallback when that library is absent: the OpenMP restatement (oracle/mpm_oracle_opt.cpp, kind "port").
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
    # dt: half of the other configs' — the same Courant number at half the cell size.  (At 1e-4 the water clusters
    # (k = 1e4, gamma = 7: c = 13 m/s at rest, c dt / dx = 0.68, stiffening under compression) blow up ~420 substeps
    # after they hit the floor and the run ends with 27 k of 64 M particles: measured, with this build and the one before.)
    "c5": dict(res=512, cells=100, material="water+elastic", cpu_material="water", clusters=(78, 334), dt=5e-5,
               desc="512^3 sparse blocked grid, 8 clusters of 100^3 cells x 8 = 64M particles, 4 water + 4 Hencky-elastic (BASELINE configs[4])"),
}
PROFILE_EVERY = int(os.environ.get("MPMHIP_BENCH_PROFILE_EVERY", "4"))  # the dominant kernel is timed in every 4th substep of the timed region
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def build_sim(tm, cfg, device):
    res, cells = cfg["res"], cfg["cells"]
    lo = res // 2 - cells // 2
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=1.0 / res, base_delta_t=cfg.get("dt", 1e-4),
                                                       gravity=(0, -10, 0), device=device,
                                                       keep_apic_b=bool(cfg.get("keep_apic_b", False))))
    sim.set_levelset(tm.mpm.LevelSet(friction=-1.0).add_plane((0, 1, 0), d=-0.1))  # sticky floor y = 0.1
    from taichi_mpm_amd.tiled import scene_groups
    for typ, lo, n in scene_groups(cfg):  # C5: one cube per corner of a 2x2x2 arrangement, materials alternating
        sim.add_particles(dict(type=typ, cube_lo=lo, cube_cells=n))
    return sim


def substeps_to_impact(cfg, g=10.0, floor=0.1):
    """substeps of free fall until the seeded cube touches the floor plane y = 0.1"""
    res, cells = cfg["res"], cfg["cells"]
    lo = cfg["clusters"][0] if "clusters" in cfg else res // 2 - cells // 2
    h = (lo + 0.25) / res - floor
    return int(np.sqrt(2.0 * h / g) / cfg.get("dt", 1e-4))


VIRTUAL_BUDGET_S = 60.0  # N = 1: the `virtual` block of the line takes K = 8 first, then 2 and 4 while the block stays below this
EVOLVE_AFTER_IMPACT = 400  # substeps run after the block has touched the floor before the `evolved` state is timed


def evolve_to_impact(sim, cfg):
    """run the scene on the device until EVOLVE_AFTER_IMPACT substeps after the block hit the floor: uneven cells, decayed
    slot order, active return map — the state the `evolved` object of the JSON line is timed on.  Returns the number of
    substeps run after the impact."""
    n = substeps_to_impact(cfg) + EVOLVE_AFTER_IMPACT
    sim.run_substeps(n)
    sim.synchronize()
    return EVOLVE_AFTER_IMPACT


def cpu_baseline_reference(cfg, budget_s=25.0):
    """the REFERENCE's own solver (oracle/_ref/libmpm_ref.so: /root/reference/src/{mpm,transfer,particles}.cpp compiled
    where they lie against oracle/taichi_shim) on this box's host cores.  Sample: the reference's own benchmark
    generator (src/mpm.cpp:149-186, benchmark=125: ~1 M particles) on the workload's grid and material for the thread
    sweep and the threads=1 row (what scripts/benchmark/benchmark_3d.py:17 uses), then the full workload (8 M) at the
    best thread count when the budget allows.  svd / polar_decomp are the shim's (double-precision Jacobi), slower
    than the legacy taichi core's fp32 routines: the stress-bound phases are pessimistic."""
    from oracle import refmpm as ref
    from taichi_mpm_amd.mpm import lattice_cube
    res = cfg["res"]
    dx = 1.0 / res
    mat = cfg.get("cpu_material", cfg["material"])
    hw = os.cpu_count() or 1

    def run(sim, steps):
        ref.profile(reset=True)
        t0 = time.perf_counter()
        sim.substep(steps)
        sec = time.perf_counter() - t0
        return sec, ref.profile()

    def small(threads):
        ref.set_threads(threads)
        sim = ref.Sim(res, dx, cfg.get("dt", 1e-4), shapes=[(0, 0, 0, 1, 0, -0.1)], friction=-1.0)
        sim.add_benchmark(mat, 125)
        return sim
    sweep, n_small = {}, 0
    t_start = time.time()
    for th in sorted({1, 8, 16, 32, 64, 128} | {hw}):
        if th > hw:
            continue
        sim = small(th)
        n_small = sim.num_particles()
        sim.substep(1)  # warm-up: first sort + physical reorder, page faults
        sec, prof = run(sim, 3)
        sim.close()
        sweep[th] = dict(value=n_small * 3 / sec, p2g_ns=1e9 * prof.get("P2G optimized", 0) / (3 * n_small),
                         g2p_ns=1e9 * prof.get("G2P optimized", 0) / (3 * n_small),
                         sort_ns=1e9 * prof.get("sort_particles_and_populate_grid", 0) / (3 * n_small))
        if time.time() - t_start > budget_s * 0.5:
            break
    best = max(sweep, key=lambda k: sweep[k]["value"])
    out = {"value": sweep[best]["value"], "unit": "particle-steps/s", "cores": best, "kind": "reference",
           "sample": "%d %s particles (the reference's benchmark=125 generator, src/mpm.cpp:149-186) on the %d^3 grid, "
                     "3 substeps after a warm-up, %d OpenMP threads = best of the sweep on a %d-thread host; the reference's "
                     "own sources (mpm.cpp, transfer.cpp, particles.cpp) built against oracle/taichi_shim: svd / polar_decomp "
                     "are the shim's double-precision Jacobi" % (n_small, mat, res, best, hw),
           "threads_1_particle_steps_per_s": sweep.get(1, {}).get("value"),
           "thread_sweep": {str(k): v for k, v in sweep.items()},
           "p2g_ns_per_particle": sweep[best]["p2g_ns"], "g2p_ns_per_particle": sweep[best]["g2p_ns"],
           "sort_ns_per_particle": sweep[best]["sort_ns"]}
    # the full workload at the best thread count, if ~5 substeps fit the rest of the budget
    n_full = cfg["cells"] ** 3 * 8 * (8 if "clusters" in cfg else 1)
    est = n_full / sweep[best]["value"]
    if "clusters" not in cfg and 5 * est < budget_s:
        try:
            ref.set_threads(best)
            lo = res // 2 - cfg["cells"] // 2
            x = lattice_cube(lo, lo + cfg["cells"], dx)
            vol = dx ** 3 / 8
            sim = ref.Sim(res, dx, cfg.get("dt", 1e-4), shapes=[(0, 0, 0, 1, 0, -0.1)], friction=-1.0)
            sim.add_particles(mat, 400.0 * vol, vol, x)
            sim.substep(1)
            sec, prof = run(sim, 3)
            sim.close()
            out.update({"value": len(x) * 3 / sec, "full_workload": True,
                        "sample": "the full workload: %d %s particles on the %d^3 grid, 3 substeps after a warm-up, %d OpenMP "
                                  "threads (best of a sweep on a %d-thread host); the reference's own sources built against "
                                  "oracle/taichi_shim: svd / polar_decomp are the shim's double-precision Jacobi"
                                  % (len(x), mat, res, best, hw),
                        "p2g_ns_per_particle": 1e9 * prof.get("P2G optimized", 0) / (3 * len(x)),
                        "g2p_ns_per_particle": 1e9 * prof.get("G2P optimized", 0) / (3 * len(x)),
                        "sort_ns_per_particle": 1e9 * prof.get("sort_particles_and_populate_grid", 0) / (3 * len(x))})
        except Exception as e:
            out["full_workload_error"] = repr(e)
    return out


class SingleJob:
    """the one-GPU job: the whole problem in one ctx"""
    scaling = "strong"
    parallelism = "1 GPU"

    def __init__(self, sim):
        self.sim = sim
        self.substeps = 0
        sim._ensure_ctx()

    def num_particles(self):
        return self.sim.get_num_particles()

    def run(self, n):
        self.sim.run_substeps(n)
        self.substeps += n

    def synchronize(self):
        self.sim.synchronize()

    def set_profiling(self, level, every=1):
        self.sim.set_profiling(level, every)
        self.sim.profile(reset=True)

    def profile(self):
        return self.sim.profile()

    def kernel_name(self, phase):
        return self.sim.g2p_kernel() if phase == "g2p" else "k_" + phase


def cpu_baseline(cfg, budget_s=20.0):
    """(fallback when oracle/_ref/libmpm_ref.so is absent) restated reference algorithm (CPU) on a bounded sample: same grid/material/ppc, a smaller cube.
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
    ocfg = orc.make_config(res, dx, cfg.get("dt", 1e-4), planes=[(0, 1, 0, -0.1)], friction=-1.0)
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


_LOADED_HASHES = None


def loaded_kernel_hashes():
    """{kernel base name: code hash} of the library this process runs (profiles/kernel_diff.py: sha256 over the disassembly of all
    instantiations), or a string saying why they cannot be had"""
    global _LOADED_HASHES
    if _LOADED_HASHES is None:
        try:
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            import kernel_diff
            from taichi_mpm_amd import _lib
            _LOADED_HASHES = kernel_diff.kernel_hashes(_lib.lib_path())
        except Exception as e:
            _LOADED_HASHES = "code hash unavailable: %r" % (e,)
    return _LOADED_HASHES


def pmc_traffic(config_name, kernel, check=True):
    """(HBM bytes per launch of `kernel` from the committed PMC summary of this workload (profiles/), its source text, a note).
    The summary records the code hash of every kernel it was taken on (make_traffic.py --lib): when the loaded library's `kernel`
    differs — a kernel edit without a new PMC pass — the bytes are withheld (None) and the note says so: a stale ratio must not
    look measured."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % config_name)
    try:
        with open(path) as f:
            d = json.load(f)
        tbytes, src = float(d["kernels"][kernel]["hbm_bytes_per_launch"]), d.get("source")
    except Exception:
        return None, None, None
    if not check:
        return tbytes, src, None
    want = (d.get("code") or {}).get("kernel_hashes", {}).get(kernel)
    if want is None:
        return None, src, "withheld: the PMC summary records no code hash for %s" % kernel
    have = loaded_kernel_hashes()
    if isinstance(have, str):
        return None, src, "withheld: " + have
    if have.get(kernel) != want:
        return None, src, ("withheld: %s of the loaded library (code %s) is not the kernel the PMC passes ran on (code %s, commit %s): "
                           "run profiles/run_profile.sh again" % (kernel, have.get(kernel), want, (d.get("code") or {}).get("commit")))
    return tbytes, src, "code %s = the PMC build (commit %s)" % (want, (d.get("code") or {}).get("commit"))


def virtual_run(tm, cfg, args, K=None, overlap=None):
    """--virtual K: the K-brick job as K ctx on ONE GPU, on the library's own data plane (MPMHIP_WIRE_LOCAL: peer writes with
    plain pointers; loop, exchange and migration inside mpmhip_tiled_advance_group).  All ctx share one stream, so the ranks'
    kernels run one after another and a rank's event-bracketed parts are the per-GPU compute of a K-GPU run without the wire."""
    import torch

    from taichi_mpm_amd import tiled
    K = K or args.virtual
    part = tiled.scene_partition(cfg, K, margin=int(os.environ.get("MPMHIP_TILE_MARGIN", 4)))
    engines = []
    for r in range(K):
        sim, _ = tiled.build_rank_sim(tm, cfg, part, r, 0)
        engines.append(tiled.HipEngine(sim, 0))
    if overlap is None:
        overlap = os.environ.get("MPMHIP_TILE_OVERLAP", "1") != "0"
    python_loop = os.environ.get("MPMHIP_VIRTUAL_PYTHON") == "1"  # the round-3 path: Python loop, exchanges as torch copies
    job = tiled.VirtualTiledJob(engines, part, overlap=overlap) if python_loop else tiled.NativeVirtualJob(engines, part, overlap=overlap)
    job.run(args.warmup)

    def measured(level, steps):
        for e in engines:
            e.sim.set_profiling(level)
            e.sim.profile(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.run(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        profs = [e.sim.profile() for e in engines]
        return el, profs, [{k: v / max(p["substeps"], 1) for k, v in p["phases"].items()} for p in profs]

    # wall time of all ranks' substeps with no event anywhere, then level 4 (begin / interior / end of a substep bracketed as
    # wholes — every hipEventRecord idles the GPU for ~5 us), then the per-phase table (level 1, six records per substep,
    # overlap split off)
    el0, profs, _ = measured(0, args.steps)
    el, _, per_rank = measured(int(os.environ.get("MPMHIP_VIRTUAL_PROFILE", "4")), args.steps)
    _, _, per_phase = measured(1, max(args.steps // 2, 4))
    n_total = sum(p["particles"] for p in profs)
    if python_loop:
        halo, migrated = [r.plan.total // 4 for r in job.ranks], [r.migrated_out for r in job.ranks]
    else:
        st = job.state()
        halo, migrated = [t["halo_nodes"] for t in st], [t["migrated_out"] for t in st]
    for e in engines:  # (the next K of the default line's block builds its own ranks)
        e.sim.close()
    return ({"diagnostic": "virtual ranks on one GPU", "K": K, "dims": part.dims, "cuts": part.cuts, "workload": cfg["desc"],
             "loop": "python (VirtualTiledJob)" if python_loop else "native (mpmhip_tiled_advance_group, MPMHIP_WIRE_LOCAL)",
             "overlap_split": overlap, "particles": n_total,
             "particles_per_rank": [p["particles"] for p in profs], "active_blocks": [p["active_blocks"] for p in profs],
             "halo_nodes_per_rank": halo, "halo_bytes_per_rank": [16 * h for h in halo],
             "ms_per_step_all_ranks_serial_no_events": 1e3 * el0 / args.steps,
             "per_rank_ms_serial_no_events": 1e3 * el0 / args.steps / K,
             "ms_per_step_all_ranks_serial": 1e3 * el / args.steps,
             "per_rank_compute_ms": [sum(v for k, v in pr.items() if k != "exchange") for pr in per_rank],
             "per_rank_compute_ms_per_phase_events": [sum(v for k, v in pr.items() if k != "exchange") for pr in per_phase],
             "rank0_phases_ms": per_phase[0], "migrated": migrated})


def c5_block(tm, args, steps=10, warmup=3):
    """BASELINE configs[4] on one GPU: the whole 64 M problem in one ctx, then its 8 clusters as 8 ctx on this GPU (tiled.scene_partition:
    one cluster per rank) — time / 8 = what one rank of the 8-GPU job computes per substep before the wire"""
    cfg = dict(CONFIGS["c5"])
    t_block = time.time()
    sim = build_sim(tm, cfg, 0)
    job = SingleJob(sim)
    n = job.num_particles()
    job.run(warmup)
    job.synchronize()
    t0 = time.perf_counter()
    job.run(steps)
    job.synchronize()
    el = time.perf_counter() - t0
    prof_blocks = None
    try:
        job.set_profiling(1, 1)
        job.run(4)
        job.synchronize()
        p = job.profile()
        prof_blocks = p.get("active_blocks")
        phases = {k: v / max(p["substeps"], 1) for k, v in p["phases"].items()}
    except Exception:
        phases = None
    sim.close()
    one = 1e3 * el / steps
    out = {"workload": cfg["desc"], "particles": n, "dt": cfg["dt"], "steps": steps, "warmup": warmup, "ms_per_step": one,
           "value": n * steps / el, "unit": "particle-steps/s", "active_blocks": prof_blocks, "phases_ms_per_step": phases,
           "whole_step_hbm_frac_algorithmic": (n * 252.0 + (prof_blocks or 0) * 64 * 80.0) / (el / steps) / 1e9 / HBM_PEAK_GBS}
    a = argparse.Namespace(**vars(args))
    a.steps, a.warmup = steps, warmup
    try:
        v = virtual_run(tm, cfg, a, K=8, overlap=False)
        out["virtual8"] = {"per_rank_ms": v["per_rank_ms_serial_no_events"], "x_over_one_gpu": one / v["per_rank_ms_serial_no_events"],
                           "particles_per_rank": v["particles_per_rank"], "halo_bytes_per_rank": v["halo_bytes_per_rank"],
                           "dims": v["dims"], "migrated": v["migrated"],
                           "what": "the 8 clusters as 8 ctx on THIS GPU over the local wire: per_rank_ms = all ranks' substeps back to back / 8 = "
                                   "the per-GPU compute of the 8-GPU job before the wire; x_over_one_gpu = upper bound of its factor over this GPU alone"}
    except Exception as e:
        out["virtual8"] = {"error": repr(e)}
    out["seconds"] = time.time() - t_block
    return out


class Watchdog:
    """A multi-rank run that hangs (a collective one rank never enters, a transport that stalls on first contact) must cost
    seconds, not the lease: a timer that prints where the run was and ends the PROCESS (os._exit: a thread stuck inside a
    collective cannot be interrupted).  `phase(name, seconds)` re-arms it; every rank runs its own."""

    def __init__(self, rank):
        import threading
        self.rank, self.t, self.name, self.threading = rank, None, "start", threading

    def _fire(self):
        sys.stderr.write("bench.py[rank %d]: no progress in phase '%s' within its deadline — giving up (exit 3)\n" % (self.rank, self.name))
        sys.stderr.flush()
        os._exit(3)

    def phase(self, name, seconds):
        if self.t is not None:
            self.t.cancel()
        self.name = name
        self.t = self.threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.t.start()

    def stop(self):
        if self.t is not None:
            self.t.cancel()


def rccl_probe_main():
    """`bench.py --probe-rccl` (started by every rank as a CHILD process, own rendezvous port): exactly what the job will do with
    the wire — the library dlopens librccl, the ncclUniqueId travels over gloo, ncclCommInitRank, then the library's loopback
    check (an all-gather and a grouped ncclSend / ncclRecv ring on device buffers, verified on the host).  A transport that hangs
    or aborts takes the child down, not the job: the parent waits with a timeout."""
    import ctypes as C
    import datetime

    import torch
    import torch.distributed as dist

    import taichi_mpm_amd as tm
    from taichi_mpm_amd import _lib
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=45))
    sim = tm.create_simulation3("mpm").initialize(dict(res=(16,) * 3, device=local, max_particles=1024))
    sim._ensure_ctx()
    L = sim._L
    ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        if L.mpmhip_comm_unique_id(buf) != 0:
            print("RCCL_PROBE_FAILED " + L.mpmhip_last_error(None).decode(), flush=True)
            os._exit(4)
        ident = torch.tensor(list(buf), dtype=torch.uint8)
    dist.broadcast(ident, src=0)
    buf = (C.c_uint8 * _lib.COMM_ID_BYTES)(*ident.tolist())
    ok = L.mpmhip_comm_init(sim._ctx, buf, rank, world) == 0 and L.mpmhip_comm_selftest(sim._ctx) == 0
    if not ok:
        print("RCCL_PROBE_FAILED " + L.mpmhip_last_error(sim._ctx).decode(), flush=True)
        os._exit(4)
    L.mpmhip_comm_destroy(sim._ctx)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_PROBE_OK", flush=True)
    os._exit(0)


def probe_rccl_in_child(rank, world, local_rank, timeout_s=75.0):
    """(ok, why): run rccl_probe_main in a child of this rank; the children of all ranks rendezvous among themselves"""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + 17)
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank))
    env.pop("TORCHELASTIC_RUN_ID", None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-rccl"], env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return False, "the RCCL probe did not finish within %.0f s" % timeout_s
    if r.returncode == 0 and "RCCL_PROBE_OK" in r.stdout:
        return True, ""
    return False, "the RCCL probe exited with code %d: %s" % (r.returncode, (r.stdout or r.stderr).strip().splitlines()[-1:] or "")


def main():
    # (multi-process GPU work on this pool needs dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "--probe-rccl" in sys.argv:
        return rccl_probe_main()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cells", type=int, default=0, help="override the cube / cluster edge in cells (reduced runs for tests)")
    ap.add_argument("--no-evolved", action="store_true", help="skip the second measurement on the evolved state")
    ap.add_argument("--state", default="lattice", choices=["lattice", "evolved"],
                    help="state the main measurement is taken on: the freshly seeded lattice (the metric's configuration) or "
                         "the same scene %d substeps after the block hit the floor (for profiling runs)" % EVOLVE_AFTER_IMPACT)
    ap.add_argument("--wire", default=os.environ.get("MPMHIP_TILE_WIRE", "rccl"), choices=["rccl", "ipc", "both", "torch"],
                    help="N > 1: the data plane of the halo exchange and the migration.  rccl (default): ncclSend / ncclRecv groups "
                         "issued by libmpmhip itself; ipc: peer writes into IPC-mapped receive buffers, no collective; both: the line is "
                         "measured over rccl, then the SAME job is re-wired to ipc and measured again (`tiled.other_wire`); torch: the "
                         "round-3 path (torch.distributed all_to_all from a Python callback per substep)")
    ap.add_argument("--allow-staged", action="store_true",
                    help="N > 1: if the RCCL wire cannot be brought up, stage the exchange through gloo and host memory instead of "
                         "exiting with an error (such a line says so in config.wire and is NOT a scaling measurement)")
    ap.add_argument("--no-virtual", action="store_true", help="N = 1: skip the `virtual` block of the line (K bricks as K ctx on this GPU)")
    ap.add_argument("--no-c5", action="store_true", help="N = 1: skip the `c5` block of the line (BASELINE configs[4] in one ctx and as 8 virtual ranks)")
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
    staged = False
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    force_tiled = world == 1 and os.environ.get("MPMHIP_FORCE_TILED") == "1"  # test hook: TiledJob over RCCL, 1 rank
    data_group, wire, native_wire = None, None, None
    shared = os.environ.get("MPMHIP_BENCH_BACKEND", "")  # test hooks: "gloo" / "ipc" = the ranks share the visible GPU(s)
    dog = Watchdog(rank)
    if world > 1 or force_tiled:
        import datetime
        # control plane (barriers, the scalar reductions below, 64 / 128 bytes of wire set-up) = gloo, the default group;
        # data plane (halo sums, migration) = the library's own: RCCL send / receive groups or IPC peer writes over xGMI
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dog.phase("gloo rendezvous", 180)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=150))
        if shared == "gloo":
            staged, wire = True, "gloo, staged through host memory (test hook)"
        elif shared == "ipc":
            native_wire, wire = "ipc", "IPC peer writes (test hook: the ranks share the visible GPUs), handles over gloo"
        else:
            ok, why = True, ""
            if args.wire in ("rccl", "both", "torch") and world > torch.cuda.device_count():
                # (one node by contract: ranks then share devices, which RCCL refuses — after a long rendezvous; say so at once)
                ok, why = False, "%d ranks on %d visible devices: RCCL does not take two ranks on one device" % (world, torch.cuda.device_count())
            elif args.wire in ("rccl", "both", "torch") and os.environ.get("MPMHIP_SKIP_RCCL_PROBE") != "1":
                # probe the transport in a CHILD process per rank (a hang or an abort inside RCCL then costs the child and a
                # bounded wait); every rank must see it work
                dog.phase("RCCL probe", 240)
                ok, why = probe_rccl_in_child(rank, world, local_rank)
                flag = torch.tensor([int(ok)])
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(int(flag.item()))
            if ok and args.wire == "torch":
                dog.phase("RCCL group", 120)
                data_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=90))
                wire = "RCCL through torch.distributed (all_to_all from a Python callback per substep)"
            elif ok:
                native_wire = "rccl" if args.wire == "both" else args.wire
                wire = {"rccl": "RCCL: ncclSend / ncclRecv groups issued by libmpmhip (no Python in the substep loop)",
                        "ipc": "IPC peer writes into mapped receive buffers + epoch flags (no collective, no Python in the substep loop)"}[native_wire]
            elif args.wire in ("rccl", "both") and os.environ.get("MPMHIP_NO_IPC_FALLBACK") != "1":
                # still a native data plane: the library's IPC peer writes need no collective library at all (handles over gloo)
                print("bench.py[rank %d]: the RCCL wire could not be brought up (%s); using the library's IPC wire" %
                      (rank, why or "failed on another rank"), file=sys.stderr, flush=True)
                native_wire = "ipc"
                wire = "IPC peer writes into mapped receive buffers + epoch flags (no collective, no Python in the substep loop; the RCCL probe failed)"
            else:
                msg = "bench.py[rank %d]: the RCCL wire could not be brought up (%s)" % (rank, why or "failed on another rank")
                if not args.allow_staged:
                    # a scaling curve measured over host-staged buffers is worthless: refuse rather than degrade silently
                    print(msg + "; refusing to fall back to a host-staged exchange (pass --allow-staged to run anyway)", file=sys.stderr, flush=True)
                    dog.stop()
                    dist.destroy_process_group()
                    sys.exit(5)
                print(msg + "; --allow-staged: staging the exchange through gloo", file=sys.stderr, flush=True)
                staged, wire = True, "gloo, staged through host memory (RCCL probe failed; --allow-staged)"

    cfg = dict(CONFIGS[args.config])
    if args.cells:  # reduced problem (tests, smoke runs): NOT the metric's configuration — the line says so
        cfg["cells"] = args.cells
        cfg["desc"] += " [REDUCED: cube edge %d cells]" % args.cells
    if args.virtual > 1:
        return emit(virtual_run(tm, cfg, args))
    dog.phase("scene set-up", 900)
    if world > 1 or force_tiled:
        from taichi_mpm_amd import tiled
        if native_wire:
            job = tiled.make_native_job(tm, cfg, rank, world, local_rank, wire=native_wire, dist=dist)
        else:
            comm = (tiled.StagedDistComm(dist) if staged else
                    tiled.DistComm(dist, torch.device("cuda", local_rank), data_group))
            job = tiled.make_tiled_job(tm, cfg, rank, world, local_rank, comm=comm)
    else:
        job = SingleJob(build_sim(tm, cfg, local_rank))
    n_local = job.num_particles()
    if world > 1 or force_tiled:
        print("bench.py[rank %d/%d]: device %d, wire: %s, %d particles on this rank" % (rank, world, local_rank, wire, n_local),
              file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(warmup, steps):
        """W untimed warm-up substeps | a short untimed pass with every phase bracketed (phase table, picks the
        dominant transfer kernel) | barrier | K timed substeps with only the dominant kernel bracketed | barrier.
        Returns (seconds, phase table in ms per substep with the dominant kernel taken from the timed region, which
        kernel dominates, profile of the timed region)."""
        job.run(warmup)
        job.synchronize()
        job.set_profiling(1)
        job.run(min(10, max(steps, 1)))
        phase_prof = job.profile()
        pms = {k: v / max(phase_prof["substeps"], 1) for k, v in phase_prof["phases"].items()}
        dom = "g2p" if pms["g2p"] >= pms["p2g"] else "p2g"
        # timed region: only the dominant kernel is bracketed, and only in every PROFILE_EVERY-th substep (a hipEventRecord idles
        # the GPU for ~5 us; two per substep would be 2 % of a C3 substep spent on the measurement itself)
        job.set_profiling(2 if dom == "g2p" else 3, every=PROFILE_EVERY if steps >= 4 * PROFILE_EVERY else 1)
        barrier()
        t0 = time.perf_counter()
        job.run(steps)
        job.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        prof = job.profile()
        job.set_profiling(0)
        ms = dict(pms)
        ms[dom] = prof["phases"][dom] / max(prof["substeps"], 1)
        return elapsed, ms, dom, prof

    def roofline_of(ms, dom, prof, traffic_tag):
        n_per_gpu = prof["particles"]
        nodes = prof["active_blocks"] * 64.0  # touched 4^3 blocks x 64 nodes (upper bound of touched nodes)
        per_launch = {"p2g": n_per_gpu * 100.0 + nodes * 16.0, "g2p": n_per_gpu * 152.0 + nodes * 16.0}
        achieved = per_launch[dom] / (ms[dom] * 1e-3) / 1e9
        kname = job.kernel_name(dom) if hasattr(job, "kernel_name") else "k_" + dom  # (k_g2p_packed for large one-material problems)
        tbytes, tsrc, tnote = pmc_traffic(traffic_tag, kname) if world == 1 else (None, None, None)
        roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (tbytes / (ms[dom] * 1e-3) / 1e9) if tbytes else None,
                "traffic_bytes_per_launch": tbytes, "traffic_source": tsrc, "traffic_check": tnote,
                "algorithmic_bytes_per_launch": per_launch[dom], "avg_launch_ms": ms[dom],
                "launches_timed": prof["substeps"]}  # (every PROFILE_EVERY-th substep of the timed region)
        both = (per_launch["p2g"] + per_launch["g2p"]) / ((ms["p2g"] + ms["g2p"]) * 1e-3) / 1e9 / HBM_PEAK_GBS
        # P2G + G2P together — the quantity north_star sets its 40 % target on.  Algorithmic bytes: the 252 B per particle-step
        # of SURVEY section 8(d) (P2G reads the 100 B state, G2P reads 52 B and writes 100 B) + 16 B per touched node written and
        # read; what the layout actually moves (G2P hands P2G a 64-byte record with the affine matrix precomputed) is `traffic`.
        # The kernel that is not the dominant one is timed in the untimed phase pass, not in the timed region.
        other = "p2g" if dom == "g2p" else "g2p"
        oname = job.kernel_name(other) if hasattr(job, "kernel_name") else "k_" + other
        t_other, _, _ = pmc_traffic(traffic_tag, oname) if world == 1 else (None, None, None)
        t_both = (tbytes + t_other) if (tbytes and t_other) else None
        dur = (ms["p2g"] + ms["g2p"]) * 1e-3
        roof["p2g_plus_g2p"] = {"kernels": sorted([kname, oname], reverse=True), "algorithmic_bytes_per_step": per_launch["p2g"] + per_launch["g2p"],
                                "avg_ms": ms["p2g"] + ms["g2p"], "achieved": (per_launch["p2g"] + per_launch["g2p"]) / dur / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": both,
                                "traffic_bytes_per_step": t_both, "traffic": (t_both / dur / 1e9) if t_both else None,
                                "traffic_source": tsrc}
        return roof, both, n_per_gpu, nodes

    if args.state == "evolved":  # make the evolved state the one that is measured (profiling runs)
        dog.phase("evolving to impact", 1200)
        job.run(substeps_to_impact(cfg) + EVOLVE_AFTER_IMPACT)  # (through the job: a tiled ctx cannot be stepped on its own)
        job.synchronize()
    overlap_note = None
    if (world > 1 or force_tiled) and os.environ.get("MPMHIP_TILE_OVERLAP", "auto") == "auto":
        # The boundary / interior split hides the exchange behind the interior kernels but costs three more launches of
        # latency-bound kernels per substep (1 M particles per rank: +25 us on one GPU): whether it pays depends
        # on the wire, which only this run can measure.  Both ways are timed on a few untimed substeps (max over ranks, so
        # every rank takes the same decision) and the faster one is kept.
        dog.phase("overlap tuning", 600)
        job.run(args.warmup)

        def trial(flag, n=12):
            job.set_overlap(flag)
            barrier()
            t0 = time.perf_counter()
            job.run(n)
            job.synchronize()
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt.item()) / n
        t_on, t_off = trial(True), trial(False)
        job.set_overlap(t_on <= t_off)
        overlap_note = {"kept": "on" if t_on <= t_off else "off", "ms_per_step_on": t_on, "ms_per_step_off": t_off}
        print("bench.py[rank %d]: overlap split %s (%.4f ms per substep with, %.4f without)" % (rank, overlap_note["kept"], t_on, t_off),
              file=sys.stderr, flush=True)
    dog.phase("measurement", 900)
    elapsed, ms, dom, prof = measure(args.warmup, args.steps)
    dog.phase("report", 900)
    PHASES = ("sort", "p2g", "exchange", "grid", "g2p")

    def rank_table(ms_here, elapsed_here):
        """every rank's phase table (ms per substep; `exchange` = from the end of the halo pack to the arrival of the peers' sums:
        signal + wire + wait, i.e. where a rank waits for a slower peer), its elapsed time and what its plan moves — gathered over
        the control group, reported as max / min over the ranks and rank by rank"""
        st = (job.state()[0] if hasattr(job, "state") else {})
        row = [ms_here.get(k, 0.0) for k in PHASES] + [1e3 * elapsed_here / args.steps, float(n_local), float(st.get("halo_nodes", 0) * 16),
                                                       float(st.get("halo_boxes", 0)), float(st.get("migrations", 0)), float(st.get("migrated_out", 0)),
                                                       float(st.get("replans", 0))]
        rows = [torch.zeros(len(row), dtype=torch.float64) for _ in range(world)]
        if world > 1:
            dist.all_gather(rows, torch.tensor(row, dtype=torch.float64))
        else:
            rows = [torch.tensor(row, dtype=torch.float64)]
        tab = torch.stack(rows)
        names = list(PHASES) + ["ms_per_step", "particles", "halo_bytes_per_substep", "halo_boxes", "migrations", "migrated_out", "replans"]
        return {"max": {k: float(tab[:, i].max()) for i, k in enumerate(names)}, "min": {k: float(tab[:, i].min()) for i, k in enumerate(names)},
                "per_rank": [{k: float(tab[r, i]) for i, k in enumerate(names)} for r in range(world)]}

    tiled_info = None
    if world > 1 or force_tiled:
        tiled_info = rank_table(ms, elapsed)
        tiled_info["wire"] = native_wire or ("torch" if data_group is not None else "staged")
        if native_wire and hasattr(job, "totals"):
            try:  # live particles / sticky error word of the whole job, reduced inside the library (mpmhip_tiled_totals)
                tiled_info["totals"] = job.totals()
            except Exception as e:
                tiled_info["totals"] = {"error": repr(e)}
        if args.wire == "both" and native_wire in ("rccl", "ipc"):
            # the same job, re-wired: IPC peer writes instead of send / receive groups (collective: every rank comes here).  (Where the
            # first wire already is ipc — the RCCL probe failed, or the test hook that lets the ranks share a GPU — the job is
            # re-planned and re-connected all the same: the second object then measures the re-wired job on the same wire.)
            dog.phase("second wire (ipc)", 600)
            try:
                job.rewire("ipc")
                e2, ms2, _, _ = measure(args.warmup, args.steps)
                other = rank_table(ms2, e2)
                t2 = torch.tensor([e2], dtype=torch.float64)
                if world > 1:
                    dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                other.update(wire="ipc", ms_per_step=1e3 * float(t2.item()) / args.steps, overlap_split=job.overlap)
                tiled_info["other_wire"] = other
            except Exception as e:
                tiled_info["other_wire"] = {"wire": "ipc", "error": repr(e)}
            dog.phase("report", 900)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nt = torch.tensor([n_local], dtype=torch.float64)
        dist.all_reduce(nt, op=dist.ReduceOp.SUM)
        n_total = int(nt.item())
    else:
        n_total = n_local
    copy_gbs = None
    if world == 1 and not force_tiled:  # what a plain streaming copy reaches on THIS box, measured after the timed region
        try:
            copy_gbs = job.sim.copy_bandwidth(1 << 30, 5)
        except Exception as e:
            print("bench.py: copy bandwidth probe failed: %r" % (e,), file=sys.stderr)
    if rank != 0:
        dog.stop()
        if world > 1:
            dist.destroy_process_group()
        return

    tag = args.config + ("_evolved" if args.state == "evolved" else "")
    roof, both_frac, n_per_gpu, nodes = roofline_of(ms, dom, prof, tag)
    roof["measured_copy_GBs"] = copy_gbs  # plain float4 copy kernel on this box (read + written bytes)
    roof["frac_of_measured_copy"] = (roof["achieved"] / copy_gbs) if copy_gbs else None
    value = n_total * args.steps / elapsed
    whole_step_bytes = n_per_gpu * 252.0 + nodes * 16.0 * 5
    state_desc = ("the lattice the reference's benchmark seeds (8 particles per cell, F = I, slots in sorted order): the best "
                  "case; see `evolved` for the same scene after the block has hit the floor")
    if args.state == "evolved":
        state_desc = "%d substeps after the block hit the floor (--state evolved)" % EVOLVE_AFTER_IMPACT
    out = {
        "metric": ("particle-steps/sec (P2G+grid+G2P), 256^3 grid 8M particles; %HBM roofline" if args.config == "c3" and not args.cells
                   else "particle-steps/sec (P2G+grid+G2P), " + cfg["desc"] + "; %HBM roofline"),
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": job.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["desc"], "particles": n_total, "dt": cfg.get("dt", 1e-4), "parallelism": job.parallelism, "wire": wire,
                   "overlap_split": overlap_note,
                   "timed": "full substep: sort+reorder, P2G, grid normalise+boundary, G2P, boundary cleanup",
                   "state": state_desc},
        "roofline": roof,
        "phases_ms_per_step": ms,
        "tiled": tiled_info,
        "p2g_plus_g2p_particle_steps_per_s": n_per_gpu / ((ms["p2g"] + ms["g2p"]) * 1e-3),
        "p2g_plus_g2p_hbm_frac_algorithmic": both_frac,
        "whole_step_hbm_frac_algorithmic": whole_step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
    }
    if world == 1 and not force_tiled:
        # the same K substeps with the cells in creation-id order behind every sort (bitwise reproducible runs): what the mode costs
        try:
            job.sim.set_deterministic(True)
            job.run(args.warmup)
            barrier()
            t0 = time.perf_counter()
            job.run(args.steps)
            job.synchronize()
            d_el = time.perf_counter() - t0
            job.sim.set_deterministic(False)
            out["deterministic"] = {"ms_per_step": 1e3 * d_el / args.steps, "value": n_total * args.steps / d_el,
                                    "extra_ms_per_step": 1e3 * (d_el - elapsed) / args.steps,
                                    "what": "mpmhip_config.deterministic: k_cell_order_blocks behind every sort (in-cell order by creation id; the ids in a 4-byte array the key writers keep)"}
        except Exception as e:
            out["deterministic"] = {"error": repr(e)}
    if world == 1 and not force_tiled and args.state == "lattice" and not args.no_evolved:
        # the same scene after impact, on the same ctx: the state the lattice number flatters
        try:
            after = EVOLVE_AFTER_IMPACT
            job.run(substeps_to_impact(cfg) + after)
            job.synchronize()
            e_el, e_ms, e_dom, e_prof = measure(5, args.steps)
            e_roof, e_both, e_n, e_nodes = roofline_of(e_ms, e_dom, e_prof, args.config + "_evolved")
            out["evolved"] = {
                "substeps_before": job.substeps - args.steps,  # run on this ctx before the timed region of this object
                "substeps_after_impact": after, "particles": e_n, "active_blocks": e_prof["active_blocks"],
                "value": e_n * args.steps / e_el, "ms_per_step": 1e3 * e_el / args.steps, "phases_ms_per_step": e_ms,
                "roofline": e_roof, "p2g_plus_g2p_hbm_frac_algorithmic": e_both,
                "whole_step_hbm_frac_algorithmic": (e_n * 252.0 + e_nodes * 80.0) / (e_el / args.steps) / 1e9 / HBM_PEAK_GBS}
            try:  # how ill-conditioned the state's deformation gradients are (the device's fp32 tolerances hold to cond(F) 1e2)
                out["evolved"]["cond_F"] = job.sim.cond_census()
            except Exception as e:
                out["evolved"]["cond_F"] = {"error": repr(e)}
            if e_n < 0.99 * n_per_gpu:  # particles deleted on the way (left the domain / non-finite): not the workload any more
                out["evolved"]["warning"] = "%d of %d particles were deleted before the timed region" % (n_per_gpu - e_n, n_per_gpu)
        except Exception as e:
            out["evolved"] = {"error": repr(e)}
    if world == 1 and not force_tiled and not args.no_virtual and args.config == "c3" and not args.cells and args.state == "lattice":
        # The only multi-GPU evidence one GPU can give, on the same box and in the same process as the headline: the K bricks of the
        # workload as K ctx on this GPU (the library's own loop, halo exchange and migration over the local wire; the ranks' kernels
        # run one after another on one stream, so the time per rank is what one rank alone on a device computes per substep).
        try:
            job.sim.close()  # (the one-GPU ctx is done with: the ranks get the memory)
            t_block = time.time()
            vout = {"what": "K bricks of the workload as K ctx on THIS GPU (mpmhip_tiled_advance_group, MPMHIP_WIRE_LOCAL): per_rank_ms = "
                            "all ranks' substeps back to back / K = the per-GPU compute of a K-GPU run before the wire; x_over_one_gpu = "
                            "ms_per_step of this line / per_rank_ms = upper bound of the strong-scaling factor", "one_gpu_ms_per_step": out["ms_per_step"],
                    "ranks": {}}
            for K in (8, 2, 4):
                if K != 8 and time.time() - t_block > VIRTUAL_BUDGET_S * (0.5 if K == 2 else 0.75):
                    vout["ranks"][str(K)] = {"skipped": "time budget of the block (%.0f s) nearly spent" % VIRTUAL_BUDGET_S}
                    continue
                v = virtual_run(tm, cfg, args, K=K, overlap=False)
                row = {"per_rank_ms": v["per_rank_ms_serial_no_events"], "x_over_one_gpu": out["ms_per_step"] / v["per_rank_ms_serial_no_events"],
                       "particles_per_rank": v["particles_per_rank"], "halo_bytes_per_rank": v["halo_bytes_per_rank"],
                       "rank0_phases_ms": v["rank0_phases_ms"], "per_rank_compute_ms": v["per_rank_compute_ms_per_phase_events"],
                       "dims": v["dims"], "migrated": v["migrated"]}
                if K == 8 and time.time() - t_block < VIRTUAL_BUDGET_S * 0.3:
                    v2 = virtual_run(tm, cfg, args, K=K, overlap=True)  # (the boundary / interior split a real wire may want: three more launches)
                    row["per_rank_ms_overlap_split"] = v2["per_rank_ms_serial_no_events"]
                vout["ranks"][str(K)] = row
            vout["seconds"] = time.time() - t_block
            out["virtual"] = vout
        except Exception as e:
            out["virtual"] = {"error": repr(e)}
    if world == 1 and not force_tiled and not args.no_c5 and args.config == "c3" and not args.cells and args.state == "lattice":
        # BASELINE configs[4] (512^3 / 64 M mixed particles: the configuration the >= 6x at 8 GPUs is claimed for) on THIS GPU in one ctx,
        # then as 8 virtual ranks (one 8 M cluster each): the same proxy as `virtual`, driver-timed
        try:
            try:
                job.sim.close()
            except Exception:
                pass
            out["c5"] = c5_block(tm, args)
        except Exception as e:
            out["c5"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import refmpm
            out["cpu_baseline"] = cpu_baseline_reference(cfg) if refmpm.available() else cpu_baseline(cfg)
        except Exception as e:  # the baseline is a reported extra: never lose the GPU line because of it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    emit(out)
    dog.stop()
    if world > 1 or force_tiled:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
