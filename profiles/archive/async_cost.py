"""cost of the device-resident asynchronous stepper next to the synchronous one: the two-stiffness scene of
tests/test_gpu_async.py at 128^3 (soft Hencky-elastic slab next to stiff sand, ~0.5 M particles): wall time per simulated
millisecond, particle updates, advances.   python profiles/async_cost.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import taichi_mpm_amd as tm
from tests.common import lattice_cube, make_state

res = 128; dx = 1.0 / res
xa = lattice_cube(res, 30, 70, dx, jitter=0.2, seed=11); xa = xa[xa[:, 0] < 50 * dx]
xb = lattice_cube(res, 30, 70, dx, jitter=0.2, seed=12); xb = xb[xb[:, 0] >= 50 * dx]
sa = make_state(xa, "elastic", dx, perturb_F=0.01, vel_scale=0.5, seed=13)
sb = make_state(xb, "sand", dx, perturb_F=0.01, vel_scale=0.5, seed=14)
out = {"particles": len(xa) + len(xb)}
T = 4e-3

def scene(kind, **cfg):
    sim = tm.create_simulation3(kind).initialize(dict(res=(res,) * 3, delta_x=dx, **cfg))
    sim.set_levelset(tm.mpm.LevelSet(friction=0.4).add_plane((0, 1, 0), d=-0.2))
    for s, mat in ((sa, "elastic"), (sb, "sand")):
        sim.add_particles(dict(type=mat, positions=s.x, velocities=s.v, F=s.F, B=s.B, aux=s.aux, params=s.gparams[0]))
    return sim

a = scene("async_mpm", unit_delta_t=2e-6, max_units=256)
import ctypes as C
a.step(4e-4)  # warm-up (allocations, first limits)
pr0 = (C.c_double * 6)(); a._L.mpmhip_async_profile(a._ctx, int(os.environ.get("ASYNC_PROFILE_SYNC", "0")), pr0)
st0 = a._state(); t0 = time.perf_counter()
a.step(T); a.synchronize() if hasattr(a, "synchronize") else None
el = time.perf_counter() - t0; st1 = a._state()
tab = a.block_table()
pr1 = (C.c_double * 6)(); a._L.mpmhip_async_profile(a._ctx, 0, pr1)
out["async_host_ms"] = dict(zip(("update_dt_limits", "neighbour_lists", "advance", "substep", "compaction", "advances"), [pr1[k] - pr0[k] for k in range(6)]))
out["async"] = {"wall_ms_per_simulated_ms": el / T, "particle_updates": st1[1] - st0[1], "limits_in_use": sorted(set(int(v) for v in tab["continuous"][tab["count"] > 0])),
                "store_containers": st1[5], "live_containers": st1[4], "compactions": st1[6], "host_particle_bytes_during_step": 0}
a.close()
# the synchronous stepper has to take the stiffest block's step everywhere
lim = min(out["async"]["limits_in_use"])
dt = 2e-6 * lim
s = scene("mpm", base_delta_t=dt, keep_apic_b=True)
s.run_substeps(20); s.synchronize()
n = int(round(T / dt)); t0 = time.perf_counter()
s.run_substeps(n); s.synchronize()
el = time.perf_counter() - t0
out["sync"] = {"dt": dt, "substeps": n, "wall_ms_per_simulated_ms": el / T, "particle_updates": n * out["particles"]}
s.close()
print(json.dumps(out))
