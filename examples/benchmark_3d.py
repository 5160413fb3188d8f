"""The reference's scripts/benchmark/benchmark_3d.py with one changed import: `taichi_mpm_amd` instead of `taichi`.

Same scene (125^3 grid, built-in benchmark lattice of 100^3 cells x 8 linear-elastic particles, no gravity, no boundary
cleaning, dt = frame_dt = 1e-2), same driver calls (`MPM(...)`, `add_particles(benchmark=8000, ...)`,
`simulate(print_profile_info=True)`); the substep runs in the HIP kernels of libmpmhip.  Needs an MI355X.

    python examples/benchmark_3d.py [frames]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taichi_mpm_amd as tc_amd  # noqa: E402

r = 125

if __name__ == '__main__':
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    res = (r, r, r)
    mpm = tc_amd.MPM(
        res=res,
        base_delta_t=1e-2,
        gravity=0,
        clean_boundary=False,
        frame_dt=1e-2,
        num_frames=frames,
        # accepted and ignored (CPU-path switches of the reference): benchmark_resample, num_threads, optimized
        benchmark_resample=False,
        num_threads=1,
        optimized=True)

    mpm.add_particles(
        benchmark=8000,
        type='linear',
        initial_velocity=(0, 0, 0),
        E=1e2)

    mpm.simulate(print_profile_info=True)
    print("particles:", mpm.c.get_num_particles(), " simulated time:", mpm.get_current_time(),
          " wall time in step(): %.3f s" % mpm.simulation_total_time)
