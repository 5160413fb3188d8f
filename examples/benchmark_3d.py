"""The reference's scripts/benchmark/benchmark_3d.py on libmpmhip — the reference's own file, not a copy of it.

    python examples/benchmark_3d.py [frames] [/path/to/taichi_mpm/scripts/benchmark/benchmark_3d.py]

puts compat/ (the `taichi` alias package) on the import path and runs the reference's script with runpy, as
`PYTHONPATH=compat python <reference>/scripts/benchmark/benchmark_3d.py` does: 125^3 grid, the built-in benchmark lattice of
100^3 cells x 8 linear-elastic particles, no gravity, dt = frame_dt = 1e-2.  Where the reference tree is absent (the GPU boxes of
this project) the calls the script makes are replayed from tests/golden/script_traces.json, recorded from that very file by
tests/golden/make_script_traces.py.  Needs an MI355X.
"""
import json
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

if __name__ == "__main__":
    os.environ.setdefault("TAICHI_MPM_NUM_FRAMES", sys.argv[1] if len(sys.argv) > 1 else "10")
    script = sys.argv[2] if len(sys.argv) > 2 else "/root/reference/scripts/benchmark/benchmark_3d.py"
    if os.path.exists(script):
        runpy.run_path(script, run_name="__main__")
    else:
        import taichi as tc
        with open(os.path.join(ROOT, "tests", "golden", "script_traces.json")) as f:
            calls = json.load(f)["benchmark_3d"]["calls"]
        mpm = tc.dynamics.MPM(**{k: tuple(v) if isinstance(v, list) else v for k, v in calls[0][1].items()})
        for name, kw in calls[1:]:
            getattr(mpm, name)(**{k: tuple(v) if isinstance(v, list) else v for k, v in kw.items()})
        print("particles:", mpm.c.get_num_particles(), " simulated time:", mpm.get_current_time(),
              " wall time in step(): %.3f s" % mpm.simulation_total_time)
