#!/usr/bin/env python
"""Runs the reference's own PYTHON, unmodified and from where it lies (/root/reference/scripts), against the `taichi` alias
package of this repository (compat/taichi) and records what it asks of the MPM path -> tests/golden/script_traces.json.

The GPU box has no /root/reference, so the scripts themselves cannot run there; what travels is DATA: the argument values of the
calls the scripts made.  tests/test_gpu_scripts.py replays them against libmpmhip and compares with the reference's solver;
tests/test_scripts_cpu.py (here, where the reference exists) re-records them and checks that the committed file is current.

  benchmark_3d   scripts/benchmark/benchmark_3d.py run with runpy as __main__, `tc.dynamics.MPM` replaced by a recorder:
                 constructor kwargs + every driver call, in order (driver level: exactly the script's statements)
  async_driver   scripts/async/async_mpm.py: its `AsyncMPM` driver CLASS (the reference's async scenes need meshes / textures, so
                 the scene is the small one below) recorded one level down, at the object `tc_core.create_simulation3('async_mpm')`
                 returns: every call the reference's driver makes on it (initialize(P(**kwargs)), set_levelset(DynamicLevelSet),
                 step, visualize, ...)

    python tests/golden/make_script_traces.py            # rewrites tests/golden/script_traces.json
"""
import json
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SCRIPTS = "/root/reference/scripts"
OUT = os.path.join(HERE, "script_traces.json")
FRAMES = 3  # TAICHI_MPM_NUM_FRAMES of the recording (the scripts' own default is 1000)

ASYNC_SCENE = dict(  # the small two-stiffness scene the async driver class is recorded on (cells of a 32^3 grid)
    ctor=dict(res=(32, 32, 32), unit_delta_t=2e-6, max_units=1024, cfl_dt_mul=1.0, frame_dt=2.5e-3, num_frames=2, task_id="trace"),
    plane=((0.0, 1.0, 0.0), -0.2), friction=0.4,
    groups=[dict(type="elastic", cube=(9, 15), E=5e3), dict(type="sand", cube=(15, 21))])


def encode(v):
    """argument values -> JSON (tuples become lists; level sets become their shapes)"""
    from taichi_mpm_amd.mpm import DynamicLevelSet, LevelSet
    if isinstance(v, LevelSet):
        return {"__levelset__": [list(s[:2]) + [list(s[2])] for s in v.shapes], "friction": v.friction}
    if isinstance(v, DynamicLevelSet):
        return {"__dynamic_levelset__": [v.t0, v.t1, encode(v.levelset0), encode(v.levelset1)]}
    if isinstance(v, dict):
        return {str(k): encode(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [encode(x) for x in v]
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    raise TypeError("cannot record %r" % (v,))


def record_benchmark_3d():
    import taichi as tc
    calls = []

    class Recorder:
        def __init__(self, **kwargs):
            calls.append(["MPM", encode(kwargs)])

        def __getattr__(self, name):
            def call(*args, **kwargs):
                assert not args, "the script passes keywords only"
                calls.append([name, encode(kwargs)])
                return ""
            return call
    real = tc.dynamics.MPM
    tc.dynamics.MPM = Recorder
    try:
        runpy.run_path(os.path.join(REF_SCRIPTS, "benchmark", "benchmark_3d.py"), run_name="__main__")
    finally:
        tc.dynamics.MPM = real
    return {"script": "scripts/benchmark/benchmark_3d.py", "level": "driver (tc.dynamics.MPM)", "calls": calls}


def record_async_driver(tmp):
    import taichi as tc
    from taichi.core import tc_core
    calls = []

    class Vis:
        x, y = 32, 32

    class RecordingSim:
        """stands where tc_core.create_simulation3('async_mpm') stands; keeps the clock the driver reads"""
        def __init__(self):
            self.t = 0.0

        def get_current_time(self):
            return self.t

        def get_vis_resolution(self):
            return Vis()

        def step(self, dt):
            calls.append(["step", [encode(dt)]])
            self.t += dt

        def __setattr__(self, k, v):
            if k == "frame":
                calls.append(["frame=", [encode(v)]])
            object.__setattr__(self, k, v)

        def __getattr__(self, name):
            def call(*args):
                a = encode(list(args))
                if name == "initialize":  # the driver injects its frame directory (scripts/async/async_mpm.py:49): a path of this run
                    a[0]["frame_directory"] = "<frame_directory>"
                calls.append([name, a])
                return ""
            return call
    made = []

    def create3(name):
        made.append(name)
        return RecordingSim()
    real3 = tc_core.create_simulation3, tc.core.create_simulation3
    tc_core.create_simulation3 = staticmethod(create3) if isinstance(tc_core, type) else create3
    tc.core.create_simulation3 = create3
    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["trace.py"]
    sys.path.insert(0, os.path.join(REF_SCRIPTS, "async"))
    os.environ["TAICHI_MPM_OUTPUT"] = tmp
    try:
        import async_mpm  # the reference's file, from the reference tree
        S = ASYNC_SCENE
        mpm = async_mpm.AsyncMPM(**S["ctor"])
        levelset = mpm.create_levelset()
        levelset.add_plane(tc.Vector(*S["plane"][0]), S["plane"][1])
        levelset.set_friction(S["friction"])
        mpm.set_levelset(levelset, False)
        for g in S["groups"]:
            mpm.add_particles(**g)
        mpm.simulate(clear_output_directory=True, print_profile_info=True)
    finally:
        tc_core.create_simulation3, tc.core.create_simulation3 = real3
        sys.argv = argv
        sys.path.pop(0)
        os.chdir(cwd)
    assert made == ["async_mpm"]
    return {"script": "scripts/async/async_mpm.py (class AsyncMPM)", "level": "simulation object (tc_core.create_simulation3('async_mpm'))",
            "scene": encode(ASYNC_SCENE), "calls": calls}


def record_all():
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    os.environ["TAICHI_MPM_NUM_FRAMES"] = str(FRAMES)
    with tempfile.TemporaryDirectory() as tmp:
        return {"frames": FRAMES, "benchmark_3d": record_benchmark_3d(), "async_driver": record_async_driver(tmp)}


if __name__ == "__main__":
    traces = record_all()
    with open(OUT, "w") as f:
        json.dump(traces, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, {k: len(v["calls"]) for k, v in traces.items() if isinstance(v, dict)})
