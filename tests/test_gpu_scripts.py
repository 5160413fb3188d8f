"""The reference's scene-script calls, replayed on the device.  tests/golden/script_traces.json holds the calls the reference's own
Python makes when it runs UNMODIFIED from /root/reference/scripts against compat/taichi (recorded by tests/golden/make_script_traces.py;
tests/test_scripts_cpu.py re-records them where the reference exists) — the scripts cannot travel to the GPU box, their argument values
can.  Here the same calls go through the alias package into libmpmhip and the result is compared with the reference's solver
(oracle/_ref/libmpm_ref.so) fed the same generator."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def traces():
    with open(os.path.join(ROOT, "tests", "golden", "script_traces.json")) as f:
        return json.load(f)


@pytest.fixture()
def tc(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "compat"))
    import taichi
    return taichi


def _ref():
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so did not travel to this box")
    return ref


def _tuples(kw):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()}


def test_benchmark_3d_calls_run_through_the_alias_and_match_the_reference(tc, traces, monkeypatch, tmp_path, capsys):
    """scripts/benchmark/benchmark_3d.py:9-27, statement by statement: tc.dynamics.MPM(**kwargs), add_particles(benchmark=8000,
    type='linear', ...), simulate(clear_output_directory=True, print_profile_info=True) — 8 M particles on a 125^3 grid; frames
    bounded from outside by TAICHI_MPM_NUM_FRAMES, as when the file itself is run.  Against MPM<3> of the reference with the same
    config and its own benchmark generator (src/mpm.cpp:149-186), stepped frame by frame (MPM::step, :428-439)."""
    ref = _ref()
    frames = int(traces["frames"])
    monkeypatch.setenv("TAICHI_MPM_NUM_FRAMES", str(frames))
    monkeypatch.setenv("TAICHI_MPM_OUTPUT", str(tmp_path))
    calls = traces["benchmark_3d"]["calls"]
    assert calls[0][0] == "MPM"
    ctor = _tuples(calls[0][1])
    mpm = tc.dynamics.MPM(**ctor)
    for name, kw in calls[1:]:
        getattr(mpm, name)(**_tuples(kw))
    assert mpm.c.frame == frames
    got = mpm.c.get_particles()
    assert len(got["x"]) == 8000000
    out = capsys.readouterr().out
    assert out.count('"phases"') == frames  # print_profile_info=True: one phase table per frame
    files = sorted(os.listdir(mpm.c.frame_directory))
    assert len(files) == frames and all(f.endswith(".bgeo") for f in files)  # one frame file per frame, as the reference's visualize()
    # the reference, same config keys (the CPU-only ones included), same generator, same frame loop
    ref.set_threads(min(16, os.cpu_count() or 1))
    add = _tuples(calls[1][1])
    r = ref.Sim(ctor["res"], 1.0 / ctor["res"][0], ctor["base_delta_t"], gravity=(ctor["gravity"],) * 3,
                clean_boundary=ctor["clean_boundary"], optimized=ctor["optimized"], num_threads=ctor["num_threads"])
    r.add_benchmark(add["type"], add["benchmark"], E=add["E"], initial_velocity=add["initial_velocity"])
    for _ in range(frames):
        r.step(ctor["frame_dt"])
    assert abs(r.time() - mpm.get_current_time()) < 1e-7 and mpm.get_current_time() > 0  # the same number of substeps
    want = r.download()
    r.close()
    assert np.array_equal(got["id"], want["id"])
    assert np.abs(got["x"] - want["x"]).max() <= 2e-7  # (the lattice itself: cell centre +- dx / 4 rounded once here, term by term there)
    assert np.abs(got["v"] - want["v"]).max() <= 1e-6 and np.abs(got["F"] - want["F"]).max() <= 1e-6
    mpm.c.close()


def test_reference_async_driver_calls_match_the_reference_async_stepper(traces, tmp_path):
    """the calls scripts/async/async_mpm.py's AsyncMPM class makes on tc_core.create_simulation3('async_mpm') — initialize(P(**kwargs))
    with its injected keys, a DynamicLevelSet of two equal key frames before every step (:119-127), step(frame_dt), visualize(), the
    frame counter — replayed on the library's async stepper; against AsyncMPM<3> of the reference on the same two-stiffness scene"""
    ref = _ref()
    import taichi_mpm_amd as tm
    from taichi_mpm_amd.mpm import lattice_cube
    from tests.common import rel_l2
    T = traces["async_driver"]

    def decode(v):
        if isinstance(v, dict) and "__levelset__" in v:
            ls = tm.mpm.LevelSet(friction=v["friction"])
            for t_, io, p in v["__levelset__"]:
                ls._add(t_, io, p)
            return ls
        if isinstance(v, dict) and "__dynamic_levelset__" in v:
            t0, t1, a, b = v["__dynamic_levelset__"]
            return tm.mpm.DynamicLevelSet().initialize(t0, t1, decode(a), decode(b))
        if isinstance(v, dict):
            return {k: (str(tmp_path) if x == "<frame_directory>" else decode(x)) for k, x in v.items()}
        if isinstance(v, list):
            return tuple(decode(x) for x in v)
        return v
    sim = tm.create_simulation3("async_mpm")
    steps, frames_written = [], []
    for name, args in T["calls"]:
        a = [decode(x) for x in args]
        if name == "frame=":
            sim.frame = a[0]
        elif name == "visualize":
            frames_written.append(sim.visualize())
        else:
            getattr(sim, name)(*a)
            if name == "step":
                steps.append(a[0])
    assert len(steps) == 2 and len(frames_written) == 2 and all(os.path.getsize(f) > 0 for f in frames_written)
    S = T["scene"]
    res = S["ctor"]["res"][0]
    dx = 1.0 / res
    ref.set_threads(1)
    r = ref.AsyncSim(res, dx, shapes=[(0, 0) + tuple(S["plane"][0]) + (S["plane"][1],)], friction=S["friction"],
                     unit_delta_t=S["ctor"]["unit_delta_t"], max_units=S["ctor"]["max_units"], cfl_dt_mul=S["ctor"]["cfl_dt_mul"])
    vol = dx ** 3 / 8
    for g in S["groups"]:
        kw = {k: v for k, v in g.items() if k not in ("type", "cube")}
        r.add_particles(g["type"], 400.0 * vol, vol, lattice_cube(g["cube"][0], g["cube"][1], dx), **kw)
    for dt in steps:
        r.step(dt)
    assert sim.current_t_int == r.time_int() and sim.update_counter == r.update_counter()
    a, b = sim.get_pool_particles(), r.download()
    assert np.array_equal(a["id"], b["id"]) and len(a["id"]) == 2 * 6 ** 3 * 8
    assert len(np.unique(b["limits"][:, 0])) >= 2, "the scene must step with at least two block step sizes"
    assert np.array_equal(a["continuous"], b["limits"][:, 0]) and np.array_equal(a["particle_t"], b["limits"][:, 3])
    assert np.abs(a["x"] - b["x"]).max() <= 1e-6
    assert rel_l2(a["v"], b["v"]) <= 5e-4 and rel_l2(a["F"], b["F"]) <= 1e-4
    sim.close(); r.close()
