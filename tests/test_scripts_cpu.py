"""The `taichi` alias package (compat/taichi): the names the reference's scene scripts import, bound to taichi_mpm_amd — and, where
the reference tree exists (this container), the reference's own Python run UNMODIFIED from /root/reference/scripts against it:
scripts/benchmark/benchmark_3d.py reaches the device boundary without one edited line, and the recorded calls of
tests/golden/script_traces.json (replayed on the GPU by tests/test_gpu_scripts.py) are what the scripts make today."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")
REF_SCRIPTS = "/root/reference/scripts"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_SCRIPTS), reason="the reference tree is not on this box")


@pytest.fixture()
def tc(monkeypatch):
    monkeypatch.syspath_prepend(COMPAT)
    import taichi
    return taichi


def test_alias_offers_the_names_the_reference_scripts_use(tc, tmp_path, monkeypatch):
    """`grep -ho "tc\\.[A-Za-z_.0-9]*" scripts/` of the reference: every name exists; the MPM ones work, scene tooling refuses loudly"""
    from taichi.core import tc_core  # scripts/async/async_mpm.py:1-8: the driver's imports, line by line
    from taichi.dynamics.levelset import LevelSet
    from taichi.gui.image_viewer import show_image  # noqa: F401
    from taichi.misc import util
    from taichi.tools.video import VideoManager
    from taichi.visual.camera import Camera  # noqa: F401
    from taichi.visual.particle_renderer import ParticleRenderer  # noqa: F401
    from taichi.visual.post_process import LDRDisplay  # noqa: F401
    import taichi_mpm_amd as tm
    for name in ("P", "Vector", "Vectori", "function_addresses", "time", "taichi"):
        assert name in util.__all__ and hasattr(util, name)
    v = tc.Vector(0.0, 1, 0)
    assert (v.x, v.y, v.z) == (0.0, 1.0, 0.0) and tuple(v * 2.0) == (0.0, 2.0, 0.0) and tc.Vectori((3, 4, 5)).z == 5
    assert tc.set_gdb_trigger() is None and tc.P(a=1) == {"a": 1}
    f = tc.constant_function13((1, 2, 3))
    assert f(0.5) == (1.0, 2.0, 3.0) and f in tc.function_addresses and tc.constant_function((0.5, 0.35))(0) == (0.5, 0.35)
    assert tc.function13(lambda t: (t, 0, 0))(2.0) == (2.0, 0.0, 0.0)
    ls = LevelSet(tc.Vectori((32, 32, 32)), tc.Vector(0.0))
    ls.add_plane(tc.Vector(0.0, 1, 0), -0.2)
    assert ls.levelset is ls and ls.planes == [(0.0, 1.0, 0.0, -0.2)] and ls.get_delta_x() == 1.0 / 32
    assert isinstance(tc_core.create_simulation3("mpm"), tm.Simulation3D) and type(tc.core.create_simulation3("async_mpm")).__name__ == "AsyncSimulation3D"
    assert type(tc_core.create_simulation2("mpm")).__name__ == "Simulation2D"
    assert isinstance(tc.core.DynamicLevelSet3D().initialize(0.0, 1.0, ls, ls), tm.mpm.DynamicLevelSet)
    assert tuple(tc_core.Vector3f(0.0)) == (0.0, 0.0, 0.0)
    monkeypatch.setenv("TAICHI_MPM_OUTPUT", str(tmp_path))
    d = tc.get_output_path("async_mpm/unit", True)
    assert os.path.isdir(d) and VideoManager(d).get_frame_directory() == os.path.join(d, "frames")
    open(os.path.join(d, "frames", "0001.bgeo"), "w").close()
    tc.clear_directory_with_suffix(os.path.join(d, "frames"), "bgeo")
    assert os.listdir(os.path.join(d, "frames")) == []
    for refuse in (lambda: tc.Texture("sphere", center=(0.5, 0.5, 0.5), radius=0.1), lambda: tc.SegmentMesh()):
        with pytest.raises(tm.MPMError, match="outside this build"):
            refuse()
    monkeypatch.setenv("TAICHI_MPM_NUM_FRAMES", "3")
    m = tc.dynamics.MPM(res=(32, 32, 32), task_id="unit")
    assert isinstance(m, tm.MPM) and m.num_frames == 3 and m.c.frame_directory == os.path.join(str(tmp_path), "mpm", "unit", "frames")


@needs_reference
def test_the_committed_traces_are_what_the_reference_scripts_ask_for_today():
    """tests/golden/make_script_traces.py run again (the reference's benchmark_3d.py under runpy, its AsyncMPM driver class on the
    small scene) gives the committed tests/golden/script_traces.json: the GPU replay cannot go stale"""
    code = ("import json, sys; sys.path.insert(0, %r); import make_script_traces as m; "
            "sys.stdout = sys.stderr; t = m.record_all(); sys.__stdout__.write(json.dumps(t, sort_keys=True))" % os.path.join(ROOT, "tests", "golden"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(os.path.join(ROOT, "tests", "golden", "script_traces.json")) as f:
        want = json.load(f)
    got = json.loads(r.stdout)
    assert got == want
    assert [c[0] for c in want["benchmark_3d"]["calls"]] == ["MPM", "add_particles", "simulate"]
    assert want["benchmark_3d"]["calls"][1][1]["benchmark"] == 8000


@needs_reference
def test_reference_benchmark_script_runs_verbatim_up_to_the_device(tmp_path):
    """`PYTHONPATH=compat python /root/reference/scripts/benchmark/benchmark_3d.py`: zero edited lines.  On a box without a GPU it
    builds the driver, stages the 8 M benchmark particles and stops exactly where the first step creates the device context
    (no CPU fallback); tests/test_gpu_scripts.py runs the same calls through to the end on an MI355X."""
    lib = os.path.join(ROOT, "taichi_mpm_amd", "lib", "libmpmhip.so")
    if not os.path.exists(lib):
        pytest.skip("libmpmhip.so is not built")
    from tests.conftest import _hip_device_count
    if _hip_device_count() > 0:
        pytest.skip("a GPU is present: the script would run through (covered by the gpu tests)")
    env = dict(os.environ, PYTHONPATH=COMPAT, TAICHI_MPM_NUM_FRAMES="2")
    r = subprocess.run([sys.executable, os.path.join(REF_SCRIPTS, "benchmark", "benchmark_3d.py")], capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode != 0
    tail = r.stderr.strip().splitlines()[-1]
    assert "MPMError" in tail and "no HIP device available" in tail, r.stderr[-2000:]
    assert "in simulate" in r.stderr and "_ensure_ctx" in r.stderr  # it got as far as the first step of the frame loop
