"""Loader / ctypes prototypes for libmpmhip.so (the C ABI declared in include/mpmhip.h)."""
import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_HERE, "csrc", "mpmhip.hip")
_DEPS = [_SRC, os.path.join(_ROOT, "include", "mpmhip.h")] + sorted(
    os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith(".h"))
_LIBDIR = os.path.join(_HERE, "lib")
# MPMHIP_LIB_VARIANT=<name> loads lib/libmpmhip_<name>.so: an A/B build of the same sources with extra -D flags
# (build_variant below; tuning experiments only — the default library is the product)
_VARIANT = os.environ.get("MPMHIP_LIB_VARIANT", "")
_LIB = os.path.join(_LIBDIR, "libmpmhip%s.so" % (("_" + _VARIANT) if _VARIANT else ""))

NPARAM = 16

# (-munsafe-fp-atomics: hardware float atomics for the 2D solvers' grid scatter (k_mpm88.h, k_mpm2d.h) and for the impulses a
# particle hands to a rigid body (k_rigid.h); the 3D transfers themselves use no float atomics)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-value"]


class Config(C.Structure):
    """mirror of mpmhip_config"""
    _fields_ = [("res", C.c_int32 * 3), ("dx", C.c_float), ("dt", C.c_float), ("gravity", C.c_float * 3),
                ("particle_gravity", C.c_int32), ("apic_damping", C.c_float), ("rpic_damping", C.c_float),
                ("clean_boundary", C.c_int32), ("n_planes", C.c_int32), ("planes", (C.c_float * 4) * 8),
                ("friction", C.c_float), ("max_particles", C.c_int64), ("max_blocks", C.c_int64),
                ("device", C.c_int32), ("reorder_interval", C.c_int32), ("particle_collision", C.c_int32),
                ("discard_apic_b", C.c_int32), ("generic_path", C.c_int32), ("deterministic", C.c_int32), ("reserved", C.c_int32 * 2)]


class Shape(C.Structure):
    """mirror of mpmhip_shape"""
    _fields_ = [("type", C.c_int32), ("inside_out", C.c_int32), ("p", C.c_float * 6)]


MAX_SHAPES = 16


class AsyncConfig(C.Structure):
    """mirror of mpmhip_async_config"""
    _fields_ = [("unit_delta_t", C.c_float), ("max_units", C.c_int64), ("cfl_dt_mul", C.c_float), ("strength_dt_mul", C.c_float),
                ("left_boundary", C.c_int32)]


SCRIPT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_float, C.POINTER(C.c_float))  # mpmhip_script_fn
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)  # mpmhip_exchange_fn


class RigidConfig(C.Structure):
    """mirror of mpmhip_rigid_config"""
    _fields_ = [("codimensional", C.c_int32), ("recenter", C.c_int32), ("reverse_vertices", C.c_int32), ("reserved0", C.c_int32),
                ("density", C.c_float), ("friction", C.c_float * 2), ("restitution", C.c_float), ("scale", C.c_float * 3),
                ("initial_position", C.c_float * 3), ("initial_rotation", C.c_float * 3), ("initial_velocity", C.c_float * 3),
                ("initial_angular_velocity", C.c_float * 3), ("rotation_axis", C.c_float * 3),
                ("linear_damping", C.c_float), ("angular_damping", C.c_float),
                ("scripted_position", SCRIPT_FN), ("position_user", C.c_void_p),
                ("scripted_rotation", SCRIPT_FN), ("rotation_user", C.c_void_p)]


class JointConfig(C.Structure):
    """mirror of mpmhip_joint_config"""
    _fields_ = [("type", C.c_int32), ("obj0", C.c_int32), ("obj1", C.c_int32), ("has_offset1", C.c_int32),
                ("has_target_distance", C.c_int32), ("offset0", C.c_float * 3), ("offset1", C.c_float * 3),
                ("target_distance", C.c_float), ("penalty", C.c_float), ("axis", C.c_float * 3), ("axis_length", C.c_float),
                ("power", C.c_float), ("angular_velocity", C.c_float)]


class RigidConfig2D(C.Structure):
    """mirror of mpmhip2d_rigid_config"""
    _fields_ = [("codimensional", C.c_int32), ("recenter", C.c_int32), ("reverse_vertices", C.c_int32), ("reserved0", C.c_int32),
                ("density", C.c_float), ("friction", C.c_float * 2), ("restitution", C.c_float), ("scale", C.c_float * 2),
                ("initial_position", C.c_float * 2), ("initial_rotation", C.c_float), ("initial_velocity", C.c_float * 2),
                ("initial_angular_velocity", C.c_float), ("linear_damping", C.c_float), ("angular_damping", C.c_float),
                ("scripted_position", SCRIPT_FN), ("position_user", C.c_void_p),
                ("scripted_rotation", SCRIPT_FN), ("rotation_user", C.c_void_p)]


class Snap2DHeader(C.Structure):
    """mirror of the header of a 2D snapshot blob (csrc/frame2d_api.h: Snap2D) — only its size and the first fields are used"""
    _fields_ = [("magic", C.c_char * 8), ("abi", C.c_uint32), ("n_groups", C.c_uint32), ("res", C.c_int32 * 2), ("next_pid", C.c_int32),
                ("n_bodies", C.c_int32), ("n_joints", C.c_int32), ("has_async", C.c_int32), ("n", C.c_int64), ("dx", C.c_float),
                ("base_dt", C.c_float), ("t", C.c_float), ("request_t", C.c_float), ("n_dead", C.c_uint32), ("n_ranked", C.c_uint32),
                ("nb", C.c_int32 * 2), ("unit_delta_t", C.c_float), ("a_request_t", C.c_float), ("a_current_t", C.c_float), ("pad", C.c_float),
                ("nblk", C.c_int64), ("containers", C.c_int64), ("current_t_int", C.c_int64), ("min_delta_t_int", C.c_int64),
                ("max_delta_t_int", C.c_int64), ("update_counter", C.c_int64), ("step_counter", C.c_int64)]


class Config2D(C.Structure):
    """mirror of mpmhip2d_config"""
    _fields_ = [("res", C.c_int32 * 2), ("dx", C.c_float), ("dt", C.c_float), ("gravity", C.c_float * 2),
                ("particle_gravity", C.c_int32), ("apic_damping", C.c_float), ("rpic_damping", C.c_float),
                ("clean_boundary", C.c_int32), ("particle_collision", C.c_int32), ("max_particles", C.c_int64),
                ("device", C.c_int32), ("reserved", C.c_int32 * 3)]


class HaloBox(C.Structure):
    """mirror of mpmhip_halo_box"""
    _fields_ = [("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3), ("peer", C.c_int32), ("reserved", C.c_int32),
                ("send", C.c_void_p), ("recv", C.c_void_p)]


class TiledConfig(C.Structure):
    """mirror of mpmhip_tiled_config"""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("dims", C.c_int32 * 3), ("margin", C.c_int32),
                ("clip_lo", C.c_int32 * 3), ("clip_hi", C.c_int32 * 3), ("migrate_interval", C.c_int32),
                ("migrate_cap", C.c_int32), ("wire", C.c_int32), ("overlap", C.c_int32), ("inbox_records", C.c_int64)]


MAX_PARTS = 16
MAX_HALO_BOXES = 64
MIGRATE_FLOATS = 44
WIRE_RCCL, WIRE_IPC, WIRE_LOCAL, WIRE_LOCAL_RCCL = 1, 2, 3, 4
COMM_ID_BYTES, IPC_HANDLE_BYTES = 128, 64


def lib_path():
    return _LIB


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def build(force=False, verbose=False):
    """Compile csrc/mpmhip.hip for gfx950 into lib/libmpmhip.so (in-tree, so it ships with the snapshot)."""
    os.makedirs(_LIBDIR, exist_ok=True)
    if not force and os.path.exists(_LIB):
        if all(os.path.getmtime(_LIB) >= os.path.getmtime(d) for d in _DEPS):
            return _LIB
    cmd = [_hipcc()] + HIPCC_FLAGS + [_SRC, "-o", _LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _LIB


def build_variant(name, extra_flags):
    """lib/libmpmhip_<name>.so from the same sources with extra compiler flags (A/B timing on one box)"""
    os.makedirs(_LIBDIR, exist_ok=True)
    out = os.path.join(_LIBDIR, "libmpmhip_%s.so" % name)
    subprocess.check_call([_hipcc()] + HIPCC_FLAGS + list(extra_flags) + [_SRC, "-o", out])
    return out


_lib = None

_SYMBOLS = ["mpmhip_abi_version", "mpmhip_set_profile_sampling", "mpmhip_create", "mpmhip_destroy", "mpmhip_last_error", "mpmhip_set_stream", "mpmhip_set_deterministic", "mpmhip_set_levelset", "mpmhip_set_rigid_levelset_collision", "mpmhip_set_dirichlet", "mpmhip2d_set_dirichlet", "mpmhip_set_levelset_shapes", "mpmhip_set_levelset_keyframes",
            "mpmhip_add_group", "mpmhip_add_particles", "mpmhip_num_particles", "mpmhip_download",
            "mpmhip_upload", "mpmhip_substep", "mpmhip_run_substeps", "mpmhip_step", "mpmhip_current_time",
            "mpmhip_synchronize", "mpmhip_sort", "mpmhip_p2g", "mpmhip_grid_update", "mpmhip_g2p",
            "mpmhip_download_grid", "mpmhip_upload_grid", "mpmhip_calculate_energy", "mpmhip_snapshot_size", "mpmhip_snapshot_save", "mpmhip_snapshot_load", "mpmhip_delete_particles_inside_level_set", "mpmhip_bgeo_size", "mpmhip_bgeo_encode", "mpmhip_write_bgeo", "mpmhip_set_profiling", "mpmhip_profile",
            "mpmhip_profile_reset", "mpmhip_set_partition", "mpmhip_set_halo", "mpmhip_halo_pack",
            "mpmhip_substep_begin", "mpmhip_substep_end", "mpmhip_substep_interior", "mpmhip_set_overlap", "mpmhip_tiled_run", "mpmhip_leaver_counts", "mpmhip_migration_scan", "mpmhip_export_leavers",
            "mpmhip_comm_unique_id", "mpmhip_comm_init", "mpmhip_comm_destroy", "mpmhip_comm_selftest", "mpmhip_tiled_setup", "mpmhip_tiled_ipc_handle",
            "mpmhip_tiled_ipc_connect", "mpmhip_tiled_connect_local", "mpmhip_tiled_advance", "mpmhip_tiled_advance_group", "mpmhip_tiled_state", "mpmhip_tiled_plan",
            "mpmhip_tiled_reduce", "mpmhip_tiled_reduce_group", "mpmhip_calculate_energy_group", "mpmhip_tiled_totals", "mpmhip_tiled_totals_group",
            "mpmhip_import_particles", "mpmhip_active_bounds", "mpmhip_num_slots", "mpmhip_request_compaction", "mpmhip_reserve", "mpmhip_capacity", "mpmhip_mpm88_create", "mpmhip_mpm88_destroy", "mpmhip_mpm88_last_error", "mpmhip_mpm88_add",
            "mpmhip_mpm88_num_particles", "mpmhip_mpm88_advance", "mpmhip_mpm88_download", "mpmhip_mpm88_download_grid",
            "mpmhip_async_enable", "mpmhip_async_begin", "mpmhip_async_pool_particles", "mpmhip_async_step", "mpmhip_async_load_pools", "mpmhip_async_state", "mpmhip_async_current_time", "mpmhip_async_block_times", "mpmhip_async_download_pools", "mpmhip_async_profile", "mpmhip_async_snapshot_size", "mpmhip_async_snapshot_save", "mpmhip_async_snapshot_load", "mpmhip_host_particle_bytes", "mpmhip_async_update_dt_limits", "mpmhip_async_blocks", "mpmhip_async_set_time_int", "mpmhip_async_table", "mpmhip_clear_particles", "mpmhip_set_dt", "mpmhip_set_time", "mpmhip_get_clock", "mpmhip_set_clock", "mpmhip_debug_allowed_dt",
            "mpmhip2d_create", "mpmhip2d_destroy", "mpmhip2d_last_error", "mpmhip2d_set_levelset", "mpmhip2d_add_group", "mpmhip2d_add_particles",
            "mpmhip2d_substep", "mpmhip2d_step", "mpmhip2d_current_time", "mpmhip2d_num_particles", "mpmhip2d_download", "mpmhip2d_download_grid",
            "mpmhip2d_async_begin", "mpmhip2d_async_pool_particles", "mpmhip2d_async_step", "mpmhip2d_async_load_pools", "mpmhip2d_async_view_blocks",
            "mpmhip2d_async_state", "mpmhip2d_async_current_time", "mpmhip2d_async_table", "mpmhip2d_bgeo_size", "mpmhip2d_bgeo_encode", "mpmhip2d_write_bgeo", "mpmhip2d_snapshot_size", "mpmhip2d_snapshot_save", "mpmhip2d_snapshot_load",
            "mpmhip2d_set_rigid_coupling", "mpmhip2d_set_rigid_levelset_collision", "mpmhip2d_add_articulation", "mpmhip2d_set_articulation_iterations", "mpmhip2d_add_rigid_body", "mpmhip2d_rigid_get_state", "mpmhip2d_rigid_get_samples", "mpmhip2d_cdf_phase",
            "mpmhip2d_download_cdf", "mpmhip2d_download_colours",
            "mpmhip_set_rigid_coupling", "mpmhip_add_rigid_body", "mpmhip_num_rigid_bodies", "mpmhip_rigid_get_state", "mpmhip_rigid_set_velocity",
            "mpmhip_rigid_get_samples", "mpmhip_rigid_get_mesh", "mpmhip2d_rigid_get_mesh", "mpmhip_rasterize_rigid_boundary", "mpmhip_gather_cdf", "mpmhip_advect_rigid_bodies", "mpmhip_download_cdf",
            "mpmhip_add_articulation", "mpmhip_num_articulations", "mpmhip_set_articulation_iterations", "mpmhip_articulate",
            "mpmhip_download_boundary",
            "mpmhip_debug_copy_bandwidth", "mpmhip_debug_g2p_is_packed", "mpmhip_debug_scan_grid", "mpmhip_debug_cond_census", "mpmhip_debug_gather_bandwidth", "mpmhip_debug_svd3", "mpmhip_debug_force", "mpmhip_debug_plasticity"]


def exported_symbols():
    return list(_SYMBOLS)


def load():
    """dlopen libmpmhip.so and attach prototypes.  Raises if the library has not been built: the product
    path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(taichi_mpm_amd has no CPU fallback)" % _LIB)
    # torch bundles its own HIP runtime: if it is going to be used in this process (tiled runs, bench.py) it must
    # be loaded BEFORE libmpmhip resolves libamdhip64, or the process ends up with two runtimes and torch reports
    # "No HIP GPUs are available".  torch stays plumbing: nothing below calls into it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB)
    if _VARIANT:  # an A/B build may predate the newest entry points: give those inert stand-ins
        class _Tolerant:
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    return C.CFUNCTYPE(C.c_int)(lambda *a: -38)
        L = _Tolerant(L)
    P = C.POINTER
    vp, fp = C.c_void_p, P(C.c_float)
    L.mpmhip_abi_version.restype = C.c_uint32
    L.mpmhip_create.argtypes = [P(Config), P(vp)]
    L.mpmhip_destroy.argtypes = [vp]
    L.mpmhip_destroy.restype = None
    L.mpmhip_last_error.argtypes = [vp]
    L.mpmhip_last_error.restype = C.c_char_p
    L.mpmhip_set_stream.argtypes = [vp, vp]
    L.mpmhip_set_levelset.argtypes = [vp, C.c_int32, fp, C.c_float]
    L.mpmhip_set_levelset_shapes.argtypes = [vp, C.c_int32, P(Shape), C.c_float]
    L.mpmhip_set_dirichlet.argtypes = [vp, C.c_int32]
    L.mpmhip_set_deterministic.argtypes = [vp, C.c_int32]
    L.mpmhip_set_rigid_levelset_collision.argtypes = [vp, C.c_int32]
    L.mpmhip2d_set_dirichlet.argtypes = [vp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float]
    L.mpmhip_set_levelset_keyframes.argtypes = [vp, C.c_float, C.c_float, C.c_int32, P(Shape), C.c_int32, P(Shape), C.c_float]
    L.mpmhip_add_group.argtypes = [vp, C.c_int32, fp]
    L.mpmhip_add_particles.argtypes = [vp, C.c_int32, C.c_int64, fp, fp, fp, fp, fp]
    L.mpmhip_num_particles.argtypes = [vp]
    L.mpmhip_num_particles.restype = C.c_int64
    L.mpmhip_download.argtypes = [vp, C.c_int32, vp, C.c_int64]
    L.mpmhip_upload.argtypes = [vp, C.c_int32, vp, C.c_int64]
    ip = P(C.c_int32)
    L.mpmhip_set_partition.argtypes = [vp, C.c_int32, ip, ip, ip, ip, C.c_int32]
    L.mpmhip_set_halo.argtypes = [vp, C.c_int32, P(HaloBox)]
    L.mpmhip_leaver_counts.argtypes = [vp, C.c_int32, P(C.c_int64)]
    L.mpmhip_migration_scan.argtypes = [vp, C.c_int32, P(C.c_int64), ip, ip, P(C.c_float)]
    L.mpmhip_export_leavers.argtypes = [vp, C.c_int32, P(C.c_int64), vp]
    L.mpmhip_import_particles.argtypes = [vp, C.c_int64, vp]
    L.mpmhip_active_bounds.argtypes = [vp, ip, ip]
    L.mpmhip_set_overlap.argtypes = [vp, C.c_int32]
    L.mpmhip_tiled_run.argtypes = [vp, C.c_int64, C.c_int64, EXCHANGE_FN, vp]
    L.mpmhip_tiled_run.restype = C.c_int64
    bp = P(C.c_uint8)
    L.mpmhip_comm_unique_id.argtypes = [bp]
    L.mpmhip_comm_init.argtypes = [vp, bp, C.c_int32, C.c_int32]
    L.mpmhip_comm_destroy.argtypes = [vp]
    L.mpmhip_comm_selftest.argtypes = [vp]
    L.mpmhip_tiled_setup.argtypes = [vp, P(TiledConfig), ip, ip, ip]
    L.mpmhip_tiled_ipc_handle.argtypes = [vp, bp]
    L.mpmhip_tiled_ipc_connect.argtypes = [vp, bp]
    L.mpmhip_tiled_connect_local.argtypes = [P(vp), C.c_int32]
    L.mpmhip_tiled_advance.argtypes = [vp, C.c_int64]
    L.mpmhip_tiled_advance.restype = C.c_int64
    L.mpmhip_tiled_advance_group.argtypes = [P(vp), C.c_int32, C.c_int64]
    L.mpmhip_tiled_advance_group.restype = C.c_int64
    L.mpmhip_tiled_reduce.argtypes = [vp, P(C.c_double), C.c_int32, C.c_int32]
    L.mpmhip_tiled_reduce_group.argtypes = [P(vp), C.c_int32, P(C.c_double), C.c_int32, C.c_int32]
    L.mpmhip_calculate_energy_group.argtypes = [P(vp), C.c_int32, P(C.c_double), P(C.c_double)]
    L.mpmhip_tiled_totals.argtypes = [vp, P(C.c_int64)]
    L.mpmhip_tiled_totals_group.argtypes = [P(vp), C.c_int32, P(C.c_int64)]
    L.mpmhip_tiled_state.argtypes = [vp, P(C.c_int64)]
    L.mpmhip_tiled_plan.argtypes = [vp, C.c_int32, P(HaloBox)]
    L.mpmhip_num_slots.argtypes = [vp]
    L.mpmhip_num_slots.restype = C.c_int64
    L.mpmhip_reserve.argtypes = [vp, C.c_int64]
    L.mpmhip_capacity.argtypes = [vp]
    L.mpmhip_capacity.restype = C.c_int64
    for name in ("mpmhip_substep", "mpmhip_synchronize", "mpmhip_sort", "mpmhip_p2g", "mpmhip_grid_update",
                 "mpmhip_g2p", "mpmhip_profile_reset", "mpmhip_halo_pack", "mpmhip_substep_begin",
                 "mpmhip_substep_end", "mpmhip_substep_interior", "mpmhip_request_compaction"):
        getattr(L, name).argtypes = [vp]
    L.mpmhip_run_substeps.argtypes = [vp, C.c_int32]
    L.mpmhip_step.argtypes = [vp, C.c_float]
    L.mpmhip_current_time.argtypes = [vp]
    L.mpmhip_current_time.restype = C.c_double
    L.mpmhip_download_grid.argtypes = [vp, C.c_int32, fp]
    L.mpmhip_upload_grid.argtypes = [vp, fp]
    L.mpmhip_calculate_energy.argtypes = [vp, P(C.c_double), P(C.c_double)]
    L.mpmhip_delete_particles_inside_level_set.argtypes = [vp, P(C.c_int64)]
    fp = P(C.c_float)
    L.mpmhip_mpm88_create.argtypes = [C.c_int32, C.c_float, C.c_int32, C.c_int32, P(C.c_void_p)]
    L.mpmhip_mpm88_destroy.argtypes = [vp]
    L.mpmhip_mpm88_destroy.restype = None
    L.mpmhip_mpm88_last_error.argtypes = [vp]
    L.mpmhip_mpm88_last_error.restype = C.c_char_p
    L.mpmhip_mpm88_add.argtypes = [vp, C.c_int64, fp, fp, fp, fp, fp]
    L.mpmhip_mpm88_num_particles.argtypes = [vp]
    L.mpmhip_mpm88_num_particles.restype = C.c_int64
    L.mpmhip_mpm88_advance.argtypes = [vp, C.c_int32]
    L.mpmhip_mpm88_download.argtypes = [vp, fp, fp, fp, fp, fp]
    L.mpmhip_mpm88_download_grid.argtypes = [vp, fp]
    lp = P(C.c_int64)
    L.mpmhip_async_enable.argtypes = [vp, P(AsyncConfig)]
    L.mpmhip_async_update_dt_limits.argtypes = [vp]
    L.mpmhip_async_begin.argtypes = [vp, P(AsyncConfig)]
    L.mpmhip_async_pool_particles.argtypes = [vp]
    L.mpmhip_async_step.argtypes = [vp, C.c_float]
    L.mpmhip_async_load_pools.argtypes = [vp]
    L.mpmhip_async_state.argtypes = [vp, lp]
    L.mpmhip_async_current_time.argtypes = [vp]
    L.mpmhip_async_current_time.restype = C.c_double
    L.mpmhip_async_block_times.argtypes = [vp, C.c_int64, lp, lp, lp]
    L.mpmhip_async_block_times.restype = C.c_int64
    L.mpmhip_async_download_pools.argtypes = [vp, C.c_int64, fp, ip]
    L.mpmhip_async_download_pools.restype = C.c_int64
    L.mpmhip_async_profile.argtypes = [vp, C.c_int32, P(C.c_double)]
    L.mpmhip_async_snapshot_size.argtypes = [vp]
    L.mpmhip_async_snapshot_size.restype = C.c_int64
    L.mpmhip_async_snapshot_save.argtypes = [vp, vp, C.c_size_t]
    L.mpmhip_async_snapshot_load.argtypes = [vp, vp, C.c_size_t]
    L.mpmhip_host_particle_bytes.argtypes = [vp]
    L.mpmhip_host_particle_bytes.restype = C.c_int64
    L.mpmhip_async_blocks.argtypes = [vp, C.c_int64, ip, lp, lp, lp, lp, lp]
    L.mpmhip_async_blocks.restype = C.c_int64
    L.mpmhip_async_set_time_int.argtypes = [vp, C.c_int64]
    L.mpmhip_async_table.argtypes = [vp, ip, C.c_int64, lp, lp, lp, lp]
    L.mpmhip_async_table.restype = C.c_int64
    L.mpmhip_clear_particles.argtypes = [vp]
    L.mpmhip_set_dt.argtypes = [vp, C.c_float]
    L.mpmhip_set_time.argtypes = [vp, C.c_double]
    L.mpmhip_get_clock.argtypes = [vp, P(C.c_double), P(C.c_double), lp]
    L.mpmhip_set_clock.argtypes = [vp, C.c_double, C.c_double, C.c_int64]
    L.mpmhip_debug_allowed_dt.argtypes = [vp, C.c_int32, fp, C.c_int64, fp, fp, fp, C.c_float, fp]
    L.mpmhip2d_create.argtypes = [P(Config2D), P(vp)]
    L.mpmhip2d_destroy.argtypes = [vp]
    L.mpmhip2d_destroy.restype = None
    L.mpmhip2d_last_error.argtypes = [vp]
    L.mpmhip2d_last_error.restype = C.c_char_p
    L.mpmhip2d_set_levelset.argtypes = [vp, C.c_int32, P(Shape), C.c_int32, P(Shape), C.c_float, C.c_float, C.c_float]
    L.mpmhip2d_add_group.argtypes = [vp, C.c_int32, fp]
    L.mpmhip2d_add_particles.argtypes = [vp, C.c_int32, C.c_int64, fp, fp, fp, fp, fp]
    L.mpmhip2d_substep.argtypes = [vp]
    L.mpmhip2d_step.argtypes = [vp, C.c_float]
    L.mpmhip2d_current_time.argtypes = [vp]
    L.mpmhip2d_current_time.restype = C.c_double
    L.mpmhip2d_num_particles.argtypes = [vp]
    L.mpmhip2d_num_particles.restype = C.c_int64
    L.mpmhip2d_download.argtypes = [vp, C.c_int64, fp, fp, fp, fp, fp, ip, ip]
    L.mpmhip2d_download.restype = C.c_int64
    L.mpmhip2d_download_grid.argtypes = [vp, fp]
    L.mpmhip2d_bgeo_size.argtypes = [vp, C.c_int32, P(C.c_size_t)]
    L.mpmhip2d_bgeo_encode.argtypes = [vp, C.c_int32, vp, C.c_size_t, P(C.c_size_t)]
    L.mpmhip2d_write_bgeo.argtypes = [vp, C.c_char_p, C.c_int32]
    L.mpmhip2d_snapshot_size.argtypes = [vp]
    L.mpmhip2d_snapshot_size.restype = C.c_int64
    L.mpmhip2d_snapshot_save.argtypes = [vp, vp, C.c_size_t]
    L.mpmhip2d_snapshot_load.argtypes = [vp, vp, C.c_size_t]
    L.mpmhip2d_async_begin.argtypes = [vp, P(AsyncConfig)]
    L.mpmhip2d_async_pool_particles.argtypes = [vp]
    L.mpmhip2d_async_step.argtypes = [vp, C.c_float]
    L.mpmhip2d_async_load_pools.argtypes = [vp]
    L.mpmhip2d_async_load_pools.restype = C.c_int64
    L.mpmhip2d_async_view_blocks.argtypes = [vp, C.c_int64, ip]
    L.mpmhip2d_async_view_blocks.restype = C.c_int64
    L.mpmhip2d_async_state.argtypes = [vp, P(C.c_int64)]
    L.mpmhip2d_async_current_time.argtypes = [vp]
    L.mpmhip2d_async_current_time.restype = C.c_double
    L.mpmhip2d_async_table.argtypes = [vp, P(C.c_int32), C.c_int64] + [P(C.c_int64)] * 7
    L.mpmhip2d_async_table.restype = C.c_int64
    L.mpmhip_debug_cond_census.argtypes = [vp, P(C.c_double)]
    L.mpmhip_debug_scan_grid.argtypes = [C.c_int32, C.c_int32, C.c_int32, P(C.c_uint32), P(C.c_uint32)]
    L.mpmhip_debug_copy_bandwidth.argtypes = [vp, C.c_size_t, C.c_int32, P(C.c_double)]
    L.mpmhip_debug_gather_bandwidth.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, P(C.c_double)]
    L.mpmhip_bgeo_size.argtypes = [vp, C.c_int32, P(C.c_size_t)]
    L.mpmhip_bgeo_encode.argtypes = [vp, C.c_int32, C.c_void_p, C.c_size_t, P(C.c_size_t)]
    L.mpmhip_write_bgeo.argtypes = [vp, C.c_char_p, C.c_int32]
    L.mpmhip_snapshot_size.argtypes = [vp]
    L.mpmhip_snapshot_size.restype = C.c_int64
    L.mpmhip_snapshot_save.argtypes = [vp, vp, C.c_size_t]
    L.mpmhip_snapshot_load.argtypes = [vp, vp, C.c_size_t]
    L.mpmhip_set_profiling.argtypes = [vp, C.c_int32]
    L.mpmhip_set_profile_sampling.argtypes = [vp, C.c_int32]
    L.mpmhip_profile.argtypes = [vp, C.c_char_p, C.c_size_t]
    up = P(C.c_uint32)
    L.mpmhip_set_rigid_coupling.argtypes = [vp, C.c_float, C.c_float]
    L.mpmhip_add_rigid_body.argtypes = [vp, P(RigidConfig), C.c_int64, fp]
    L.mpmhip_num_rigid_bodies.argtypes = [vp]
    L.mpmhip_rigid_get_state.argtypes = [vp, C.c_int32, fp]
    L.mpmhip_rigid_set_velocity.argtypes = [vp, C.c_int32, fp, fp]
    L.mpmhip_rigid_get_samples.argtypes = [vp, C.c_int32, C.c_int64, fp, fp, ip]
    L.mpmhip_rigid_get_samples.restype = C.c_int64
    L.mpmhip_rigid_get_mesh.argtypes = [vp, C.c_int32, C.c_int64, fp]
    L.mpmhip_rigid_get_mesh.restype = C.c_int64
    L.mpmhip2d_rigid_get_mesh.argtypes = [vp, C.c_int32, C.c_int64, fp]
    L.mpmhip2d_rigid_get_mesh.restype = C.c_int64
    L.mpmhip_add_articulation.argtypes = [vp, P(JointConfig)]
    L.mpmhip_num_articulations.argtypes = [vp]
    L.mpmhip_num_articulations.restype = C.c_int32
    L.mpmhip_set_articulation_iterations.argtypes = [vp, C.c_int32]
    for name in ("mpmhip_rasterize_rigid_boundary", "mpmhip_gather_cdf", "mpmhip_advect_rigid_bodies", "mpmhip_articulate"):
        getattr(L, name).argtypes = [vp]
    L.mpmhip_download_cdf.argtypes = [vp, up, fp]
    L.mpmhip_download_boundary.argtypes = [vp, fp, C.c_int64]
    L.mpmhip_download_boundary.restype = C.c_int64
    L.mpmhip2d_set_rigid_coupling.argtypes = [vp, C.c_float, C.c_float]
    L.mpmhip2d_set_rigid_levelset_collision.argtypes = [vp, C.c_int32]
    L.mpmhip2d_add_articulation.argtypes = [vp, P(JointConfig)]
    L.mpmhip2d_set_articulation_iterations.argtypes = [vp, C.c_int32]
    L.mpmhip2d_add_rigid_body.argtypes = [vp, P(RigidConfig2D), C.c_int64, fp]
    L.mpmhip2d_rigid_get_state.argtypes = [vp, C.c_int32, fp]
    L.mpmhip2d_rigid_get_samples.argtypes = [vp, C.c_int32, C.c_int64, fp]
    L.mpmhip2d_rigid_get_samples.restype = C.c_int64
    L.mpmhip2d_cdf_phase.argtypes = [vp]
    L.mpmhip2d_download_cdf.argtypes = [vp, up, fp]
    L.mpmhip2d_download_colours.argtypes = [vp, C.c_int64, up, fp, fp, ip]
    L.mpmhip2d_download_colours.restype = C.c_int64
    L.mpmhip_debug_svd3.argtypes = [vp, C.c_int64, fp, fp, fp, fp]
    L.mpmhip_debug_force.argtypes = [vp, C.c_int32, fp, C.c_int64, fp, fp, fp]
    L.mpmhip_debug_plasticity.argtypes = [vp, C.c_int32, fp, C.c_int64, fp, fp, fp, fp]
    _lib = L
    return L
