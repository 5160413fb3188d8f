"""Loader / ctypes prototypes for libmpmhip.so (the C ABI declared in include/mpmhip.h)."""
import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_HERE, "csrc", "mpmhip.hip")
_DEPS = [_SRC, os.path.join(_HERE, "csrc", "mpm_math.h"), os.path.join(_ROOT, "include", "mpmhip.h")]
_LIBDIR = os.path.join(_HERE, "lib")
_LIB = os.path.join(_LIBDIR, "libmpmhip.so")

NPARAM = 16

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-value"]


class Config(C.Structure):
    """mirror of mpmhip_config"""
    _fields_ = [("res", C.c_int32 * 3), ("dx", C.c_float), ("dt", C.c_float), ("gravity", C.c_float * 3),
                ("particle_gravity", C.c_int32), ("apic_damping", C.c_float), ("rpic_damping", C.c_float),
                ("clean_boundary", C.c_int32), ("n_planes", C.c_int32), ("planes", (C.c_float * 4) * 8),
                ("friction", C.c_float), ("max_particles", C.c_int64), ("max_blocks", C.c_int64),
                ("device", C.c_int32), ("reorder_interval", C.c_int32), ("discard_apic_b", C.c_int32),
                ("reserved", C.c_int32 * 5)]


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


_lib = None

_SYMBOLS = ["mpmhip_abi_version", "mpmhip_create", "mpmhip_destroy", "mpmhip_last_error", "mpmhip_set_stream", "mpmhip_set_levelset",
            "mpmhip_add_group", "mpmhip_add_particles", "mpmhip_num_particles", "mpmhip_download",
            "mpmhip_upload", "mpmhip_substep", "mpmhip_run_substeps", "mpmhip_step", "mpmhip_current_time",
            "mpmhip_synchronize", "mpmhip_sort", "mpmhip_p2g", "mpmhip_grid_update", "mpmhip_g2p",
            "mpmhip_download_grid", "mpmhip_upload_grid", "mpmhip_set_profiling", "mpmhip_profile",
            "mpmhip_profile_reset", "mpmhip_debug_svd3", "mpmhip_debug_force", "mpmhip_debug_plasticity"]


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
    L = C.CDLL(_LIB)
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
    L.mpmhip_add_group.argtypes = [vp, C.c_int32, fp]
    L.mpmhip_add_particles.argtypes = [vp, C.c_int32, C.c_int64, fp, fp, fp, fp, fp]
    L.mpmhip_num_particles.argtypes = [vp]
    L.mpmhip_num_particles.restype = C.c_int64
    L.mpmhip_download.argtypes = [vp, C.c_int32, vp, C.c_int64]
    L.mpmhip_upload.argtypes = [vp, C.c_int32, vp, C.c_int64]
    for name in ("mpmhip_substep", "mpmhip_synchronize", "mpmhip_sort", "mpmhip_p2g", "mpmhip_grid_update",
                 "mpmhip_g2p", "mpmhip_profile_reset"):
        getattr(L, name).argtypes = [vp]
    L.mpmhip_run_substeps.argtypes = [vp, C.c_int32]
    L.mpmhip_step.argtypes = [vp, C.c_float]
    L.mpmhip_current_time.argtypes = [vp]
    L.mpmhip_current_time.restype = C.c_double
    L.mpmhip_download_grid.argtypes = [vp, C.c_int32, fp]
    L.mpmhip_upload_grid.argtypes = [vp, fp]
    L.mpmhip_set_profiling.argtypes = [vp, C.c_int32]
    L.mpmhip_profile.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.mpmhip_debug_svd3.argtypes = [vp, C.c_int64, fp, fp, fp, fp]
    L.mpmhip_debug_force.argtypes = [vp, C.c_int32, fp, C.c_int64, fp, fp, fp]
    L.mpmhip_debug_plasticity.argtypes = [vp, C.c_int32, fp, C.c_int64, fp, fp, fp, fp]
    _lib = L
    return L
