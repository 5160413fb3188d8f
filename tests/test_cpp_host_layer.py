"""The C++ host layer (include/mpm_amd/*.h: MPMKernel, particle registry, MPM<3>, MPM<2>, the asynchronous steppers) compiled with g++ against
libmpmhip.so.  CPU mode: the reference's kernel known-answer tests + registry + "no GPU => throws".  GPU mode:
MPM<3> end to end on the device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_layer.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "test_host_layer")


def _build():
    from taichi_mpm_amd import _lib
    lib = _lib.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC] + [os.path.join(ROOT, "include", "mpm_amd", h) for h in ("kernel.h", "particles.h", "mpm.h", "mpm2d.h")] + [lib]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        libdir = os.path.dirname(lib)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", OUT,
                               "-L", libdir, "-lmpmhip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib",
                               "-Wl,--allow-shlib-undefined"])
    return OUT


def test_cpp_host_layer_cpu():
    exe = _build()
    env = dict(os.environ)
    env.pop("MPMHIP_TEST_HAS_GPU", None)
    import torch
    if torch.cuda.is_available():
        env["MPMHIP_TEST_HAS_GPU"] = "1"
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_layer_gpu():
    exe = _build()
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
