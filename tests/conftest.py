import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count():
    """number of HIP devices, probed in a child process: loading a HIP runtime here would clash with the copy torch
    bundles (taichi_mpm_amd/_lib.py: torch must load its runtime first)"""
    import subprocess
    code = ("import ctypes\n"
            "n = ctypes.c_int(0)\n"
            "for name in ('libamdhip64.so', '/opt/rocm/lib/libamdhip64.so'):\n"
            "    try:\n"
            "        hip = ctypes.CDLL(name)\n"
            "    except OSError:\n"
            "        continue\n"
            "    print(n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0)\n"
            "    break\n"
            "else:\n"
            "    print(0)\n")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip()
        return int(out.splitlines()[-1]) if out else 0
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a device and the built library: a plain `pytest tests` on a CPU box skips them instead of
    failing in the first ctx creation (the product path itself still fails loudly without a GPU)."""
    if not any("gpu" in it.keywords for it in items):
        return
    lib = os.path.join(ROOT, "taichi_mpm_amd", "lib", "libmpmhip.so")
    why = None
    if not os.path.exists(lib):
        why = "taichi_mpm_amd/lib/libmpmhip.so is not built"
    elif _hip_device_count() == 0:
        why = "no HIP device"
    if why:
        skip = pytest.mark.skip(reason=why)
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
