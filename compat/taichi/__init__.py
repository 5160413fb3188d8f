"""`taichi` — the import name the reference's scene scripts use, bound to libmpmhip.

The scripts under the reference's scripts/ directory open with `import taichi as tc` and then call into the Python
package of the (un-vendored) legacy taichi: `tc.dynamics.MPM(...)`, `tc.Vector(...)`, `tc.set_gdb_trigger()`,
`tc.constant_function13(...)`, `tc.core.create_simulation3('mpm')` ...  (scripts/benchmark/benchmark_3d.py:1-27,
scripts/async/async_mpm.py:1-32).  This package offers those names over taichi_mpm_amd, so that such a script runs
UNMODIFIED with

    PYTHONPATH=<repo>/compat python <reference>/scripts/benchmark/benchmark_3d.py

Only the MPM path is behind it: what the legacy package offered for rendering, textures and meshes (`tc.Texture`,
`tc.SegmentMesh`, the renderer classes) raises `MPMError` when a script tries to USE it — out of scope (SURVEY
section 2, #15 / #17) — and imports as an inert name where the reference's driver only imports it
(scripts/async/async_mpm.py:4-8).

Knobs (environment, so that no line of a script has to change):
  TAICHI_MPM_NUM_FRAMES   upper bound of the frames `simulate()` runs (the scripts' default is 1000)
  TAICHI_MPM_OUTPUT       root of `tc.get_output_path()` (default ./taichi_outputs)
"""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:  # the alias directory alone is on PYTHONPATH: find the package it forwards to
    sys.path.insert(0, _REPO)

from taichi_mpm_amd.mpm import MPMError  # noqa: E402

from . import core, dynamics  # noqa: E402,F401
from .misc.util import (P, Vector, Vectori, constant_function, constant_function11, constant_function13,  # noqa: E402,F401
                        function11, function13, function_addresses)

tc_core = core.tc_core


def set_gdb_trigger(on=True):
    """legacy taichi: attach gdb on a crash (scripts/benchmark/benchmark_3d.py:6).  Nothing to arm here."""
    return None


def get_output_path(path, create=False):
    """`tc.get_output_path('async_mpm/' + task_id, True)` (scripts/async/async_mpm.py:46)"""
    full = os.path.join(os.environ.get("TAICHI_MPM_OUTPUT", os.path.join(os.getcwd(), "taichi_outputs")), path)
    if create:
        os.makedirs(full, exist_ok=True)
    return full


def clear_directory_with_suffix(directory, suffix):
    """scripts/async/async_mpm.py:211-215"""
    if os.path.isdir(directory):
        for f in os.listdir(directory):
            if f.endswith("." + suffix.lstrip(".")):
                os.remove(os.path.join(directory, f))


def duplicate_stdout_to_file(fn):  # scripts/async/async_mpm.py:56 (the log file of a run: not kept here)
    return None


def redirect_print_to_log():  # scripts/async/async_mpm.py:57
    return None


def trace(fmt, *args):  # scripts/async/async_mpm.py:58
    print(fmt.format(*args))


class _OutOfScope:
    """a name of the legacy package that scene scripts USE for scene tooling this build does not carry"""

    def __init__(self, name, what):
        self._name, self._what = name, what

    def __call__(self, *a, **k):
        raise MPMError("tc.%s: %s is scene tooling outside this build (SURVEY section 2: out of scope); give add_particles "
                       "explicit positions=, cube= or benchmark= instead" % (self._name, self._what))


Texture = _OutOfScope("Texture", "density textures / Poisson-disk sampling")
SegmentMesh = _OutOfScope("SegmentMesh", "2D segment meshes from the taichi core")
