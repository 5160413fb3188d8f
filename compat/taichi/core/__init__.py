"""`taichi.core` / `tc_core`: the pybind11 module of the legacy taichi core as far as the MPM scripts reach into it
(scripts/async/async_mpm.py:1,25-32,119-127,245)."""
import json

import taichi_mpm_amd as _tm
from taichi_mpm_amd.mpm import DynamicLevelSet as _DynamicLevelSet

from ..misc.util import Vector, Vectori

_SIMS = []  # simulations created through this module (print_profile_info reports on them)


def create_simulation3(name):
    """tc_core.create_simulation3('mpm' | 'async_mpm') (TC_IMPLEMENTATION(Simulation3D, ...), src/mpm.cpp:986-988)"""
    s = _tm.create_simulation3(name)
    _SIMS.append(s)
    return s


def create_simulation2(name):
    s = _tm.create_simulation2(name)
    _SIMS.append(s)
    return s


def print_profile_info():
    """tc.core.print_profile_info() (scripts/async/async_mpm.py:245): the phase table of every live simulation"""
    for s in _SIMS:
        if hasattr(s, "profile") and getattr(s, "_ctx", None) is not None:
            print(json.dumps(s.profile(reset=True)))


class DynamicLevelSet3D(_DynamicLevelSet):
    pass


class DynamicLevelSet2D(_DynamicLevelSet):
    pass


def Vector2f(*v):
    return Vector(*v) if len(v) != 1 else Vector(v[0], v[0])


def Vector3f(*v):
    return Vector(*v) if len(v) != 1 else Vector(v[0], v[0], v[0])


Vector2i, Vector3i = Vectori, Vectori


def Array2DVector3(res, value):
    """image buffer of the renderer (scripts/async/async_mpm.py:159-161): never drawn into here"""
    return None


class _Core:
    """the module as an object: `from taichi.core import tc_core`"""
    create_simulation2 = staticmethod(create_simulation2)
    create_simulation3 = staticmethod(create_simulation3)
    print_profile_info = staticmethod(print_profile_info)
    DynamicLevelSet2D, DynamicLevelSet3D = DynamicLevelSet2D, DynamicLevelSet3D
    Vector2f, Vector3f, Vector2i, Vector3i = staticmethod(Vector2f), staticmethod(Vector3f), staticmethod(Vector2i), staticmethod(Vector3i)
    Array2DVector3 = staticmethod(Array2DVector3)


tc_core = _Core()
