"""`from taichi.misc.util import *` (scripts/async/async_mpm.py:3): P, Vector, Vectori, function_addresses, time ..."""
import time  # noqa: F401  (the reference's driver calls time.time() through this star import, scripts/async/async_mpm.py:143)

import numpy as np

import taichi  # noqa: F401  (the reference's driver says `taichi.clear_directory_with_suffix` after this star import, scripts/async/async_mpm.py:213)

__all__ = ["taichi", "P", "Vector", "Vectori", "function_addresses", "time", "constant_function", "constant_function11", "constant_function13",
           "function11", "function13", "image_buffer_to_ndarray"]


class Vector(tuple):
    """tc.Vector(x, y[, z]) — a tuple with .x / .y / .z"""

    def __new__(cls, *v):
        if len(v) == 1 and hasattr(v[0], "__len__"):
            v = tuple(v[0])
        return tuple.__new__(cls, (float(a) for a in v))

    x = property(lambda s: s[0])
    y = property(lambda s: s[1])
    z = property(lambda s: s[2])

    def __add__(self, o):
        return Vector(*(a + b for a, b in zip(self, o)))

    def __sub__(self, o):
        return Vector(*(a - b for a, b in zip(self, o)))

    def __mul__(self, s):
        return Vector(*(a * s for a in self)) if np.isscalar(s) else Vector(*(a * b for a, b in zip(self, s)))

    __rmul__ = __mul__


class Vectori(tuple):
    def __new__(cls, *v):
        if len(v) == 1 and hasattr(v[0], "__len__"):
            v = tuple(v[0])
        return tuple.__new__(cls, (int(a) for a in v))

    x = property(lambda s: s[0])
    y = property(lambda s: s[1])
    z = property(lambda s: s[2])


def P(**kwargs):
    """the config dict a simulation object takes (legacy taichi: string -> string `Config`; here the dict itself)"""
    return dict(kwargs)


# legacy taichi hands scripted motions to C++ as function addresses; here a script IS the Python callable (include/mpmhip.h:
# mpmhip_script_fn), so the "address" of a function is the function and the table only keeps them alive / indexable
function_addresses = []


def _register(f):
    function_addresses.append(f)
    return f


def function13(f):
    """t -> 3-vector (scripted_position / scripted_rotation of a rigid body, scripts/mls-cpic/*.py)"""
    return _register(lambda t: tuple(float(a) for a in f(t)))


def function11(f):
    return _register(lambda t: float(f(t)))


def constant_function13(v):
    v = tuple(float(a) for a in v)
    return _register(lambda t: v)


def constant_function11(v):
    return _register(lambda t: float(v))


def constant_function(v):
    """tc.constant_function((x, y, z)) / tc.constant_function(s)"""
    return constant_function13(v) if hasattr(v, "__len__") else constant_function11(v)


def image_buffer_to_ndarray(buf):
    return None
