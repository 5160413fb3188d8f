"""Deterministic particle states for the .bgeo fixtures: pure integer hashing (no RNG stream that could change
between numpy versions), shared by tests/golden/make_bgeo_golden.py (reference side) and the GPU tests."""
import numpy as np

MATERIALS = ("jelly", "water", "sand", "elastic")  # debug triple differs per material (src/particles.cpp:157..839)
MASS = np.array([1.5e-3, 2.5e-3, 1.25e-3, 0.75e-3], np.float32)
VOL = np.float32(3.0e-6)
E = 5000.0
CASES = (("empty", 0, 1), ("small", 37, 2), ("at_switch", 1 << 16, 3), ("past_switch", (1 << 16) + 1, 4))


def _hash(i, salt):
    """splitmix64-style finaliser on uint64 arrays -> uint64"""
    with np.errstate(over="ignore"):
        z = (i.astype(np.uint64) + np.uint64(salt) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _unit(i, salt):
    """exact multiples of 2^-20 in [0, 1) as float32"""
    return ((_hash(i, salt) >> np.uint64(44)).astype(np.float64) / float(1 << 20)).astype(np.float32)


def make_state(n, seed):
    """particle state in the unit box; ids = a scrambled, gappy set (deleted particles leave holes)"""
    i = np.arange(n, dtype=np.uint64)
    order = np.argsort(_hash(i, 100 + seed), kind="stable")
    ids = (3 * order + (order % 2)).astype(np.int32)  # unique, non-contiguous, NOT ascending in storage order
    x = np.stack([0.3 + 0.4 * _unit(i, 10 * seed + k) for k in range(3)], 1).astype(np.float32)
    v = np.stack([4.0 * _unit(i, 10 * seed + 3 + k) - 2.0 for k in range(3)], 1).astype(np.float32)
    # dyadic entries: || 0.5 (B - B^T) ||_F^2 is exact in fp32 whatever the summation order
    B = np.stack([((_hash(i, 10 * seed + 6 + k) % np.uint64(17)).astype(np.float32) - 8.0) / 16.0 for k in range(9)], 1)
    gid = (_hash(i, 900 + seed) % np.uint64(len(MATERIALS))).astype(np.int32)
    aux = (0.9 + 0.2 * _unit(i, 950 + seed)).astype(np.float32)
    return dict(id=ids, x=x, v=v, B=B.astype(np.float32), gid=gid, aux=aux)


def debug_triple(material, aux, E):
    """get_debug_info(): (0, y, 0) with y the material number; water (j, 5, sticky=0); elastic (E, 8, 0)"""
    y = {"visco": 1, "snow": 2, "linear": 3, "jelly": 4, "water": 5, "sand": 6, "von_mises": 7, "elastic": 8}[material]
    return (aux if material == "water" else E if material == "elastic" else 0.0, float(y), 0.0)
