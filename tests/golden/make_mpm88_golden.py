"""Generates tests/golden/ref_mpm88.npz: OUTPUT OF THE REFERENCE's 2D demo, /root/reference/mls-mpm88.cpp:16-69 advance(),
compiled where it lies into oracle/_ref/libmpm_ref.so (oracle/Makefile: ref_mpm, oracle/ref_mpm88_driver.cpp) —
BASELINE configs[0].  Run where /root/reference exists:  python tests/golden/make_mpm88_golden.py

Cases (the file's own constants: n = 80, dt = 1e-4):
  stir_<plastic|elastic>   ~8 k / ~2 k particles (three squares as add_object seeds them, mls-mpm88.cpp:70-78), random v / C,
                           perturbed F / Jp: inputs, particles and grid after ONE advance()
  fall                     1 500 particles from rest (F = I, C = 0, Jp = 1): particles after 40 advance() calls
What the compiled file pins: the whole of advance() incl. its kernels, stress, boundary rule and clamps.  Not reference
code: svd / polar_decomp of 2x2 matrices (the absent taichi core; the shim's are exact closed forms in double).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refmpm as ref  # noqa: E402


def seed(n_per, rng, stir):
    xs = [(rng.random((n_per, 2)) * 2 - 1) * 0.08 + np.array(c) for c in ((0.55, 0.45), (0.45, 0.65), (0.55, 0.85))]
    x = np.concatenate(xs).astype(np.float32)
    n = len(x)
    v = rng.normal(0, stir, (n, 2)).astype(np.float32)
    F = (np.tile(np.eye(2).reshape(1, 4), (n, 1)) + rng.normal(0, 0.02 if stir else 0.0, (n, 4))).astype(np.float32)
    Cm = rng.normal(0, stir, (n, 4)).astype(np.float32)
    Jp = (1.0 + rng.normal(0, 0.02 if stir else 0.0, n)).astype(np.float32)
    return x, v, F, Cm, Jp


def main():
    assert ref.mpm88_available(), "build oracle/_ref/libmpm_ref.so first (make -C oracle ref_mpm)"
    out = {}
    for name, plastic, n_per in (("stir_plastic", True, 2667), ("stir_elastic", False, 667)):
        s = seed(n_per, np.random.default_rng(88), 1.0)
        for k, a in zip("xvFCJ", s):
            out["%s_in_%s" % (name, k)] = a.copy()
        grid = ref.mpm88_advance(*s, steps=1, plastic=plastic)
        for k, a in zip("xvFCJ", s):
            out["%s_out_%s" % (name, k)] = a
        out["%s_grid" % name] = grid
    s = seed(500, np.random.default_rng(89), 0.0)
    out["fall_in_x"] = s[0].copy()
    ref.mpm88_advance(*s, steps=40, plastic=True)
    for k, a in zip("xvFCJ", s):
        out["fall_out_%s" % k] = a
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_mpm88.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
