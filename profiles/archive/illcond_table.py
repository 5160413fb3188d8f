"""error of the device's constitutive path against the reference per condition number of F (tests/test_gpu_illcond.py holds the
assertions; this prints the whole table, for whichever library MPMHIP_LIB_VARIANT selects)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import taichi_mpm_amd as tm
from tests import test_gpu_illcond as T
from tests.common import lattice_cube

sim = tm.create_simulation3("mpm")
sim.initialize(dict(res=(32, 32, 32), delta_x=1 / 32, base_delta_t=1e-4))
sim.add_particles(dict(type="jelly", positions=lattice_cube(32, 10, 12, 1 / 32)))
sim._ensure_ctx()
g = np.load(os.path.join(ROOT, "tests", "golden", "ref_illcond.npz"))
print("# library variant: %r" % os.environ.get("MPMHIP_LIB_VARIANT", ""))
print("%-9s  %-7s %4s   %-9s %-9s %-9s" % ("material", "cond", "rows", "force", "F_new", "next force"))
for mat in T.MATS:
    for c, n, ef, eF, en in T.errors(g, mat, T.device_outputs(sim, g, mat)):
        print("%-9s  %-7g %4d   %.2e  %.2e  %.2e" % (mat, c, n, ef, eF, en))
F = np.ascontiguousarray(g["F"], np.float32); n = len(F)
U = np.zeros((n, 9), np.float32); S = np.zeros((n, 3), np.float32); V = np.zeros((n, 9), np.float32)
sim._check(sim._L.mpmhip_debug_svd3(sim._ctx, n, F.ctypes.data_as(T.FP), U.ctypes.data_as(T.FP), S.ctypes.data_as(T.FP), V.ctypes.data_as(T.FP)))
rel = np.abs(np.sort(np.abs(S.astype(np.float64)), 1)[:, ::-1] - np.abs(g["sigma"])) / np.abs(g["sigma"])
print("# relative error of the singular values (max / mid / min)")
for c, sel in T.cond_classes(g["cond"]):
    r = rel[sel].max(0)
    print("sigma      %-7g %4d   %.2e  %.2e  %.2e" % (c, sel.sum(), r[0], r[1], r[2]))
sim.close()
