"""cost of MPM::articulate on the device (k_articulate: one lane walks the joints' Gauss-Seidel chain, 100 sweeps):
    python profiles/joint_cost.py
three free boxes (tests/cpic_scenes.py), joints set up, 40 substeps of drift so that every anchor pair is apart, then the
phase is timed on its own for 1, 2 and 4 hinges and for the four-joint chain of the parity tests."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_mpm_amd as tm  # noqa: E402
from tests import cpic_scenes as cs  # noqa: E402


def scene(joints):
    sim = tm.create_simulation3("mpm").initialize(dict(res=(32,) * 3, delta_x=1 / 32, base_delta_t=cs.JOINT_DT, gravity=(0, -10, 0),
                                                       max_particles=1 << 12))
    for body in cs.JOINT_BODIES:
        sim.add_particles(dict(type="rigid", **body))
    for b, (v, w) in enumerate(cs.JOINT_VELOCITIES):
        sim.set_rigid_velocity(b + 1, v, w)
    for j in joints:
        sim.general_action(dict(action="add_articulation", **j))
    for _ in range(cs.JOINT_DRIFT):
        sim.advect_rigid_bodies()
    return sim


hinge = dict(type="axial_rotation", obj0=1, obj1=2, axis=(0.0, 0.0, 1.0), offset0=(0.1, 0.0, 0.0))
for name, joints in (("1 hinge", [hinge]), ("2 hinges", [hinge, dict(hinge, obj0=2, obj1=3)]),
                     ("4 hinges", [hinge, dict(hinge, obj0=2, obj1=3), dict(hinge, obj0=1, obj1=3), dict(hinge, obj0=3, obj1=0)]),
                     ("chain of the parity test", cs.JOINT_CASES["chain"])):
    sim = scene(joints)
    for _ in range(5):
        sim.articulate()
    sim.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        sim.articulate()
    sim.synchronize()
    print("%-26s %.1f us per articulate (100 sweeps)" % (name, (time.perf_counter() - t0) / n * 1e6))
    sim.close()
