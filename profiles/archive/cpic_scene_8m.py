"""CPIC at the scale of the headline workload: 8 M sand particles on the 256^3 grid with a scripted paddle wheel (two blades of
0.5 x 0.4) turning inside the block and a small free shell falling onto it — 42.5 k boundary particles, 0.71 M coloured
material particles.  Round 2, one MI355X: 1.81 ms per substep (0.59 ms without bodies); rocprofv3: k_p2g_rigid 0.56,
k_g2p (blocks away from bodies) 0.37, k_g2p_rigid 0.34, k_p2g 0.15, k_gather_cdf 0.15, k_cdf_alloc + k_cdf_rasterize 0.09 ms.
Round 3: 1.52 ms — k_p2g_rigid 0.38 (its impulse part moved behind the scatter: 43 -> 7 spilled registers; the bodies' kinematic
fields mirrored in LDS), k_g2p_rigid 0.32, k_g2p 0.30, k_gather_cdf 0.15, k_p2g 0.14, CDF 0.09; later in round 3 1.30-1.33 ms —
flagged blocks as a list, per-material kernels, boundary particles listed for the impulse pass: k_p2g_rigid 0.25, k_g2p_rigid 0.25;
then 1.17 ms with the colour-aware kernels on a second stream beside the plain ones (profiles/r03_p_cpic.txt).
    python profiles/cpic_scene_8m.py"""
import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import taichi_mpm_amd as tm
from tests.test_gpu_cpic import paddle
res = 256
sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=1 / res, base_delta_t=1e-4, gravity=(0, -10, 0), max_particles=8_100_000, penalty=1e4))
ls = tm.mpm.LevelSet(friction=-1.0).add_plane((0, 1, 0), d=-0.1)
sim.set_levelset(ls)
rid = int(sim.add_particles(dict(type="rigid", mesh=paddle(0.25, 0.2), codimensional=True, friction=-2, scripted_position=lambda t: (0.5, 0.5, 0.5),
                                 scripted_rotation=lambda t: (0.0, 0.0, 360.0 * t))))
free = int(sim.add_particles(dict(type="rigid", mesh=paddle(0.08, 0.08) * 1.0, codimensional=True, friction=0.3, density=100.0,
                                  initial_position=(0.5, 0.82, 0.5), initial_rotation=(10, 20, 30))))
sim.add_particles(dict(type="sand", cube_lo=(78, 78, 78), cube_cells=100))
print("particles", sim.get_num_particles(), "samples", len(sim.get_rigid_samples()["pos"]))
sim.run_substeps(20); sim.synchronize()
t = time.time(); sim.run_substeps(200); sim.synchronize(); dt = (time.time() - t) / 200
p = sim.get_particles()
print("ms/substep %.3f" % (dt * 1e3), "alive", len(p["x"]), "coloured", int((p["states"] != 0).sum()), "finite", bool(np.isfinite(p["x"]).all()),
      "max |v|", float(np.abs(p["v"]).max()))
print("scripted", sim.get_rigid_state(rid)["rotation"], "free body pos", sim.get_rigid_state(free)["position"], "vel", sim.get_rigid_state(free)["velocity"])
