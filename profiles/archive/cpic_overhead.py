"""What the CPIC coupling costs: 1 M jelly particles on the 128^3 grid (BASELINE configs[1] size) with and without a scripted
paddle wheel turning inside the block (120 k coloured particles).  Round 2, one MI355X: 0.138 ms per substep without, 0.47 ms
with the body.  The way there (each step measured): 39 ms with one float atomic per impulse on the body's six accumulator
words -> 0.75 ms with the impulses reduced per wave -> 0.53 ms with the pages handed out per boundary particle and
neighbouring lanes deduplicated (the rigid-page bitmap and the slot words are a few cache lines: every thread going there
queued at one L2 bank) -> 0.47 ms with gather_cdf per block from LDS and the transfers near bodies in three passes.
    python profiles/cpic_overhead.py"""
import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import taichi_mpm_amd as tm
from tests.test_gpu_cpic import paddle
from tests.common import lattice_cube
from oracle import oracle as orc
res, dx = 128, 1 / 128
x = lattice_cube(res, 39, 89, dx, jitter=0.15, seed=3)
gp, _ = orc.group_params("jelly", dx ** 3 / 8 * 400, dx ** 3 / 8)
for rigid in (False, True):
    sim = tm.create_simulation3("mpm").initialize(dict(res=(res,) * 3, delta_x=dx, base_delta_t=1e-4, max_particles=len(x) + 16))
    if rigid:
        sim.add_particles(dict(type="rigid", mesh=paddle(), codimensional=True, friction=-1.0, scripted_position=lambda t: (0.5, 0.5, 0.5),
                               scripted_rotation=lambda t: (0.0, 0.0, 720.0 * t)))
    sim.add_particles(dict(type="jelly", positions=x, params=gp))
    sim.run_substeps(20); sim.synchronize()
    t = time.time(); sim.run_substeps(200); sim.synchronize(); dt = (time.time() - t) / 200
    print("rigid" if rigid else "plain", "ms/substep %.4f" % (dt * 1e3), "coloured", int((sim.get_particles()["states"] != 0).sum()))
