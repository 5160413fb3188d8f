#!/bin/bash
# profiles/run_to_run.sh — on the GPU box: how far two runs of the same small sand scene are apart (ranks inside a cell are handed out by atomics, so the in-cell summation order of P2G differs from run to run) — packed G2P walk against per-block walk, and per-block against itself; calibrates the tolerance of test_packed_g2p_walk_equals_the_per_block_walk
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'P'
import os, numpy as np, importlib, sys
sys.path.insert(0, "tests")
import taichi_mpm_amd as tm
import test_gpu_parity as T
RES, DX = T.RES, T.DX
rng = np.random.default_rng(31)
dense = T.lattice_cube(RES, 8, 14, DX, jitter=0.2, seed=30)
spray = (rng.uniform(8.0, 24.0, (3000, 3)) * DX).astype(np.float32)
x = np.concatenate([dense, spray])
def run(knob):
    os.environ["MPMHIP_G2P_PACKED"] = knob
    s = T.make_state(x, "sand", DX, perturb_F=0.02, seed=32)
    s.v[-40:] = (0.0, 0.0, 500.0)
    sim = T.make_sim(tm, s)
    for _ in range(5):
        sim.substep()
    out = sim.get_particles(); sim.close(); return out
worst = {f: [] for f in ("x", "v", "F", "aux")}
for it in range(40):
    a, b = run("0"), run("1" if it % 2 == 0 else "0")
    for f in worst:
        worst[f].append(float(np.abs(a[f] - b[f]).max()))
for f in worst:
    w = np.array(worst[f]); print(f, "packed-vs-block max %.3g  block-vs-block max %.3g  median %.3g" % (w[0::2].max(), w[1::2].max(), np.median(w)))
P
python - <<'P'
# the scenes of test_every_form_of_the_sort_gives_the_same_substep and test_crowded_cells_take_the_side_array_of_the_sort (jelly)
import os, numpy as np, sys
sys.path.insert(0, "tests")
import taichi_mpm_amd as tm
import test_gpu_parity as T
RES, DX = T.RES, T.DX
os.environ.pop("MPMHIP_G2P_PACKED", None)
rng = np.random.default_rng(41)
dense = T.lattice_cube(RES, 8, 14, DX, jitter=0.2, seed=40)
spray = (rng.uniform(6.0, 26.0, (2500, 3)) * DX).astype(np.float32)
x = np.concatenate([dense, spray])
def run_a():
    s = T.make_state(x, "jelly", DX, perturb_F=0.02, seed=42)
    s.v[-30:] = (0.0, 0.0, 500.0)
    sim = T.make_sim(tm, s)
    for _ in range(4):
        sim.substep()
    out = sim.get_particles(); sim.close(); return out
xc = T.lattice_cube(RES, 9, 15, DX, jitter=0.3, seed=21)
xc = np.concatenate([xc, xc + np.float32(1e-3), xc - np.float32(1e-3)])
def run_b():
    s = T.make_state(xc, "jelly", DX, perturb_F=0.02, seed=23)
    sim = T.make_sim(tm, s)
    for _ in range(3):
        sim.substep()
    out = sim.get_particles(); sim.close(); return out
for name, run in (("every_form scene", run_a), ("crowded scene", run_b)):
    ref = run()
    worst = {f: 0.0 for f in ("x", "v", "F")}
    for it in range(30):
        b = run()
        for f in worst:
            worst[f] = max(worst[f], float(np.abs(ref[f] - b[f]).max()) / max(1.0, float(np.abs(ref[f]).max())))
    print(name, {f: "%.3g" % w for f, w in worst.items()}, "(relative to max(1, |field|max); the tests allow 2e-6)")
P
