"""The fused transfer (k_g2p<FUSED>, taichi_mpm_amd/csrc/k_g2p.h: G2P of substep n + P2G of substep n+1 in one kernel, the P2G
result in 8^3-node tiles summed by k_grid_fused) against the two-kernel path of the same library and against the oracle.

The two paths compute the same sums in different orders (P2G: the particles of a cell are accumulated per NEW cell from an
LDS list instead of per sorted cell from HBM records; the grid: eight 8^3 tiles per node instead of eight 6^3 ones), so they
agree to fp32 summation noise: after one substep x 2e-7 absolute, v / F 2e-6 relative; over tens of substeps the noise is
amplified by the dynamics like any rounding difference (bounds in the tests).

The fused transfer lost its A/B and is compiled only into the variant library lib/libmpmhip_fused.so
(-DMPMHIP_WITH_FUSED; __graft_entry__.build() builds it).  These tests run in a child interpreter with
MPMHIP_LIB_VARIANT=fused, where the env MPMHIP_FUSED (read when a ctx is created) switches between the two paths.
"""
import os
import subprocess
import sys
import numpy as np
import pytest

from tests.common import lattice_cube, make_state, rel_l2
from tests.test_gpu_parity import DT, DX, RES, make_sim, ocfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get("MPMHIP_LIB_VARIANT") != "fused":
    # parent: re-run this file under the variant library, one child per test function
    def test_fused_variant_suite():
        lib = os.path.join(ROOT, "taichi_mpm_amd", "lib", "libmpmhip_fused.so")
        if not os.path.exists(lib):
            pytest.skip("lib/libmpmhip_fused.so not built (python -c 'import __graft_entry__ as g; g.build()')")
        env = dict(os.environ, MPMHIP_LIB_VARIANT="fused")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    _CHILD = False
else:
    _CHILD = True


child_only = pytest.mark.skipif(not _CHILD, reason="runs in the child interpreter started by test_fused_variant_suite")


@pytest.fixture(scope="module")
def tm():
    import taichi_mpm_amd as tm
    tm.load()
    return tm


def _run(tm, monkeypatch, fused, state, steps, batches=1, **cfg):
    monkeypatch.setenv("MPMHIP_FUSED", "1" if fused else "0")
    sim = make_sim(tm, state, **cfg)
    for _ in range(batches):
        sim.run_substeps(steps)
    out = sim.get_particles()
    prof = sim.profile()
    sim.close()
    assert bool(prof.get("fused", 0)) == fused
    return out


def _close(a, b, xtol, rtol):
    assert np.array_equal(a["id"], b["id"])
    assert np.abs(a["x"] - b["x"]).max() <= xtol
    for f in ("v", "F"):
        assert rel_l2(a[f], b[f]) <= rtol, f
    assert np.allclose(a["aux"], b["aux"], rtol=0, atol=rtol * max(1.0, float(np.abs(b["aux"]).max())))


@child_only
@pytest.mark.parametrize("mat", ["jelly", "sand", "water", "snow", "visco"])
def test_fused_substeps_equal_the_two_kernel_path(tm, monkeypatch, mat):
    x = lattice_cube(RES, 9, 17, DX, jitter=0.25, seed=31)
    s = make_state(x, mat, DX, perturb_F=0.02, seed=32)
    for steps in (1, 2, 7):  # 1: classic P2G feeds the fused kernel; 2+: its own 8^3 tiles feed the next substep
        a = _run(tm, monkeypatch, True, s, steps)
        b = _run(tm, monkeypatch, False, s, steps)
        _close(a, b, 2e-7 * steps, 3e-6 * steps)
        assert rel_l2(a["B"], b["B"]) <= 2e-4  # (recovered from the stored P2G matrix in both runs)


@child_only
def test_fused_run_matches_the_oracle_over_ten_substeps(tm, orc, monkeypatch):
    x = lattice_cube(RES, 10, 16, DX, jitter=0.2, seed=41)
    s = make_state(x, "sand", DX, perturb_F=0.02, seed=42)
    got = _run(tm, monkeypatch, True, s, 10)
    cfg = ocfg(orc)
    for _ in range(10):
        orc.substep(cfg, s)
    o = np.argsort(got["id"], kind="stable")
    assert np.abs(got["x"][o] - s.x).max() <= 5e-6
    assert rel_l2(got["v"][o], s.v) <= 2e-4 and rel_l2(got["F"][o], s.F) <= 2e-4


@child_only
def test_fast_particles_leave_their_block_and_cells_every_substep(tm, monkeypatch):
    """0.4 cells per substep along a diagonal: most particles change cell within a few substeps and many leave their block
    (the leaver list of phase B); batches end in between (write_p) and start again from the stored tiles"""
    x = lattice_cube(RES, 10, 18, DX, jitter=0.3, seed=51)
    s = make_state(x, "jelly", DX, perturb_F=0.01, seed=52, vel_scale=0.2)
    s.v += np.float32(0.4 * DX / DT) * np.array([1.0, 0.6, -0.8], np.float32)
    a = _run(tm, monkeypatch, True, s, 6, batches=3, planes=[], clean_boundary=False)
    b = _run(tm, monkeypatch, False, s, 6, batches=3, planes=[], clean_boundary=False)
    _close(a, b, 5e-6, 1e-4)
    assert len(a["id"]) == len(x)
    moved = np.abs(a["x"][np.argsort(a["id"], kind="stable")] - s.x).max() / DX
    assert moved > 5.0  # it did cross blocks


@child_only
def test_crowded_blocks_take_several_rounds(tm, monkeypatch):
    """40 particles per cell = 2 560 per block: five staging rounds of 512 per block"""
    base = lattice_cube(RES, 12, 16, DX, jitter=0.3, seed=61)
    x = np.concatenate([base + np.float32(1e-3 * k) for k in range(5)])
    s = make_state(x, "sand", DX, perturb_F=0.01, seed=62)
    a = _run(tm, monkeypatch, True, s, 5)
    b = _run(tm, monkeypatch, False, s, 5)
    _close(a, b, 1e-6, 2e-5)


@child_only
def test_deletions_and_insertions_between_batches(tm, monkeypatch):
    """particles that reach the domain wall die inside the fused kernel (they must not scatter); new particles invalidate the
    stored tiles: the next substep starts from a classic P2G again"""
    out = {}
    for fused in (True, False):
        monkeypatch.setenv("MPMHIP_FUSED", "1" if fused else "0")
        xa = lattice_cube(RES, 8, 12, DX, jitter=0.2, seed=71)
        sa = make_state(xa, "jelly", DX, seed=72, vel_scale=0.1)
        sa.v += np.float32(0.45 * DX / DT) * np.array([-1.0, 0.0, 0.0], np.float32)  # towards x = 7 dx: deleted there
        sim = make_sim(tm, sa, planes=[])
        sim.run_substeps(5)
        n_mid = sim.get_num_particles()
        xb = lattice_cube(RES, 18, 22, DX, jitter=0.2, seed=73)
        sb = make_state(xb, "sand", DX, seed=74)
        sim.add_particles(dict(type="sand", positions=sb.x, velocities=sb.v, F=sb.F, B=sb.B, aux=sb.aux, params=sb.gparams[0]))
        sim.run_substeps(4)
        out[fused] = (n_mid, sim.get_particles())
        sim.close()
    assert out[True][0] == out[False][0] and out[True][0] < len(xa)  # some died, the same ones
    _close(out[True][1], out[False][1], 2e-6, 5e-5)


@child_only
def test_a_particle_faster_than_one_cell_per_substep_is_reported(tm, monkeypatch):
    """|v| dt > dx: the new base cell can lie outside the one-cell shell the fused kernel covers — a sticky error, not a
    silently wrong grid (a whole clump moves, or the gather would average the speed away)"""
    from taichi_mpm_amd.mpm import MPMError
    monkeypatch.setenv("MPMHIP_FUSED", "1")
    x = lattice_cube(RES, 12, 14, DX, seed=81)
    s = make_state(x, "jelly", DX, seed=82, vel_scale=0.0)
    s.v[:] = (1.6 * DX / DT, 0, 0)
    sim = make_sim(tm, s, planes=[], clean_boundary=False)
    with pytest.raises(MPMError, match="more than one cell"):
        sim.run_substeps(3)
        sim.synchronize()
    sim.close()
