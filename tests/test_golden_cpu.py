"""The committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle) must
keep matching the oracle: guards against silent oracle drift between rounds."""
import glob
import os

import numpy as np
import pytest

from tests.common import rel_l2

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "substep_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_golden(orc, path):
    g = np.load(path)
    cfg = orc.make_config(int(g["res"]), float(g["dx"]), float(g["dt"]), planes=g["planes"].tolist(),
                          friction=float(g["friction"]))
    s = orc.State(g["in_x"], g["in_v"], g["in_B"], g["in_F"], g["in_aux"], None, g["gparams"], g["gtype"])
    grid = orc.p2g(cfg, s)
    nz = g["nz"].astype(int)
    assert rel_l2(grid[nz[:, 0], nz[:, 1], nz[:, 2]], g["p2g_nz"]) < 1e-6
    assert np.count_nonzero(grid[..., 3]) == len(nz)
    orc.grid_update(cfg, grid)
    orc.g2p(cfg, s, grid)
    assert np.abs(s.x - g["out_x"]).max() < 1e-7
    assert rel_l2(s.v, g["out_v"]) < 1e-6 and rel_l2(s.F, g["out_F"]) < 1e-6
