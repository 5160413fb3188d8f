"""Generates tests/golden/*.npz from the CPU oracle (seeded).  Run from the repo root:
    python tests/golden/make_golden.py
The reference holds no golden vectors for P2G/G2P ("parity unpinned") and cannot be built or
imported here, so these fixtures pin the ORACLE's output; they let the GPU parity tests run on a box
without depending on the oracle's floating-point environment, and they detect accidental oracle drift."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from tests.common import lattice_cube, make_state  # noqa: E402

RES, DX, DT = 32, 1.0 / 32, 1e-4
PLANES = [(0.0, 1.0, 0.0, -0.3)]
FRICTION = 0.4


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # default: every registered particle type.  `python tests/golden/make_golden.py visco linear` regenerates a subset
    for mat in (sys.argv[1:] or ("jelly", "snow", "sand", "water", "linear", "elastic", "von_mises", "visco")):
        x = lattice_cube(RES, 9, 15, DX, jitter=0.2, seed=11)
        s = make_state(x, mat, DX, perturb_F=0.02, seed=12)
        cfg = orc.make_config(RES, DX, DT, planes=PLANES, friction=FRICTION)
        inp = s.copy()
        a = s.copy()
        g_p2g = orc.p2g(cfg, a)
        g_upd = orc.grid_update(cfg, g_p2g.copy())
        orc.g2p(cfg, a, g_upd)
        nz = np.argwhere(g_p2g[..., 3] != 0)
        b = s.copy()
        for _ in range(5):
            orc.substep(cfg, b)
        np.savez_compressed(
            os.path.join(here, "substep_%s.npz" % mat), res=RES, dx=DX, dt=DT, planes=np.array(PLANES, np.float32),
            friction=FRICTION, gparams=inp.gparams, gtype=inp.gtype,
            in_x=inp.x, in_v=inp.v, in_B=inp.B, in_F=inp.F, in_aux=inp.aux,
            nz=nz.astype(np.int16), p2g_nz=g_p2g[nz[:, 0], nz[:, 1], nz[:, 2]], upd_nz=g_upd[nz[:, 0], nz[:, 1], nz[:, 2]],
            out_x=a.x, out_v=a.v, out_B=a.B, out_F=a.F, out_aux=a.aux,
            out5_x=b.x, out5_v=b.v, out5_F=b.F, out5_aux=b.aux, out5_ids=b.ids)
        print(mat, s.n, "particles,", len(nz), "nodes")


if __name__ == "__main__":
    main()
