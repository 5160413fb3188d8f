"""Generates tests/golden/spgrid_keys.txt with the REFERENCE's own SPGrid mask header.  Run from the repo root, in
the container that has /root/reference:
    make -C oracle ref_spgrid && python tests/golden/make_spgrid_golden.py
Each line: i j k key, key = SparseMask::Linear_Offset(i, j, k) >> data_bits for SPGrid_Mask<5, 5, 3> — what
sort_particles_and_populate_grid sorts particles by (src/mpm.cpp:785-790).  Coordinates: the corners of the in-block
ranges, the first blocks along every axis, and hash-scattered nodes of a 512^3 virtual grid (C3's spgrid_size)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.bgeo_state import _hash  # noqa: E402


def coords():
    c = [(i, j, k) for i in (0, 1, 3, 4, 5, 8) for j in (0, 1, 3, 4, 7) for k in (0, 1, 7, 8, 9, 16)]
    i = np.arange(400, dtype=np.uint64)
    c += list(zip(*[(_hash(i, 40 + a) % np.uint64(512)).astype(int).tolist() for a in range(3)]))
    c += [(511, 511, 511), (256, 0, 0), (0, 256, 0), (0, 0, 256), (255, 255, 255)]
    return c


def main():
    tool = os.path.join(ROOT, "oracle", "_ref", "spgrid_keys")
    inp = "".join("%d %d %d\n" % t for t in coords())
    out = subprocess.run([tool], input=inp, capture_output=True, text=True, check=True).stdout
    rows = [ln.split() for ln in out.splitlines()]
    assert all(r[4:] == ["2", "2", "3"] for r in rows)  # block = 4 x 4 x 8 nodes
    with open(os.path.join(ROOT, "tests", "golden", "spgrid_keys.txt"), "w") as f:
        for r in rows:
            f.write(" ".join(r[:4]) + "\n")
    print(len(rows), "keys")


if __name__ == "__main__":
    main()
