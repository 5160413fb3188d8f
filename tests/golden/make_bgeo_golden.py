"""Generates tests/golden/bgeo_*.bgeo and bgeo_sha256.txt with the REFERENCE's own Partio writer.  Run from the repo root, in the
container that has /root/reference:
    make -C oracle ref_partio && python tests/golden/make_bgeo_golden.py
oracle/_ref/partio_write = the reference's vendored Partio sources (external/partio) compiled in place + our driver
that repeats MPM::write_partio's attribute sequence (src/visualize.cpp:17-100).  The particle state is an
integer-hash function of (n, seed) (tests/bgeo_state.py: no RNG stream to drift); the fixtures hold the bytes the
reference writes for it (.bgeo for the small cases, the sha256 of every case including the large ones that cross the
65536-point switch of the primitive's index width) — they PIN the .bgeo encoder of libmpmhip byte for byte."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(ROOT, "oracle", "_ref", "partio_write")

sys.path.insert(0, ROOT)

from tests.bgeo_state import CASES, E, MASS, MATERIALS, debug_triple, make_state  # noqa: E402


def reference_bytes(s, verbose, mass, E):
    n = len(s["id"])
    with tempfile.TemporaryDirectory() as d:
        raw, out = os.path.join(d, "in.raw"), os.path.join(d, "out.bgeo")
        with open(raw, "wb") as f:
            f.write(np.array([n, int(verbose)], np.int32).tobytes())
            for i in range(n):
                f.write(s["x"][i].tobytes() + s["v"][i].tobytes())
                f.write(np.array([s["id"][i], 0, 1, 1, 1], np.int32).tobytes())  # is_rigid 0, limits 1,1,1
                if verbose:
                    mat = MATERIALS[s["gid"][i]]
                    f.write(np.float32(mass[s["gid"][i]]).tobytes() + np.zeros(3, np.float32).tobytes())
                    f.write(np.array(debug_triple(mat, s["aux"][i], E), np.float32).tobytes())
                    f.write(np.int32(0).tobytes() + np.float32(0).tobytes() + np.int32(0).tobytes())
                    f.write(s["B"][i].tobytes())
        subprocess.check_call([TOOL, raw, out])
        return open(out, "rb").read()


def main():
    if not os.path.exists(TOOL):
        sys.exit("build the reference writer first: make -C oracle ref_partio")
    mass = MASS
    sha = {}
    for name, n, seed in CASES:
        s = make_state(n, seed)
        for verbose in (False, True):
            b = reference_bytes(s, verbose, mass, E)
            tag = "bgeo_%s_%s" % (name, "verbose" if verbose else "plain")
            sha[tag] = hashlib.sha256(b).hexdigest()
            if n < 1000:
                open(os.path.join(HERE, tag + ".bgeo"), "wb").write(b)
            print(tag, len(b), sha[tag])
    with open(os.path.join(HERE, "bgeo_sha256.txt"), "w") as f:
        for k in sorted(sha):
            f.write("%s %s\n" % (sha[k], k))


if __name__ == "__main__":
    main()
