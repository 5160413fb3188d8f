"""The .bgeo restatement (oracle/bgeo.py) against the bytes of the reference's own Partio writer (golden fixtures
made by tests/golden/make_bgeo_golden.py with oracle/_ref/partio_write), and the decoder round trip."""
import hashlib
import os

import numpy as np
import pytest

from oracle import bgeo as obgeo
from tests import bgeo_reader
from tests.bgeo_state import CASES, E, MASS, MATERIALS, debug_triple, make_state

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHA = dict((ln.split()[1], ln.split()[0]) for ln in open(os.path.join(GOLD, "bgeo_sha256.txt")))


def oracle_bytes(s, verbose):
    dbg = np.array([debug_triple(MATERIALS[g], a, E) for g, a in zip(s["gid"], s["aux"])], np.float32).reshape(-1, 3)
    return obgeo.encode(s["x"], s["v"], s["id"], verbose, MASS[s["gid"]], dbg, s["B"])


@pytest.mark.parametrize("name,n,seed", CASES)
@pytest.mark.parametrize("verbose", [False, True])
def test_oracle_matches_reference_partio_bytes(name, n, seed, verbose):
    b = oracle_bytes(make_state(n, seed), verbose)
    tag = "bgeo_%s_%s" % (name, "verbose" if verbose else "plain")
    assert hashlib.sha256(b).hexdigest() == SHA[tag]
    path = os.path.join(GOLD, tag + ".bgeo")
    if os.path.exists(path):
        assert b == open(path, "rb").read()


def test_reader_recovers_the_state_from_the_reference_bytes():
    s = make_state(37, 2)
    d = bgeo_reader.parse(open(os.path.join(GOLD, "bgeo_small_verbose.bgeo"), "rb").read())
    o = np.argsort(s["id"])
    assert d["n"] == 37 and np.array_equal(d["data"]["index"].ravel(), s["id"][o])
    assert np.array_equal(d["position"], s["x"][o]) and np.array_equal(d["data"]["v"], s["v"][o])
    assert np.array_equal(d["data"]["m"].ravel(), MASS[s["gid"]][o])
    assert np.all(d["data"]["limit"] == 1) and np.all(d["data"]["type"] == 0)
    assert np.array_equal(d["data"]["debug"][:, 1], np.array([4, 5, 6, 8], np.float32)[s["gid"]][o])
    Bm = s["B"][o].reshape(-1, 3, 3)
    want = np.sqrt(((0.5 * (Bm - Bm.transpose(0, 2, 1))) ** 2).reshape(-1, 9).sum(1))
    assert np.allclose(d["data"]["apic_frobenius_norm"].ravel(), want, rtol=1e-6)


def test_oracle_matches_the_reference_writer_run_live():
    """oracle/_ref/partio_write (the reference's Partio sources, compiled by `make -C oracle ref_partio`) on fresh
    states that are NOT among the committed fixtures; skipped where the binary has not been built"""
    tool = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "partio_write")
    if not os.path.exists(tool):
        pytest.skip("oracle/_ref/partio_write not built")
    from tests.golden.make_bgeo_golden import reference_bytes
    for n, seed in ((1, 21), (300, 22), (2000, 23)):
        s = make_state(n, seed)
        for verbose in (False, True):
            assert oracle_bytes(s, verbose) == reference_bytes(s, verbose, MASS, E)
