"""oracle/bgeo.py — TEST INFRASTRUCTURE: numpy restatement of the .bgeo bytes the reference writes per frame.

Follows MPM<dim>::write_partio (src/visualize.cpp:17-100: attribute set and order, ascending-id particle order,
what each attribute holds) and Partio's writeBGEO (external/partio/src/io/BGEO.cpp:57-194: version-5 big-endian
layout).  PINNED: tests/test_bgeo_cpu.py checks it byte for byte against fixtures produced by the reference's own
Partio sources (oracle/_ref/partio_write, tests/golden/make_bgeo_golden.py)."""
import struct

import numpy as np

PLAIN = (("type", 1, 1), ("index", 1, 1), ("limit", 3, 1), ("v", 3, 5))  # name, count, houdini type (visualize.cpp:24-28)
VERBOSE = (("m", 1, 5), ("boundary_normal", 3, 5), ("debug", 3, 5), ("states", 1, 1), ("boundary_distance", 1, 0),
           ("near_boundary", 1, 1), ("apic_frobenius_norm", 1, 0))  # :30-38


def _hstr(s):
    return struct.pack(">h", len(s)) + s.encode()


def encode(x, v, ids, verbose=False, mass=None, debug=None, B=None):
    """x, v (n,3) f32; ids (n,) i32; verbose extras: mass (n,), debug (n,3), B (n,9) row-major apic_b.
    Without rigid bodies / async stepping: type 0, limit (1,1,1), boundary_normal 0, states 0, boundary_distance 0,
    near_boundary 0 (src/particles.h:92-99)."""
    n = len(ids)
    attrs = PLAIN + (VERBOSE if verbose else ())
    out = [struct.pack(">ici", int.from_bytes(b"Bgeo", "big"), b"V", 5), struct.pack(">iii", n, 1, 0),
           struct.pack(">iiiii", 0, len(attrs), 0, 1, 0)]  # BGEO.cpp:70-85
    for name, cnt, ht in attrs:  # :93-130
        out.append(_hstr(name) + struct.pack(">Hi", cnt, ht) + struct.pack(">%di" % cnt, *([0] * cnt)))
    order = np.argsort(np.asarray(ids), kind="stable")  # visualize.cpp:39-43
    width = 4 + sum(c for _, c, _ in attrs)
    rows = np.zeros((n, width), np.uint32)
    f = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)  # noqa: E731
    rows[:, 0:3] = f(np.asarray(x)[order])
    rows[:, 3] = np.float32(1.0).view(np.uint32)  # homogeneous coordinate, BGEO.cpp:158-160
    rows[:, 4] = 0  # type = is_rigid
    rows[:, 5] = np.asarray(ids, np.int32)[order].view(np.uint32)
    rows[:, 6:9] = 1  # dt_limit, stiffness_limit, cfl_limit
    rows[:, 9:12] = f(np.asarray(v)[order])
    if verbose:
        Bm = np.asarray(B, np.float32)[order].reshape(n, 3, 3)
        skew = np.float32(0.5) * (Bm - Bm.transpose(0, 2, 1))
        rows[:, 12] = f(np.asarray(mass)[order])
        rows[:, 16:19] = f(np.asarray(debug)[order])
        rows[:, 22] = f(np.sqrt((skew * skew).reshape(n, 9).sum(1, dtype=np.float32)))
    out.append(rows.astype(">u4").tobytes())
    out.append(_hstr("generator") + struct.pack(">hii", 1, 4, 1) + _hstr("papi"))  # :165-170
    out.append(struct.pack(">ii", 0x8000, n))
    out.append(np.arange(n).astype(">i4" if n > (1 << 16) else ">u2").tobytes())  # :175-180
    out.append(struct.pack(">i", 0) + b"\x00\xff")
    return b"".join(out)
