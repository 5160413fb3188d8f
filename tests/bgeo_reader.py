"""Decoder for the Houdini .bgeo (version 5, big-endian) particle files Partio writes — test infrastructure.

The reference's vendored Partio has its reader stubbed out (external/partio/src/io/BGEO.cpp `readBGEO` is empty), so
this restates the layout from the writer (`writeBGEO`, same file): header, point-attribute table, one row of
big-endian words per particle (x, y, z, w=1, then the attributes in table order), the "generator"/"papi" primitive
attribute, one particle-system primitive listing every point (u16 indices up to 65536 points, i32 above), 0x00 0xff."""
import struct

import numpy as np

HOUDINI_FLOAT, HOUDINI_INT, HOUDINI_VECTOR = 0, 1, 5


def parse(buf):
    """-> dict(n, attrs=[(name, count, houdini_type)], data={name: array (n, count)}, position (n, 3))"""
    o = 0

    def take(fmt):
        nonlocal o
        v = struct.unpack_from(">" + fmt, buf, o)
        o += struct.calcsize(">" + fmt)
        return v

    def hstr():
        nonlocal o
        (ln,) = take("h")
        s = bytes(buf[o:o + ln]).decode()
        o += ln
        return s

    magic, vchar, version, n, n_prims, n_pgroups = take("ici" + "iii")
    assert magic == int.from_bytes(b"Bgeo", "big") and vchar == b"V" and version == 5
    n_primgroups, n_pattr, n_vattr, n_primattr, n_attr = take("iiiii")
    assert (n_prims, n_pgroups, n_primgroups, n_vattr, n_primattr, n_attr) == (1, 0, 0, 0, 1, 0)
    attrs, width = [], 4
    for _ in range(n_pattr):
        name = hstr()
        size, htype = take("Hi")
        assert htype in (HOUDINI_FLOAT, HOUDINI_INT, HOUDINI_VECTOR)
        defaults = take("i" * size)
        assert all(d == 0 for d in defaults)
        attrs.append((name, size, htype))
        width += size
    rows = np.frombuffer(buf, dtype=">u4", count=n * width, offset=o).reshape(n, width)
    o += n * width * 4
    pos4 = rows[:, :4].astype("<u4").view(np.float32)
    assert n == 0 or np.all(pos4[:, 3] == 1.0)
    data, c = {}, 4
    for name, size, htype in attrs:
        w = np.ascontiguousarray(rows[:, c:c + size]).astype("<u4")
        data[name] = w.view(np.int32) if htype == HOUDINI_INT else w.view(np.float32)
        c += size
    assert hstr() == "generator"
    assert take("hii") == (1, 4, 1)
    assert hstr() == "papi"
    assert take("ii") == (0x8000, n)
    if n > (1 << 16):
        idx = np.frombuffer(buf, dtype=">i4", count=n, offset=o)
        o += 4 * n
    else:
        idx = np.frombuffer(buf, dtype=">u2", count=n, offset=o)
        o += 2 * n
    assert np.array_equal(idx, np.arange(n))
    assert take("i") == (0,)
    assert bytes(buf[o:o + 2]) == b"\x00\xff" and o + 2 == len(buf)
    return dict(n=n, attrs=attrs, data=data, position=pos4[:, :3].copy())
