"""The block scheduler of the asynchronous stepper (taichi_mpm_amd/csrc/async_sched.h: block geometry, the reference's block order,
neighbour lists, the level walk and its action tables — host code shared by AsyncMPM<3> and AsyncMPM<2>) compiled for the host by
g++: against the geometry of the REFERENCE's own scheduler (src/async/async_mpm.{h,cpp} compiled in place into
oracle/_ref/libmpm_ref.so: SparseMask::LinearToCoord of every scheduler offset, cached_neighbours, the left_boundary list) in
both dimensions, and the level walk on a synthetic table.  No GPU needed; tests/test_gpu_async.py and tests/test_gpu_async2d.py
run the steppers themselves against the reference's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "async_sched_host.cpp")
HDR = os.path.join(ROOT, "taichi_mpm_amd", "csrc", "async_sched.h")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libasync_sched_host.so")
P_I, P_L = C.POINTER(C.c_int), C.POINTER(C.c_long)


def host_lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(OUT):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-shared", "-fPIC", SRC, "-o", OUT])
    L = C.CDLL(OUT)
    L.sched_geometry.restype = C.c_long
    L.sched_geometry.argtypes = [C.c_int, P_I, C.c_int, C.c_long, P_I, P_I, P_I, P_I, P_I]
    L.sched_walk.restype = C.c_long
    L.sched_walk.argtypes = [C.c_int, P_I, C.c_float, C.c_int, C.c_float, C.POINTER(C.c_uint32), C.c_int, P_L, P_L, C.c_char_p, C.c_int]
    return L


def geometry(L, dim, res, left_boundary):
    r = (C.c_int * 3)(*(list(res) + [1] * (3 - dim)))
    nb = (C.c_int * 3)()
    n = L.sched_geometry(dim, r, int(left_boundary), 0, None, None, None, None, nb)
    coord, rank, neigh, bnd = np.zeros((n, 3), np.int32), np.zeros(n, np.int32), np.zeros((n, 26), np.int32), np.zeros(n, np.int32)
    assert L.sched_geometry(dim, r, int(left_boundary), n, *(a.ctypes.data_as(P_I) for a in (coord, rank, neigh, bnd)), nb) == n
    return coord, rank, neigh, bnd, tuple(nb)


@pytest.mark.parametrize("dim,res", [(2, (128, 128)), (2, (96, 160)), (3, (32, 32, 32)), (3, (48, 24, 40))])
def test_block_geometry_order_and_neighbours_are_the_reference_schedulers(dim, res):
    from oracle import refmpm as ref
    if not ref.available():
        pytest.skip("oracle/_ref/libmpm_ref.so not built")
    L = host_lib()
    coord, rank, neigh, bnd, nb = geometry(L, dim, res, True)
    assert nb[:dim] == tuple((r >> s) + 1 for r, s in zip(res, (3, 4) if dim == 2 else (2, 2, 3)))
    r = ref.AsyncSim(res, 1.0 / res[0], dim=dim, left_boundary=True)
    rc, rn, rb = r.geometry()
    r.close()
    of = {tuple(c): i for i, c in enumerate(rc.tolist())}  # corner node -> the reference's scheduler offset
    mine = {tuple(c): b for b, c in enumerate(coord.tolist())}
    off = np.array([of[tuple(c)] for c in coord.tolist()])  # every block of the table exists in the reference's scheduler
    # (i) the order the pools are walked in: ranks order the table's blocks as the reference's offsets do
    assert np.array_equal(np.argsort(rank), np.argsort(off))
    # (ii) neighbours: the reference's cached_neighbours, restricted to the blocks of the table (the reference's scheduler covers
    #      the power-of-two SPGrid domain; blocks beyond `res` never hold a particle)
    for b in range(len(coord)):
        want = sorted(mine[tuple(rc[q])] for q in rn[off[b]] if q >= 0 and tuple(rc[q]) in mine)
        got = sorted(int(q) for q in neigh[b] if q >= 0)
        assert got == want, (b, coord[b])
    # (iii) left_boundary blocks (src/async/async_mpm.cpp:43-53)
    assert np.array_equal(bnd, rb[off])
    assert bnd.sum() > 0


def test_level_walk_keeps_pools_and_backups_consistent_on_a_three_level_table():
    """a 2D table with a stiff island (64 units) inside a soft region (512) inside empty space: 40 rounds of the walk — every
    advance finds the pools / backups it gathers at the time it expects (the reference's "particle_pool broken" / "backup_pool
    broken" assertions, src/async/async_mpm.cpp:266-300), and the levels advance at their rates"""
    L = host_lib()
    res = (128, 128)
    coord, rank, neigh, bnd, nb = geometry(L, 2, res, False)
    n = len(coord)
    tab = np.zeros((n, 3), np.uint32)
    unit = np.float32(2e-6)
    f2u = lambda v: np.array([v], np.float32).view(np.uint32)[0]  # noqa: E731
    tab[:, 0], tab[:, 1] = f2u(0.1), f2u(1e-16)
    bx, by = coord[:, 0] >> 3, coord[:, 1] >> 4
    soft = (bx >= 3) & (bx <= 9) & (by >= 2) & (by <= 5)
    stiff = (bx >= 5) & (bx <= 6) & (by == 3)
    tab[soft, 0], tab[soft, 2] = f2u(600 * unit), 100      # strength limit 600 units -> level 512
    tab[stiff, 0], tab[stiff, 2] = f2u(70 * unit), 100     # 70 units -> level 64
    out = np.zeros((n, 3), np.int64)
    adv = np.zeros(64, np.int64)
    err = C.create_string_buffer(256)
    r3 = (C.c_int * 3)(res[0], res[1], 1)
    t = L.sched_walk(2, r3, unit, 1024, 1.0 / 128, tab.ctypes.data_as(C.POINTER(C.c_uint32)), 40, out.ctypes.data_as(P_L),
                     adv.ctypes.data_as(P_L), err, 256)
    assert t > 0, err.value.decode()
    assert set(np.unique(out[soft & ~stiff, 0])) == {512} and set(np.unique(out[stiff, 0])) == {64}
    assert t == 40 * 64  # one round = the smallest level in use
    assert adv[6] == 40 and adv[9] == 40 * 64 // 512
    assert np.all(out[stiff, 1] == t)  # the stiff pools are at the current time
    assert np.all(out[soft & ~stiff, 1] >= t) and np.all(out[soft & ~stiff, 1] % 512 == 0)
