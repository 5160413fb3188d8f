"""Known-answer tests the reference holds for the hot path: the B-spline kernels.

These are the ONLY reference tests that pin numbers on the path; they are re-expressed here against
the oracle's restatement of src/kernel.h and src/transfer.cpp:162-191.
"""
import numpy as np


def test_mpm_kernel_partition_of_unity(orc):
    """src/tests.cpp:13-33 `mpm_kernel`: for dim in {2,3}, order in {2,3}: sum w == 1, sum grad w == 0
    (1e-6) at 100 random positions."""
    rng = np.random.default_rng(0)
    for _ in range(100):
        pos3 = rng.random(3).astype(np.float32)
        k = orc.kernel3_dw_w(pos3, 1.0, slow=True)
        assert abs(k[:, 3].sum() - 1) < 1e-6
        assert np.all(np.abs(k[:, :3].sum(0)) < 1e-6)
        k2 = orc.kernel2_dw_w(pos3[:2], 1.0)
        assert abs(k2[:, 2].sum() - 1) < 1e-6
        assert np.all(np.abs(k2[:, :2].sum(0)) < 1e-6)
        for dim in (2, 3):
            kc = orc.kernel_cubic_dw_w(pos3[:dim], 1.0)
            assert abs(kc[:, dim].sum() - 1) < 1e-5
            assert np.all(np.abs(kc[:, :dim].sum(0)) < 1e-5)


def test_mpm_fast_kernel32(orc):
    """src/tests.cpp:35-51 `mpm_fast_kernel32`: MPMFastKernel32::get_dw_w == MPMKernel<3,2>::get_dw_w,
    |diff| < 1e-6, 100 random positions x 27 nodes."""
    rng = np.random.default_rng(1)
    for _ in range(100):
        pos = rng.random(3).astype(np.float32)
        a = orc.kernel3_dw_w(pos, 1.0, slow=True)
        b = orc.kernel3_dw_w(pos, 1.0, slow=False)
        assert np.max(np.linalg.norm(a - b, axis=1)) < 1e-6


def test_grid_pos_offset_table():
    """src/transfer.cpp:353-359 `grid_pos_offset`: node id -> (i/9, i/3%3, i%3)."""
    for i in range(27):
        assert (i // 9, i // 3 % 3, i % 3) == np.unravel_index(i, (3, 3, 3))


def test_mls_kernel(orc):
    """src/transfer.cpp:975-989 `mls_kernel`: MLSMPMFastKernel32.kernels[i][j][k] == MPMFastKernel32::get_w
    (1e-6) for 10 000 random positions in [0.5,1.5)^3."""
    rng = np.random.default_rng(2)
    for _ in range(10000):
        pos = (rng.random(3) + 0.5).astype(np.float32)
        gt = orc.kernel3_dw_w(pos, 1.0)[:, 3]
        fast = orc.mls_kernel3_w(pos)
        assert np.max(np.abs(gt - fast)) < 1e-6


def test_closed_form_weights(orc):
    """src/kernel.h:126-130 in closed form: fx=1 -> (0.125, 0.75, 0.125); general fx in [0.5,1.5)."""
    w = orc.mls_kernel3_w(np.array([1.0, 1.0, 1.0], np.float32)).reshape(3, 3, 3)
    ref1 = np.array([0.125, 0.75, 0.125])
    assert np.allclose(w, np.einsum("i,j,k->ijk", ref1, ref1, ref1), atol=1e-7)
    rng = np.random.default_rng(3)
    for _ in range(200):
        fx = (rng.random(3) + 0.5)
        ws = [np.array([0.5 * (1.5 - f) ** 2, 0.75 - (f - 1) ** 2, 0.5 * (f - 0.5) ** 2]) for f in fx]
        ref = np.einsum("i,j,k->ijk", *ws)
        w = orc.mls_kernel3_w(fx.astype(np.float32)).reshape(3, 3, 3)
        assert np.allclose(w, ref, atol=2e-7)


def test_inv_D_and_stencil_start():
    """MPMKernelBase::inv_D() = 6 - order = 4 (src/kernel.h:68-70); get_stencil_start(x) = int(x - 0.5)
    (src/kernel.h:119-121): truncation, equal to floor for x >= 0.5."""
    assert 6.0 - 2 == 4.0
    for x in (0.5, 0.99, 1.49, 1.5, 7.3, 100.51):
        assert int(np.float32(x) - np.float32(0.5)) == int(np.floor(np.float32(x) - np.float32(0.5)))


def test_block_sort_key_equals_the_reference_spgrid_offset(orc):
    """the key the CPU baseline sorts by == SparseMask::Linear_Offset >> data_bits of the reference's own SPGrid header
    (tests/golden/spgrid_keys.txt, produced by oracle/_ref/spgrid_keys; src/mpm.cpp:785-790)"""
    import ctypes as C
    import os
    L = orc.lib()
    L.orc_spgrid_key.restype = C.c_uint64
    L.orc_spgrid_key.argtypes = [C.c_int] * 3
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spgrid_keys.txt")
    rows = [tuple(int(t) for t in ln.split()) for ln in open(path)]
    assert len(rows) > 500
    for i, j, k, key in rows:
        assert L.orc_spgrid_key(i, j, k) == key, (i, j, k)
