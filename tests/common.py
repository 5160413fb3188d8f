"""Shared seeded scene builders for oracle and GPU parity tests."""
import numpy as np

from oracle import oracle as orc


# per-material Config keys used by the seeded scenes (empty = the reference's defaults, src/particles.cpp initialize())
MAT_KW = {}


def lattice_cube(res, lo_cell, hi_cell, dx, jitter=0.0, seed=0):
    """8 particles per cell at the +-0.25*dx lattice of the reference's benchmark generator
    (src/mpm.cpp:164-180), optionally jittered."""
    rng = np.random.default_rng(seed)
    ii, jj, kk = np.meshgrid(np.arange(lo_cell, hi_cell), np.arange(lo_cell, hi_cell),
                             np.arange(lo_cell, hi_cell), indexing="ij")
    cells = np.stack([ii, jj, kk], -1).reshape(-1, 3).astype(np.float64)
    pts = []
    for i in range(8):
        sign = np.array([1.0, 1.0, 1.0])
        if i % 2 == 0:
            sign[0] = -1
        if i // 2 % 2 == 0:
            sign[1] = -1
        if i // 4 % 2 == 0:
            sign[2] = -1
        # Region index get_pos() is the cell centre (ipos + 0.5)
        pts.append((cells + 0.5) * dx + 0.25 * dx * sign)
    x = np.stack(pts, 1).reshape(-1, 3)
    if jitter:
        x = x + rng.uniform(-jitter, jitter, x.shape) * dx
    return x.astype(np.float32)


def make_state(x, type_name, dx, density=400.0, ppc=8, seed=1, vel_scale=1.0, perturb_F=0.05, **mat_kw):
    """random-but-plausible v, B, F around a rotating/shearing field so every term is exercised."""
    rng = np.random.default_rng(seed)
    n = len(x)
    vol = dx ** 3 / ppc
    gp, t = orc.group_params(type_name, vol * density, vol, **mat_kw)
    c = x.mean(0)
    r = (x - c).astype(np.float64)
    omega = np.array([0.3, 1.0, -0.5]) * vel_scale
    v = np.cross(omega, r) * 4.0 + rng.normal(0, 0.05 * vel_scale, (n, 3))
    B = rng.normal(0, 0.02 * vel_scale, (n, 9))
    F = np.tile(np.eye(3).reshape(1, 9), (n, 1)) + rng.normal(0, perturb_F, (n, 9))
    aux = np.full(n, orc.initial_aux(type_name, **mat_kw), np.float32)
    if type_name == "snow":
        aux = (1.0 + rng.normal(0, 0.02, n)).astype(np.float32)
    if type_name == "water":
        aux = (1.0 + rng.normal(0, 0.01, n)).astype(np.float32)
    if type_name == "sand":
        aux = np.abs(rng.normal(0, 0.01, n)).astype(np.float32)
    return orc.State(x, v, B, F, aux, np.zeros(n, np.int32), gp[None, :], np.array([t], np.int32))


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
