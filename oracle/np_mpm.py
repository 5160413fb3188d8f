"""Independent numpy (float64) restatement of one MLS-MPM substep — TEST INFRASTRUCTURE ONLY.

Purpose: a *second opinion* for oracle/mpm_oracle.cpp, written from the maths rather than from the
reference's instruction sequence (vectorised, float64, `numpy.linalg.svd` for every factorisation),
so that a transcription slip in the C++ oracle cannot hide.  It follows the same reference semantics
(file:line relative to /root/reference):

  P2G   src/transfer.cpp:193-278 (generic `rasterize`, same-colour branch, MLSMPM)
  grid  src/mpm.cpp:277-294, 296-372 ; friction_project src/mpm_fwd.h:25-57
  G2P   src/transfer.cpp:585-687 (generic `resample`, without the position clamp of :668-670,
        which the optimised path :837-954 does not have)
  materials  src/particles.cpp (line ranges in oracle_core.h)

Status: "parity unpinned" (see oracle/mpm_oracle.h).
"""
import numpy as np

VISCO, SNOW, LINEAR, JELLY, WATER, SAND, VON_MISES, ELASTIC = 1, 2, 3, 4, 5, 6, 7, 8


def _weights(fx):
    """quadratic B-spline, fx in [0.5,1.5): src/kernel.h:126-130 in closed form."""
    return np.stack([0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1.0) ** 2, 0.5 * (fx - 0.5) ** 2], axis=1)  # (n,3,dim)


def _svd_rot(F):
    """batched SVD with U,V rotations and the sign on the smallest singular value."""
    U, s, Vt = np.linalg.svd(F)
    V = np.swapaxes(Vt, -1, -2)
    du, dv = np.linalg.det(U), np.linalg.det(V)
    s = s.copy()
    U = U.copy(); V = V.copy()
    U[du < 0, :, 2] *= -1; s[du < 0, 2] *= -1
    V[dv < 0, :, 2] *= -1; s[dv < 0, 2] *= -1
    return U, s, V


def _diag(s):
    out = np.zeros(s.shape[:-1] + (3, 3))
    for k in range(3):
        out[..., k, k] = s[..., k]
    return out


def kirchhoff_like_force(types, gp, F, aux):
    """returns -vol * P(F) * F^T per particle (calculate_force)."""
    n = len(F)
    out = np.zeros((n, 3, 3))
    I = np.eye(3)
    vol = gp[:, 1]
    for t in np.unique(types):
        m = types == t
        Fm, g = F[m], gp[m]
        if t in (JELLY, SNOW, VISCO):
            mu, lam = g[:, 2].copy(), g[:, 3].copy()
            if t == SNOW:
                e = np.exp(g[:, 4] * (1.0 - aux[m]))
                mu, lam = mu * e, lam * e
            U, s, V = _svd_rot(Fm)
            R = U @ np.swapaxes(V, -1, -2)
            J = np.linalg.det(Fm)
            PFt = 2 * mu[:, None, None] * (Fm - R) @ np.swapaxes(Fm, -1, -2) + (lam * (J - 1) * J)[:, None, None] * I
        elif t == LINEAR:
            mu, lam = g[:, 2], g[:, 3]
            P = mu[:, None, None] * (Fm + np.swapaxes(Fm, -1, -2) - 2 * I) + (lam * (np.trace(Fm, axis1=1, axis2=2) - 3))[:, None, None] * I
            PFt = P @ np.swapaxes(Fm, -1, -2)
        elif t == WATER:
            j = aux[m]
            p = g[:, 2] * (j ** (-g[:, 3]) - 1.0)
            # -vol * j * (-p I)  ==  -vol * (P F^T) with P F^T := -j p I
            PFt = (-j * p)[:, None, None] * I
        elif t in (SAND, VON_MISES, ELASTIC):
            mu, lam = g[:, 2], g[:, 3]
            U, s, V = _svd_rot(Fm)
            with np.errstate(invalid="ignore", divide="ignore"):
                ls = np.log(s)
            center = (2 * mu[:, None] * ls + lam[:, None] * ls.sum(1, keepdims=True)) / s
            P = U @ _diag(center) @ np.swapaxes(V, -1, -2)
            PFt = P @ np.swapaxes(Fm, -1, -2)
        else:
            raise NotImplementedError(t)
        out[m] = -vol[m][:, None, None] * PFt
    return out


def _approx_exp(dt, M):
    """ViscoParticle::approximate_exponent (src/particles.cpp:88-100): 1 + s + s^2/2 with s = M dt, halving dt and
    squaring while the determinant is not positive"""
    s = M * dt
    r = (s * 0.5 + np.eye(3)) @ s + np.eye(3)
    if np.linalg.det(r) > 0:
        return r
    h = _approx_exp(dt / 2, M)
    return h @ h


def _visco_plasticity(g, cdg, F_old, tau):
    """ViscoParticle::plasticity (src/particles.cpp:102-134), one particle, float64.  g = parameter row
    (mu, lambda at [2], [3]; nu, kappa, dt at [4..6]); F_old = dg_e BEFORE the update (drives the flow)."""
    mu, lam, vnu, kappa, dt = g[2], g[3], g[4], g[5], g[6]
    Fh = _approx_exp(dt, (cdg - np.eye(3)) / dt) @ F_old
    U, s, V = _svd_rot(Fh[None])
    U, s, V = U[0], s[0], V[0]
    Uo, so, Vo = _svd_rot(F_old[None])
    R = Uo[0] @ Vo[0].T
    J = np.linalg.det(F_old)
    P = 2 * mu * (F_old - R) + lam * (J - 1) * J * np.linalg.inv(F_old.T)  # first_piola_kirchhoff, :72-80
    pnorm = np.linalg.norm(P)
    gamma = min(max(dt * vnu * (pnorm - tau) / pnorm, 0.0), 1.0) if pnorm > 1e-5 else 0.0
    det = s.prod()
    scale = 1.0 / det ** (1.0 / 3.0) if abs(det) > 1e-5 else 1.0
    mid = (s * scale) ** gamma
    inv = np.where(np.abs(mid) > 1e-5, 1.0 / mid, 1.0)
    Fn = U @ np.diag(s * inv) @ V.T
    U2, s2, V2 = _svd_rot(Fn[None])
    Fn = U2[0] @ np.diag(np.clip(s2[0], 0.1, 10.0)) @ V2[0].T
    return Fn, tau + kappa * gamma * pnorm


def plasticity(types, gp, cdg, F, aux):
    F_old = F
    F = cdg @ F
    aux = aux.copy()
    for t in np.unique(types):
        m = types == t
        g = gp[m]
        if t in (JELLY, LINEAR, ELASTIC):
            continue
        if t == VISCO:
            idx = np.nonzero(m)[0]
            for k, i in enumerate(idx):
                F[i], aux[i] = _visco_plasticity(g[k], cdg[i], F_old[i], aux[i])
            continue
        if t == WATER:
            j = aux[m] * (np.trace(cdg[m], axis1=1, axis2=2) - 2.0)
            aux[m] = np.maximum(j, 0.1)
            continue
        U, s, V = _svd_rot(F[m])
        if t == SNOW:
            lo, hi = 1.0 - g[:, 5], 1.0 + g[:, 6]
            sc = np.clip(s, lo[:, None], hi[:, None])
            F[m] = U @ _diag(sc) @ np.swapaxes(V, -1, -2)
            Jp = aux[m] * s.prod(1) / sc.prod(1)
            aux[m] = np.clip(Jp, g[:, 7], g[:, 8])
        elif t == SAND:
            mu, lam, alpha, coh, beta = g[:, 2], g[:, 3], g[:, 4], g[:, 5], g[:, 6]
            logJp = aux[m]
            eps = np.log(np.maximum(np.abs(s), 1e-4)) - coh[:, None]
            ssum = eps.sum(1)
            tr = ssum + logJp
            eh = eps - tr[:, None] / 3.0
            ehn = np.linalg.norm(eh, axis=1)
            dg = ehn + (3 * lam + 2 * mu) / (2 * mu) * tr * alpha
            newsig = np.zeros_like(s)
            new_logJp = np.zeros_like(logJp)
            case_a = tr >= 0
            newsig[case_a] = np.exp(coh[case_a])[:, None]
            new_logJp[case_a] = (beta * ssum + logJp)[case_a]
            case_b = (~case_a) & (dg <= 0)
            newsig[case_b] = np.exp(eps[case_b] + coh[case_b][:, None])
            case_c = (~case_a) & (dg > 0)
            with np.errstate(invalid="ignore", divide="ignore"):
                h = eps - (dg / ehn)[:, None] * eh + coh[:, None]
            newsig[case_c] = np.exp(h[case_c])
            F[m] = U @ _diag(newsig) @ np.swapaxes(V, -1, -2)
            aux[m] = new_logJp
        elif t == VON_MISES:
            mu, ys = g[:, 2], g[:, 4]
            e = np.log(s)
            tr = e.sum(1)
            eh = e - tr[:, None] / 3
            ehn2 = (eh ** 2).sum(1)
            dg = ehn2 - ys / (2 * mu)
            H = np.where((dg > 0)[:, None], e - (dg / np.where(ehn2 > 0, ehn2, 1))[:, None] * eh, e)
            Fn = U @ _diag(np.exp(H)) @ np.swapaxes(V, -1, -2)
            Fm = F[m]
            Fm[dg > 0] = Fn[dg > 0]
            F[m] = Fm
        else:
            raise NotImplementedError(t)
    # water keeps F untouched in the reference (dg_e is never updated, src/particles.cpp:469-478)
    return F, aux


def substep(res, dx, dt, gravity, x, v, B, F, aux, gid, gparams, gtype, particle_gravity=True,
            planes=(), friction=-1.0, return_grid=False):
    """float64 substep; B is apic_b in the reference's sign/units."""
    x = np.asarray(x, np.float64); v = np.asarray(v, np.float64)
    B = np.asarray(B, np.float64).reshape(-1, 3, 3); F = np.asarray(F, np.float64).reshape(-1, 3, 3)
    aux = np.asarray(aux, np.float64)
    gp = np.asarray(gparams, np.float64)[gid]
    types = np.asarray(gtype)[gid]
    n = len(x)
    idx = 1.0 / dx
    g = np.asarray(gravity, np.float64)
    if particle_gravity:
        v = v + g * dt
    X = x * idx
    base = np.floor(X - 0.5).astype(np.int64)
    fx = X - base
    w = _weights(fx)  # (n,3,3): [particle, node-offset, axis]
    mass = gp[:, 0]
    water = types == WATER
    F_in = F.copy()
    stress = kirchhoff_like_force(types, gp, F, aux)
    A = stress * (-4.0 * idx * dt) + B * (4.0 * mass)[:, None, None]
    nx, ny, nz = res[0] + 1, res[1] + 1, res[2] + 1
    grid = np.zeros((nx, ny, nz, 4))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                d = fx - np.array([i, j, k], np.float64)
                ww = w[:, i, 0] * w[:, j, 1] * w[:, k, 2]
                mom = mass[:, None] * v + np.einsum("nij,nj->ni", A, d)
                contrib = np.concatenate([ww[:, None] * mom, (ww * mass)[:, None]], axis=1)
                np.add.at(grid, (base[:, 0] + i, base[:, 1] + j, base[:, 2] + k), contrib)
    p2g_grid = grid.copy()
    m = grid[..., 3]
    nz_mask = m > 0
    grid[nz_mask, :3] /= m[nz_mask][:, None]
    if not particle_gravity:
        grid[nz_mask, :3] += g * dt
    if len(planes):
        ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        P = np.stack([ii, jj, kk], -1).astype(np.float64) * dx
        phis = np.stack([(P @ np.asarray(pl[:3], np.float64) + pl[3]) / dx for pl in planes], -1)
        which = phis.argmin(-1)
        phi = phis.min(-1)
        normals = np.asarray([pl[:3] for pl in planes], np.float64)[which]
        act = (m != 0) & (phi >= -3) & (phi <= 0)
        vel = grid[act, :3]
        nn = normals[act]
        if friction == -1:
            newv = np.zeros_like(vel)
        else:
            slip = friction <= -2
            mu = -friction - 2 if slip else friction
            rn = (vel * nn).sum(1)
            rt = vel - rn[:, None] * nn
            tn = np.linalg.norm(rt, axis=1)
            sc = np.maximum(tn + np.minimum(rn, 0) * mu, 0) / np.maximum(1e-30, tn)
            newv = sc[:, None] * rt + np.maximum(0, rn * (0.0 if slip else 1.0))[:, None] * nn
        grid[act, :3] = newv
    vnew = np.zeros((n, 3))
    Bn = np.zeros((n, 3, 3))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                d = fx - np.array([i, j, k], np.float64)
                ww = w[:, i, 0] * w[:, j, 1] * w[:, k, 2]
                gv = grid[base[:, 0] + i, base[:, 1] + j, base[:, 2] + k, :3]
                vnew += ww[:, None] * gv
                Bn += np.einsum("ni,nj->nij", ww[:, None] * gv, d)
    cdg = np.eye(3) + dt * (-4.0 * idx) * Bn
    Fn, auxn = plasticity(types, gp, cdg, F, aux)
    Fn[water] = F_in[water]
    xn = x + dt * vnew
    out = dict(x=xn, v=vnew, B=Bn, F=Fn, aux=auxn)
    if return_grid:
        out["p2g_grid"] = p2g_grid
        out["grid"] = grid
    return out
