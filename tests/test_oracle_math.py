"""Oracle self-checks for the pieces the reference does not pin (svd/polar/materials/friction).

The reference holds no golden vectors here ("parity unpinned"): these tests pin the oracle
by invariants and against numpy (an independent implementation)."""
import numpy as np
import pytest

from oracle import np_mpm


def _rand_F(rng, scale=0.3, n=1):
    return np.eye(3) + rng.normal(0, scale, (n, 3, 3))


def test_svd3_reconstruction_and_convention(orc):
    rng = np.random.default_rng(0)
    for F in _rand_F(rng, 0.5, 500):
        U, S, V = orc.svd3(F)
        assert np.allclose(U @ np.diag(S) @ V.T, F, atol=5e-6)
        assert np.allclose(U.T @ U, np.eye(3), atol=2e-6) and np.allclose(V.T @ V, np.eye(3), atol=2e-6)
        assert np.linalg.det(U) > 0.99 and np.linalg.det(V) > 0.99           # rotations
        assert abs(S[0]) >= abs(S[1]) - 1e-6 and abs(S[1]) >= abs(S[2]) - 1e-6  # |sigma| descending
        assert S[0] >= 0 and S[1] >= 0
        assert np.sign(S[2]) == np.sign(np.linalg.det(F)) or abs(S[2]) < 1e-6  # sign on the last one
        s_np = np.linalg.svd(F.astype(np.float32).astype(np.float64), compute_uv=False)
        assert np.allclose(np.abs(S), s_np, atol=3e-6)


def test_svd3_degenerate(orc):
    for F in (np.eye(3), 2 * np.eye(3), np.diag([1, 1, 0.5]), np.diag([3.0, 1e-3, 1e-3]), np.zeros((3, 3)),
              np.diag([1.0, 1.0, -1.0])):
        U, S, V = orc.svd3(F)
        assert np.all(np.isfinite(U)) and np.all(np.isfinite(V)) and np.all(np.isfinite(S))
        assert np.allclose(U @ np.diag(S) @ V.T, F, atol=1e-6)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-6)


def test_polar3(orc):
    rng = np.random.default_rng(1)
    for F in _rand_F(rng, 0.3, 300):
        R, S = orc.polar3(F)
        assert np.allclose(R @ S, F, atol=5e-6)
        assert np.allclose(R.T @ R, np.eye(3), atol=2e-6)
        assert np.allclose(S, S.T, atol=2e-6)
        Un, sn, Vtn = np.linalg.svd(F)
        if np.linalg.det(F) > 0:
            assert np.allclose(R, Un @ Vtn, atol=5e-6)


def test_svd2_polar2(orc):
    rng = np.random.default_rng(2)
    for _ in range(300):
        F = np.eye(2) + rng.normal(0, 0.4, (2, 2))
        R, S = orc.polar2(F)
        assert np.allclose(R @ S, F, atol=2e-6) and np.allclose(R.T @ R, np.eye(2), atol=1e-6)
        assert np.allclose(S, S.T, atol=2e-6) and np.linalg.det(R) > 0
        U, s, V = orc.svd2(F)
        assert np.allclose(U @ np.diag(s) @ V.T, F, atol=3e-6)
        assert np.allclose(np.sort(np.abs(s))[::-1], np.linalg.svd(F, compute_uv=False), atol=3e-6)


MATS = ["jelly", "snow", "linear", "water", "sand", "von_mises", "elastic", "visco"]


@pytest.mark.parametrize("mat", MATS)
def test_calculate_force_vs_numpy(orc, mat):
    """C++ oracle calculate_force == independent numpy restatement (src/particles.cpp per material)."""
    rng = np.random.default_rng(3)
    gp, t = orc.group_params(mat, 400 * 1e-6, 1e-6)
    for F in _rand_F(rng, 0.05, 100):
        aux = {"snow": 1.02, "water": 0.97}.get(mat, 0.0)
        a = orc.calculate_force(t, gp, F, aux)
        F32 = F.astype(np.float32).astype(np.float64)
        b = np_mpm.kirchhoff_like_force(np.array([t]), gp[None].astype(np.float64), F32[None], np.array([aux]))[0]
        assert np.allclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(b).max())), (mat, a, b)


@pytest.mark.parametrize("mat", MATS)
def test_plasticity_vs_numpy(orc, mat):
    rng = np.random.default_rng(4)
    gp, t = orc.group_params(mat, 400 * 1e-6, 1e-6)
    for _ in range(100):
        F = _rand_F(rng, 0.04)[0]
        cdg = np.eye(3) + rng.normal(0, 0.02, (3, 3))
        aux = {"snow": 1.0, "water": 1.0, "sand": 0.005, "visco": 1000.0}.get(mat, 0.0)
        Fa, auxa = orc.plasticity(t, gp, cdg, F, aux)
        F32 = F.astype(np.float32).astype(np.float64)
        c32 = cdg.astype(np.float32).astype(np.float64)
        Fb, auxb = np_mpm.plasticity(np.array([t]), gp[None].astype(np.float64), c32[None], F32[None], np.array([aux]))
        if mat == "water":
            Fb = F32[None]
        assert np.allclose(Fa, Fb[0], atol=1e-5), (mat, Fa, Fb[0])
        assert abs(auxa - auxb[0]) < 1e-5 * max(1, abs(auxb[0]))


def test_fixed_corotated_zero_at_rotation(orc):
    """P(R) = 0 for a pure rotation; P(I)=0 for every elastic model."""
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    for mat in ("jelly", "snow", "sand", "elastic", "von_mises"):
        gp, t = orc.group_params(mat, 1.0, 1.0)
        aux = 1.0 if mat == "snow" else 0.0
        assert np.abs(orc.calculate_force(t, gp, R, aux)).max() < 0.2  # mu ~1e5 * fp32 eps
        assert np.abs(orc.calculate_force(t, gp, np.eye(3), aux)).max() == 0


def test_snow_clamps_singular_values(orc):
    gp, t = orc.group_params("snow", 1.0, 1.0)
    F, Jp = orc.plasticity(t, gp, np.diag([1.2, 1.0, 0.9]), np.eye(3), 1.0)
    s = np.linalg.svd(F, compute_uv=False)
    assert np.allclose(sorted(s), [0.975, 1.0, 1.0075], atol=1e-6)
    assert np.isclose(Jp, 1.2 * 0.9 / (1.0075 * 0.975), rtol=1e-5)


def test_sand_cases(orc):
    gp, t = orc.group_params("sand", 1.0, 1.0)
    # expansion (tr >= 0): projects to the tip sigma = exp(cohesion) = 1, logJp accumulates
    F, logJp = orc.plasticity(t, gp, np.diag([1.1, 1.1, 1.1]), np.eye(3), 0.0)
    assert np.allclose(F, np.eye(3), atol=1e-6) and np.isclose(logJp, 3 * np.log(1.1), rtol=1e-5)
    # pure compression inside the cone: unchanged
    F, logJp = orc.plasticity(t, gp, np.diag([0.95, 0.95, 0.95]), np.eye(3), 0.0)
    assert np.allclose(F, 0.95 * np.eye(3), atol=1e-6) and logJp == 0
    # shear under slight compression: return-mapped, volume preserved
    F, logJp = orc.plasticity(t, gp, np.diag([1.05, 0.9, 1.0]), np.eye(3), 0.0)
    s = np.linalg.svd(F, compute_uv=False)
    assert np.isclose(np.log(s).sum(), np.log(1.05 * 0.9), atol=1e-5)
    assert s.max() < 1.05 and s.min() > 0.9


def test_water_pressure_and_j(orc):
    gp, t = orc.group_params("water", 1.0, 2.0)
    f = orc.calculate_force(t, gp, np.eye(3), 0.9)
    p = 10000.0 * (0.9 ** -7 - 1)
    assert np.allclose(f, 2.0 * 0.9 * p * np.eye(3), rtol=1e-5)   # -vol*j*(-p I)
    _, j = orc.plasticity(t, gp, np.diag([1.01, 0.98, 1.0]), np.eye(3), 0.9)
    assert np.isclose(j, 0.9 * (1.01 + 0.98 + 1.0 - 2), rtol=1e-6)
    _, j = orc.plasticity(t, gp, np.diag([0.1, 0.1, 0.1]), np.eye(3), 0.5)
    assert j == np.float32(0.1)


def test_friction_project(orc):
    """src/mpm_fwd.h:25-57 and README.md:326-330 friction codes."""
    n = np.array([0, 1, 0.0]); vb = np.zeros(3)
    v = np.array([1.0, -2.0, 0.5])
    assert np.allclose(orc.friction_project(v, vb, n, -1), vb)                       # sticky
    assert np.allclose(orc.friction_project(v, vb, n, -2), [1.0, 0, 0.5])           # slip, frictionless
    out = orc.friction_project(v, vb, n, 0.0)                                         # separate, mu=0
    assert np.allclose(out, [1.0, 0, 0.5])
    up = orc.friction_project(np.array([1.0, 2.0, 0.5]), vb, n, 0.4)                  # separating: untouched
    assert np.allclose(up, [1.0, 2.0, 0.5])
    out = orc.friction_project(v, vb, n, 0.4)                                         # Coulomb
    tn = np.hypot(1.0, 0.5); sc = max(tn - 2 * 0.4, 0) / tn
    assert np.allclose(out, [sc, 0, 0.5 * sc], atol=1e-6)
    out = orc.friction_project(np.array([1.0, 2.0, 0.5]), vb, n, -2.3)                # slip never separates
    assert np.isclose(out[1], 0)
    vb2 = np.array([0.3, 0.1, 0.0])
    assert np.allclose(orc.friction_project(v, vb2, n, -1), vb2)


@pytest.mark.parametrize("tau,kappa", [(10.0, 0.0), (3000.0, 0.4), (1e6, 0.4)])
def test_visco_flow_branch_vs_numpy(orc, tau, kappa):
    """ViscoParticle::plasticity (src/particles.cpp:102-134) with the flow active (|P| > tau, 0 < gamma <= 1), with
    hardening of tau (kappa) and with the flow off (tau above |P|): C++ oracle == numpy restatement"""
    rng = np.random.default_rng(14)
    gp, t = orc.group_params("visco", 400 * 1e-6, 1e-6, kappa=kappa)
    flowed = 0
    for _ in range(100):
        F = _rand_F(rng, 0.04)[0]
        cdg = np.eye(3) + rng.normal(0, 0.02, (3, 3))
        Fa, auxa = orc.plasticity(t, gp, cdg, F, tau)
        F32 = F.astype(np.float32).astype(np.float64)
        c32 = cdg.astype(np.float32).astype(np.float64)
        Fb, auxb = np_mpm.plasticity(np.array([t]), gp[None].astype(np.float64), c32[None], F32[None], np.array([tau]))
        assert np.allclose(Fa, Fb[0], atol=2e-5), (Fa, Fb[0])
        assert abs(auxa - auxb[0]) <= 1e-4 * max(1.0, abs(auxb[0]))
        still, _ = np_mpm.plasticity(np.array([t]), gp[None].astype(np.float64), c32[None], F32[None], np.array([1e9]))
        flowed += not np.allclose(Fb[0], still[0], atol=1e-6)  # differs from the same update with the flow switched off
    assert (flowed > 50) == (tau < 1e5), flowed


def _energy_density(mat, mu, lam, F):
    """strain energy densities the models derive from: fixed-corotated (jelly / snow / visco, src/particles.cpp:400-407),
    small-strain linear (:323-327), Hencky / quadratic-log (elastic :785-796; sand and von_mises use the same elastic law)"""
    U, s, Vt = np.linalg.svd(F)
    if mat in ("jelly", "snow", "visco"):
        R = U @ Vt
        J = np.linalg.det(F)
        return mu * ((F - R) ** 2).sum() + 0.5 * lam * (J - 1) ** 2
    if mat == "linear":
        e = 0.5 * (F + F.T) - np.eye(3)
        return mu * (e ** 2).sum() + 0.5 * lam * np.trace(e) ** 2
    ls = np.log(s)
    return mu * (ls ** 2).sum() + 0.5 * lam * ls.sum() ** 2


@pytest.mark.parametrize("mat", ["jelly", "snow", "visco", "linear", "elastic", "sand", "von_mises"])
def test_stress_is_the_derivative_of_the_strain_energy(orc, mat):
    """hyperelasticity ties the stress to an energy: P = d Psi / d F.  A central difference of Psi (numpy, float64)
    against the oracle's calculate_force() = -vol P F^T checks every elastic law without a second restatement of the
    stress formula."""
    rng = np.random.default_rng(31)
    vol = 1e-6
    gp, t = orc.group_params(mat, 400 * vol, vol)
    mu, lam = float(gp[2]), float(gp[3])
    aux = {"snow": 1.0, "visco": 1000.0}.get(mat, 0.0)  # Jp = 1: no hardening factor
    h = 1e-5
    for F in _rand_F(rng, 0.03, 40):
        F = F.astype(np.float32).astype(np.float64)
        P = np.zeros((3, 3))
        for i in range(3):
            for j in range(3):
                d = np.zeros((3, 3))
                d[i, j] = h
                P[i, j] = (_energy_density(mat, mu, lam, F + d) - _energy_density(mat, mu, lam, F - d)) / (2 * h)
        want = -vol * P @ F.T
        got = orc.calculate_force(t, gp, F, aux)
        assert np.allclose(got, want, rtol=2e-3, atol=2e-4 * np.abs(want).max()), (mat, got, want)


def test_sand_return_map_lands_on_or_inside_the_drucker_prager_cone(orc):
    """after plasticity() the Hencky strain of a sand particle satisfies the yield condition of the projection
    (src/particles.cpp:599-626): y = |dev eps| + (3 lambda + 2 mu) / (2 mu) * tr(eps) * alpha <= 0 under compression
    (on the cone, y = 0, whenever the trial state was outside), and the state sits at the tip (eps = 0) under
    expansion; an idempotent map: projecting again changes nothing"""
    rng = np.random.default_rng(41)
    gp, t = orc.group_params("sand", 1.0, 1.0)
    mu, lam, alpha = float(gp[2]), float(gp[3]), float(gp[4])
    on_cone = tip = inside = 0
    for _ in range(300):
        F0 = np.eye(3) + rng.normal(0, 0.03, (3, 3))
        cdg = np.eye(3) + rng.normal(0, 0.05, (3, 3))
        F, logJp = orc.plasticity(t, gp, cdg, F0, 0.0)
        eps = np.log(np.linalg.svd(F, compute_uv=False))
        tr = eps.sum()
        dev = eps - tr / 3
        y = np.linalg.norm(dev) + (3 * lam + 2 * mu) / (2 * mu) * tr * alpha
        trial = np.log(np.linalg.svd(cdg.astype(np.float32).astype(np.float64) @ F0.astype(np.float32).astype(np.float64),
                                     compute_uv=False))
        if np.abs(eps).max() < 1e-5:
            tip += 1  # expansion: projected to sigma = 1
            assert trial.sum() >= -1e-5
        else:
            assert y <= 2e-5, (y, eps)
            y_trial = np.linalg.norm(trial - trial.sum() / 3) + (3 * lam + 2 * mu) / (2 * mu) * trial.sum() * alpha
            if y_trial > 1e-4:
                on_cone += 1
                assert abs(y) <= 2e-5  # returned exactly onto the cone
                assert abs(tr - trial.sum()) <= 2e-5  # along the deviator: the volumetric strain is kept
            else:
                inside += 1
                assert np.allclose(np.sort(eps), np.sort(trial), atol=2e-5)  # elastic step: untouched
        F2, _ = orc.plasticity(t, gp, np.eye(3), F, logJp)
        assert np.allclose(F2, F, atol=2e-5)
    assert on_cone > 30 and tip > 30 and inside > 5, (on_cone, tip, inside)


def test_von_mises_return_map_bounds_the_deviator(orc):
    """src/particles.cpp:713-732: the squared norm of the deviatoric Hencky strain is capped at yield_stress / (2 mu)...
    as the reference writes it (dg = |dev|^2 - yield / (2 mu) > 0  =>  eps -= (dg / |dev|^2) dev), i.e. after the map
    |dev| = |dev_trial| * yield / (2 mu |dev_trial|^2); volume untouched; elastic below the threshold"""
    rng = np.random.default_rng(42)
    gp, t = orc.group_params("von_mises", 1.0, 1.0, yield_stress=20.0)
    mu, ys = float(gp[2]), float(gp[4])
    plastic = 0
    for _ in range(200):
        F0 = np.eye(3) + rng.normal(0, 0.02, (3, 3))
        cdg = np.eye(3) + rng.normal(0, 0.06, (3, 3))
        F, _ = orc.plasticity(t, gp, cdg, F0, 0.0)
        trial = np.log(np.linalg.svd(cdg.astype(np.float32).astype(np.float64) @ F0.astype(np.float32).astype(np.float64),
                                     compute_uv=False))
        eps = np.log(np.linalg.svd(F, compute_uv=False))
        d_t = trial - trial.sum() / 3
        n2 = (d_t ** 2).sum()
        assert abs(eps.sum() - trial.sum()) <= 2e-5
        if n2 - ys / (2 * mu) > 1e-6:
            plastic += 1
            want = trial - ((n2 - ys / (2 * mu)) / n2) * d_t
            assert np.allclose(np.sort(eps), np.sort(want), atol=3e-5)
        else:
            assert np.allclose(np.sort(eps), np.sort(trial), atol=3e-5)
    assert 20 < plastic < 200
