"""Per-material parameter rows for mpmhip_add_group — the host mirror of each
`XParticle::initialize(const Config&)` of the reference (src/particles.cpp; line ranges inline).

Row layout (float[16]) is documented in include/mpmhip.h."""
import math

import numpy as np

NPARAM = 16
MATERIAL_IDS = {"visco": 1, "snow": 2, "linear": 3, "jelly": 4, "water": 5, "sand": 6, "von_mises": 7,
                "elastic": 8}


def _lame(E, nu):
    return E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))


def group_params(type_name, mass, vol, **kw):
    """-> (float32[16], material id).  Unknown names raise, as `create_instance_placement` does for an
    unregistered alias (src/particle_allocator.h:68-74)."""
    if type_name not in MATERIAL_IDS:
        raise KeyError("unknown particle type '%s' (registered: %s)" % (type_name, ", ".join(sorted(MATERIAL_IDS))))
    if "compressibility" in kw:  # src/particles.h:117-119
        raise ValueError("'compressibility' is deprecated. Use 'initial_dg' instead")
    t = MATERIAL_IDS[type_name]
    p = np.zeros(NPARAM, np.float32)
    p[0], p[1] = mass, vol
    if type_name == "snow":  # src/particles.cpp:192-205
        mu, lam = _lame(kw.get("youngs_modulus", 1.4e5), kw.get("poisson_ratio", 0.2))
        p[2], p[3] = kw.get("mu_0", mu), kw.get("lambda_0", lam)
        p[4] = kw.get("hardening", 10.0)
        p[5], p[6] = kw.get("theta_c", 2.5e-2), kw.get("theta_s", 7.5e-3)
        p[7], p[8] = kw.get("min_Jp", 0.6), kw.get("max_Jp", 20.0)
    elif type_name in ("linear", "jelly"):  # :315-321, :383-389
        p[2], p[3] = _lame(kw.get("E", 1e5), kw.get("nu", 0.3))
    elif type_name == "water":  # :448-461 (k defaults to 1e4 in code; README says 1e5 — SURVEY quirk 6)
        p[2], p[3] = kw.get("k", 10000.0), kw.get("gamma", 7.0)
    elif type_name == "sand":  # :570-597 (pi ~ 3.141592653, degrees)
        p[2], p[3] = kw.get("mu_0", 136038.0), kw.get("lambda_0", 204057.0)
        sin_phi = math.sin(np.float32(kw.get("friction_angle", 30.0)) / np.float32(180.0) * np.float32(3.141592653))
        p[4] = math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi)
        p[5], p[6] = kw.get("cohesion", 0.0), kw.get("beta", 1.0)
    elif type_name == "von_mises":  # :691-699
        p[2], p[3] = _lame(kw.get("youngs_modulus", 5e3), kw.get("poisson_ratio", 0.4))
        p[4] = kw.get("yield_stress", 1.0)
    elif type_name == "elastic":  # :777-783
        p[2], p[3] = _lame(kw.get("E", 5e3), kw.get("nu", 0.4))
        p[4] = kw.get("E", 5e3)  # only reported back by get_debug_info() (verbose .bgeo), :838-840
    elif type_name == "visco":  # :57-70
        p[2], p[3] = _lame(kw.get("youngs_modulus", 4e4), kw.get("poisson_ratio", 0.4))
        # src/particles.cpp:57-70: the particle reads "base_delta_t" from ITS config (default 1e-4)
        p[4], p[5], p[6] = kw.get("nu", 10000.0), kw.get("kappa", 0.0), kw.get("base_delta_t", 1e-4)
    return p, t


def initial_aux(type_name, **kw):
    """initial material state: Jp (snow, :204), j (water, :460), logJp (sand, :595), visco_tau (:65)."""
    if type_name == "snow":
        return float(kw.get("Jp", 1.0))
    if type_name == "water":
        return 1.0
    if type_name == "visco":
        return float(kw.get("tau", 1000.0))
    return 0.0
