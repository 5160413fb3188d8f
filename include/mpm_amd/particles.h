// include/mpm_amd/particles.h — the MPMParticle plugin surface of the reference (src/particles.h:16-211,
// src/particles.cpp) over the C ABI: a registry of particle types (the aliases of TC_REGISTER_MPM_PARTICLE,
// src/particles.cpp:845-856) that turns a `Config` into the float[16] parameter row of include/mpmhip.h, and a
// host-side `MPMParticle` value whose `calculate_force()` / `plasticity()` run the DEVICE constitutive code on
// one element (mpmhip_debug_force / mpmhip_debug_plasticity) — there is no CPU implementation to fall back to.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>

#include "../mpmhip.h"
#include "kernel.h"

namespace mpm_amd {

// flat string->string configuration, the shape of taichi's `Config` (`config.get<T>(key, default)`)
class Config {
  std::map<std::string, std::string> data_;

 public:
  Config() = default;
  Config &set(const std::string &k, const std::string &v) { data_[k] = v; return *this; }
  Config &set(const std::string &k, const char *v) { data_[k] = v; return *this; }
  Config &set(const std::string &k, double v) { data_[k] = std::to_string(v); return *this; }
  Config &set(const std::string &k, int v) { data_[k] = std::to_string(v); return *this; }
  Config &set(const std::string &k, bool v) { data_[k] = v ? "1" : "0"; return *this; }
  Config &set(const std::string &k, const Vector3 &v) {
    data_[k] = std::to_string(v[0]) + "," + std::to_string(v[1]) + "," + std::to_string(v[2]);
    return *this;
  }
  Config &set(const std::string &k, const Vector3i &v) {
    data_[k] = std::to_string(v[0]) + "," + std::to_string(v[1]) + "," + std::to_string(v[2]);
    return *this;
  }
  bool has_key(const std::string &k) const { return data_.count(k) != 0; }
  const std::string &get_string(const std::string &k) const {
    auto it = data_.find(k);
    if (it == data_.end()) throw std::runtime_error("Config: missing key '" + k + "'");
    return it->second;
  }
  std::string get(const std::string &k, const char *d) const { return has_key(k) ? get_string(k) : std::string(d); }
  double get(const std::string &k, double d) const { return has_key(k) ? std::stod(get_string(k)) : d; }
  float get(const std::string &k, float d) const { return has_key(k) ? std::stof(get_string(k)) : d; }
  int get(const std::string &k, int d) const { return has_key(k) ? std::stoi(get_string(k)) : d; }
  bool get(const std::string &k, bool d) const {
    if (!has_key(k)) return d;
    const std::string &s = get_string(k);
    return !(s == "0" || s == "false" || s == "False" || s.empty());
  }
  template <typename V>
  V get_vec(const std::string &k, const V &d) const {
    if (!has_key(k)) return d;
    V r = d;
    const std::string &s = get_string(k);
    size_t pos = 0;
    for (size_t i = 0; i < r.size(); i++) {
      size_t used = 0;
      r[i] = (typename V::value_type)std::stod(s.substr(pos), &used);
      pos += used;
      while (pos < s.size() && (s[pos] == ',' || s[pos] == ' ' || s[pos] == '(' || s[pos] == ')')) pos++;
      if (pos >= s.size()) {
        for (size_t j = i + 1; j < r.size(); j++) r[j] = r[i];  // scalar broadcast
        break;
      }
    }
    return r;
  }
};

// one registered particle type: factory alias -> (material id, parameter row, initial aux state)
struct ParticleType {
  int32_t material = 0;              // MPMHIP_* id == y of get_debug_info() (src/particles.cpp:157..839)
  float params[MPMHIP_NPARAM] = {};  // layout: include/mpmhip.h
  float initial_aux = 0;             // Jp (snow) | j (water) | logJp (sand)
  float initial_dg = 1;              // src/particles.h:120
};

inline void lame(double E, double nu, float &mu, float &lambda) {
  mu = (float)(E / (2 * (1 + nu)));
  lambda = (float)(E * nu / ((1 + nu) * (1 - 2 * nu)));
}

// MPMParticle::initialize(config) of every registered type, with the reference's defaults.  Throws for an
// unregistered alias (create_instance_placement fails the same way, src/particle_allocator.h:68-74).
inline ParticleType create_particle_type(const std::string &alias, const Config &c, float mass, float vol) {
  if (c.has_key("compressibility"))  // src/particles.h:117-119
    throw std::runtime_error("'compressibility' is deprecated. Use 'initial_dg' instead");
  ParticleType t;
  t.params[0] = mass;
  t.params[1] = vol;
  t.initial_dg = c.get("initial_dg", 1.0f);
  float *p = t.params;
  if (alias == "snow") {  // src/particles.cpp:192-205
    t.material = MPMHIP_SNOW;
    lame(c.get("youngs_modulus", 1.4e5), c.get("poisson_ratio", 0.2), p[2], p[3]);
    p[2] = c.get("mu_0", p[2]); p[3] = c.get("lambda_0", p[3]);
    p[4] = c.get("hardening", 10.0f);
    p[5] = c.get("theta_c", 2.5e-2f); p[6] = c.get("theta_s", 7.5e-3f);
    p[7] = c.get("min_Jp", 0.6f); p[8] = c.get("max_Jp", 20.0f);
    t.initial_aux = c.get("Jp", 1.0f);
  } else if (alias == "linear" || alias == "jelly") {  // :315-321, :383-389
    t.material = alias == "linear" ? MPMHIP_LINEAR : MPMHIP_JELLY;
    lame(c.get("E", 1e5), c.get("nu", 0.3), p[2], p[3]);
  } else if (alias == "water") {  // :448-461 (k defaults to 1e4 in code, README says 1e5)
    t.material = MPMHIP_WATER;
    p[2] = c.get("k", 10000.0f); p[3] = c.get("gamma", 7.0f);
    t.initial_aux = 1.0f;
  } else if (alias == "sand") {  // :570-597 (pi ~ 3.141592653, degrees)
    t.material = MPMHIP_SAND;
    p[2] = c.get("mu_0", 136038.0f); p[3] = c.get("lambda_0", 204057.0f);
    const float sin_phi = std::sin(c.get("friction_angle", 30.0f) / 180.0f * 3.141592653f);
    p[4] = (float)(std::sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi));
    p[5] = c.get("cohesion", 0.0f); p[6] = c.get("beta", 1.0f);
  } else if (alias == "von_mises") {  // :691-699
    t.material = MPMHIP_VON_MISES;
    lame(c.get("youngs_modulus", 5e3), c.get("poisson_ratio", 0.4), p[2], p[3]);
    p[4] = c.get("yield_stress", 1.0f);
  } else if (alias == "elastic") {  // :777-783
    t.material = MPMHIP_ELASTIC;
    lame(c.get("E", 5e3), c.get("nu", 0.4), p[2], p[3]);
    p[4] = c.get("E", 5e3f);  // only reported back by get_debug_info() (verbose .bgeo), :838-840
  } else if (alias == "visco") {  // :57-70
    t.material = MPMHIP_VISCO;
    lame(c.get("youngs_modulus", 4e4), c.get("poisson_ratio", 0.4), p[2], p[3]);
    p[4] = c.get("nu", 10000.0f); p[5] = c.get("kappa", 0.0f); p[6] = c.get("base_delta_t", 1e-4f);
    t.initial_aux = c.get("tau", 1000.0f);
  } else {
    throw std::runtime_error("unknown particle type '" + alias + "'");
  }
  return t;
}

using Matrix3 = std::array<float, 9>;  // row-major

// One particle on the host (src/particles.h:16-190): the state the hot path owns plus its type.  The virtuals of
// the reference become members that evaluate the device code for this one element.
struct MPMParticle {
  Vector3 pos, v;
  Matrix3 dg_e{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, apic_b{};
  float aux = 0;  // Jp | j | logJp
  int32_t id = 0;
  ParticleType type;

  float get_mass() const { return type.params[0]; }
  float get_vol() const { return type.params[1]; }
  Vector3 get_velocity() const { return v; }
  void set_velocity(const Vector3 &nv) { v = nv; }
  // -vol * P(F) * F^T  (src/particles.h:134-137)
  Matrix3 calculate_force(mpmhip_ctx *ctx) const {
    Matrix3 out{};
    const int rc = mpmhip_debug_force(ctx, type.material, type.params, 1, dg_e.data(), &aux, out.data());
    if (rc < 0) throw std::runtime_error(mpmhip_last_error(ctx));
    return out;
  }
  // F <- cdg F followed by the material's return mapping (src/particles.h:139-141)
  void plasticity(mpmhip_ctx *ctx, const Matrix3 &cdg) {
    const int rc = mpmhip_debug_plasticity(ctx, type.material, type.params, 1, cdg.data(), dg_e.data(), &aux, nullptr);
    if (rc < 0) throw std::runtime_error(mpmhip_last_error(ctx));
  }
  // MPMParticle::get_allowed_dt(dx): dx / (c + |v|) with the material's own sound speed c — what the reference's async
  // stepper turns into a block's strength_dt_limit (src/async/async_mpm.cpp:105-111).  Per type (src/particles.cpp):
  //   visco / sand / von_mises / elastic (:136-155,649-665,734-750,814-830)
  //        J = det F, rho = rho0 / J, K = 2 mu / 3 + lambda, c^2 = max(4 mu / (3 rho) + K (1 - log J) / rho0, 1e-20)
  //   snow (:254-278)   J = det F * Jp, (mu, lambda) hardened by exp(h (1 - Jp)), c = sqrt((lambda + 2 mu) / rho)
  //   water (:480-490)  c^2 = k gamma / j^(gamma - 1)
  //   linear / jelly (:343-345,418-420)  0 (no bound: the async stepper stops on them)
  // The device evaluates the same expression per particle (mpmhip_debug_allowed_dt, mpmhip_async_update_dt_limits).
  float get_allowed_dt(float dx) const {
    const float *p = type.params;
    const float u = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float rho0 = get_mass() / get_vol();
    const float *F = dg_e.data();
    const float det = F[0] * (F[4] * F[8] - F[5] * F[7]) - F[1] * (F[3] * F[8] - F[5] * F[6]) + F[2] * (F[3] * F[7] - F[4] * F[6]);
    float c;
    switch (type.material) {
      case MPMHIP_LINEAR:
      case MPMHIP_JELLY:
        return 0.0f;
      case MPMHIP_WATER:
        c = std::sqrt(p[2] * p[3] / std::pow(aux, p[3] - 1.0f));
        break;
      case MPMHIP_SNOW: {
        const float J = det * aux, rho = rho0 / J, e = std::exp(p[4] * (1.0f - aux));
        c = std::sqrt((p[3] * e + 2.0f * p[2] * e) / rho);
        break;
      }
      default: {
        const float rho = rho0 / det, K = 2.0f * p[2] / 3.0f + p[3];
        c = std::sqrt(std::max(4.0f * p[2] / (3.0f * rho) + K * (1.0f - std::log(det)) / rho0, 1e-20f));
      }
    }
    return dx / (c + u);
  }
};

}  // namespace mpm_amd
