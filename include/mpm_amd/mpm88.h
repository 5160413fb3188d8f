// include/mpm_amd/mpm88.h — the 2D dense-grid demo of the reference (mls-mpm88.cpp:5-77: `Particle`, `advance(dt)`,
// `add_object(center, c)`) as a header-only C++ class over the C ABI of libmpmhip (include/mpmhip.h, mpmhip_mpm88_*).
// All arithmetic runs in the device kernels (csrc/k_mpm88.h); this header holds no numerics.
#pragma once
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mpmhip.h"

namespace mpm_amd {

struct Particle88 {  // mls-mpm88.cpp:11-13
  float x[2], v[2], F[4], C[4], Jp;
  int c;  // colour
};

class MLSMPM88 {
 public:
  explicit MLSMPM88(int n = 80, float dt = 1e-4f, bool plastic = true, int device = 0) : n_(n), dt_(dt) {
    if (mpmhip_mpm88_create(n, dt, plastic, device, &h_) < 0) throw std::runtime_error(mpmhip_mpm88_last_error(nullptr));
  }
  MLSMPM88(const MLSMPM88 &) = delete;
  MLSMPM88 &operator=(const MLSMPM88 &) = delete;
  ~MLSMPM88() { mpmhip_mpm88_destroy(h_); }

  // add_object(center, c): `count` particles uniformly in center +- 0.08 (:70-73)
  void add_object(float cx, float cy, int colour, int count = 1000) {
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::vector<float> x(2 * (size_t)count);
    for (int i = 0; i < count; i++) {
      x[2 * i] = u(rng_) * 0.08f + cx;
      x[2 * i + 1] = u(rng_) * 0.08f + cy;
    }
    check(mpmhip_mpm88_add(h_, count, x.data(), nullptr, nullptr, nullptr, nullptr));
    colours_.insert(colours_.end(), (size_t)count, colour);
  }
  void advance(int steps = 1) { check(mpmhip_mpm88_advance(h_, steps)); }  // advance(dt), :16-69
  int64_t num_particles() const { return mpmhip_mpm88_num_particles(h_); }
  std::vector<Particle88> particles() {
    const int64_t n = num_particles();
    std::vector<float> x(2 * n), v(2 * n), F(4 * n), C(4 * n), Jp(n);
    check(mpmhip_mpm88_download(h_, x.data(), v.data(), F.data(), C.data(), Jp.data()));
    std::vector<Particle88> out((size_t)n);
    for (int64_t i = 0; i < n; i++) {
      Particle88 &p = out[(size_t)i];
      for (int k = 0; k < 2; k++) { p.x[k] = x[2 * i + k]; p.v[k] = v[2 * i + k]; }
      for (int k = 0; k < 4; k++) { p.F[k] = F[4 * i + k]; p.C[k] = C[4 * i + k]; }
      p.Jp = Jp[(size_t)i];
      p.c = (size_t)i < colours_.size() ? colours_[(size_t)i] : 0;
    }
    return out;
  }
  int n() const { return n_; }
  float dt() const { return dt_; }

 private:
  void check(int rc) const {
    if (rc < 0) throw std::runtime_error(std::string("libmpmhip (mpm88) error ") + std::to_string(rc) + ": " + mpmhip_mpm88_last_error(h_));
  }
  mpmhip_mpm88 *h_ = nullptr;
  int n_;
  float dt_;
  std::mt19937 rng_{88};
  std::vector<int> colours_;
};

}  // namespace mpm_amd
