// include/mpm_amd/kernel.h — B-spline MPM kernels, host-side value types with the reference's interface
// (yuanming-hu/taichi_mpm src/kernel.h): `MPMKernel<dim, order>(pos_in_grid_units, inv_delta_x)`,
// `get_w / get_dw / get_dw_w(VectorI)`, `static get_stencil_start(real)`, `static inv_D()`, public
// `w_cache / dw_cache`; plus the 3D fast forms `MPMFastKernel32` (src/kernel.h:168-210) and
// `MLSMPMFastKernel32` (src/transfer.cpp:162-191).  Header-only, no dependencies.  The device kernels use the
// same polynomials (taichi_mpm_amd/csrc/mpm_math.h: bspline_weights); tests/cpp/test_host_layer.cpp checks the
// reference's own known-answer tests against these types (src/tests.cpp:13-51, src/transfer.cpp:975-989).
#pragma once
#include <array>
#include <cmath>

namespace mpm_amd {

using real = float;

template <int N, typename T>
struct VectorND : std::array<T, N> {
  VectorND() { this->fill(T(0)); }
  explicit VectorND(T v) { this->fill(v); }
  template <typename... A, typename = std::enable_if_t<sizeof...(A) == N && (N > 1)>>
  VectorND(A... a) : std::array<T, N>{{T(a)...}} {}
  VectorND &operator*=(const VectorND &o) { for (int i = 0; i < N; i++) (*this)[i] *= o[i]; return *this; }
  VectorND &operator+=(const VectorND &o) { for (int i = 0; i < N; i++) (*this)[i] += o[i]; return *this; }
  VectorND operator+(const VectorND &o) const { VectorND r = *this; r += o; return r; }
  VectorND operator-(const VectorND &o) const { VectorND r; for (int i = 0; i < N; i++) r[i] = (*this)[i] - o[i]; return r; }
  VectorND operator*(T s) const { VectorND r; for (int i = 0; i < N; i++) r[i] = (*this)[i] * s; return r; }
  T sum() const { T s = 0; for (int i = 0; i < N; i++) s += (*this)[i]; return s; }
};
using Vector3 = VectorND<3, real>;
using Vector4 = VectorND<4, real>;
using Vector3i = VectorND<3, int>;

// Common part: per-axis weight / derivative tables, products on demand (src/kernel.h:14-71).
template <int dim, int order_>
struct MPMKernelBase {
  static constexpr int D = dim;
  static constexpr int order = order_;
  static constexpr int kernel_size = order + 1;
  using Vector = VectorND<dim, real>;
  using VectorP = VectorND<dim + 1, real>;  // (dw/dx.., w)
  using VectorI = VectorND<dim, int>;

  Vector4 w_cache[dim];   // w_cache[axis][k]: weight of stencil node k along the axis
  Vector4 dw_cache[dim];  // derivative w.r.t. the grid-unit coordinate (multiply by inv_delta_x for world units)
  real inv_delta_x = 1;

  // (dw/dx_0 .. dw/dx_{d-1}, w) of stencil node k: entry i takes the derivative along axis i and the weight
  // along every other axis (src/kernel.h:30-50)
  VectorP get_dw_w(const VectorI &k) const {
    VectorP r;
    for (int i = 0; i <= dim; i++) {
      real p = 1;
      for (int a = 0; a < dim; a++) p *= (a == i) ? dw_cache[a][k[a]] * inv_delta_x : w_cache[a][k[a]];
      r[i] = p;
    }
    return r;
  }
  Vector get_dw(const VectorI &k) const {
    const VectorP t = get_dw_w(k);
    Vector r;
    for (int i = 0; i < dim; i++) r[i] = t[i];
    return r;
  }
  real get_w(const VectorI &k) const {
    real p = 1;
    for (int a = 0; a < dim; a++) p *= w_cache[a][k[a]];
    return p;
  }
  static constexpr real inv_D() { return 6.0f - real(order); }  // src/kernel.h:68-70
};

template <int dim, int order>
struct MPMKernel;

// linear (src/kernel.h:77-100): base = int(x), w = (1-f, f)
template <int dim>
struct MPMKernel<dim, 1> : MPMKernelBase<dim, 1> {
  using Vector = typename MPMKernelBase<dim, 1>::Vector;
  MPMKernel(const Vector &pos, real inv_dx) {
    this->inv_delta_x = inv_dx;
    for (int a = 0; a < dim; a++) {
      const real f = pos[a] - std::floor(pos[a]);
      this->w_cache[a] = Vector4(1 - f, f, 0, 0);
      this->dw_cache[a] = Vector4(-1, 1, 0, 0);
    }
  }
  static int get_stencil_start(real x) { return int(x); }
};

// quadratic (src/kernel.h:103-135): base = int(x - 0.5); with f = fract(x - 0.5) and t = f - (-0.5, 0.5, 1.5)
//   w = (0.5 t0^2 - 1.5 t0 + 1.125, 0.75 - t1^2, 0.5 t2^2 + 1.5 t2 + 1.125),  dw = (t0 - 1.5, -2 t1, t2 + 1.5)
template <int dim>
struct MPMKernel<dim, 2> : MPMKernelBase<dim, 2> {
  using Vector = typename MPMKernelBase<dim, 2>::Vector;
  MPMKernel(const Vector &pos, real inv_dx) {
    this->inv_delta_x = inv_dx;
    for (int a = 0; a < dim; a++) {
      const real s = pos[a] - 0.5f, f = s - std::floor(s);
      const real t0 = f + 0.5f, t1 = f - 0.5f, t2 = f - 1.5f;
      this->w_cache[a] = Vector4(0.5f * t0 * t0 - 1.5f * t0 + 1.125f, 0.75f - t1 * t1, 0.5f * t2 * t2 + 1.5f * t2 + 1.125f, 0);
      this->dw_cache[a] = Vector4(t0 - 1.5f, -2.0f * t1, t2 + 1.5f, 0);
    }
  }
  static int get_stencil_start(real x) { return int(x - 0.5f); }
};

// cubic (src/kernel.h:138-165): base = int(x) - 1; with f = fract(x), distances d = (f+1, f, 1-f, 2-f)
template <int dim>
struct MPMKernel<dim, 3> : MPMKernelBase<dim, 3> {
  using Vector = typename MPMKernelBase<dim, 3>::Vector;
  MPMKernel(const Vector &pos, real inv_dx) {
    this->inv_delta_x = inv_dx;
    for (int a = 0; a < dim; a++) {
      const real f = pos[a] - std::floor(pos[a]);
      const real d[4] = {f + 1, f, 1 - f, 2 - f};
      // N(d) = 1/2 d^3 - d^2 + 2/3 (d < 1);  1/6 (2 - d)^3 (1 <= d < 2); sign of the derivative follows the side
      auto near = [](real x) { return 0.5f * x * x * x - x * x + 2.0f / 3.0f; };
      auto far = [](real x) { const real u = 2 - x; return u * u * u / 6.0f; };
      auto dnear = [](real x) { return 1.5f * x * x - 2 * x; };
      auto dfar = [](real x) { const real u = 2 - x; return -0.5f * u * u; };
      this->w_cache[a] = Vector4(far(d[0]), near(d[1]), near(d[2]), far(d[3]));
      this->dw_cache[a] = Vector4(dfar(d[0]), dnear(d[1]), -dnear(d[2]), -dfar(d[3]));
    }
  }
  static int get_stencil_start(real x) { return int(x) - 1; }
};

// 3D quadratic, all 27 (dw/dx, dw/dy, dw/dz, w) at once (src/kernel.h:168-210); node n = (i*3 + j)*3 + k
struct MPMFastKernel32 {
  Vector4 kernels[3][3][3];
  MPMFastKernel32(const Vector3 &pos, real inv_dx) {
    const MPMKernel<3, 2> base(pos, inv_dx);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) kernels[i][j][k] = base.get_dw_w(Vector3i(i, j, k));
  }
  Vector4 get_dw_w(const Vector3i &k) const { return kernels[k[0]][k[1]][k[2]]; }
  static int get_stencil_start(real x) { return int(x - 0.5f); }
};

// 3D quadratic weights only, as MLS-MPM needs them (src/transfer.cpp:162-191): kernels[i][j] = w(i,j,0..2).
// Constructed from the position RELATIVE to the base cell, in [0.5, 1.5)^3 (src/transfer.cpp:490,518).
struct MLSMPMFastKernel32 {
  Vector4 kernels[3][3];
  explicit MLSMPMFastKernel32(const Vector3 &rel_pos) {
    real w[3][3];
    for (int a = 0; a < 3; a++) {
      const real p = rel_pos[a] - 0.5f, t0 = p + 0.5f, t1 = p - 0.5f, t2 = p - 1.5f;
      w[a][0] = 0.5f * t0 * t0 - 1.5f * t0 + 1.125f;
      w[a][1] = 0.75f - t1 * t1;
      w[a][2] = 0.5f * t2 * t2 + 1.5f * t2 + 1.125f;
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const real wij = w[0][i] * w[1][j];
        kernels[i][j] = Vector4(wij * w[2][0], wij * w[2][1], wij * w[2][2], 0);
      }
  }
};

}  // namespace mpm_amd
