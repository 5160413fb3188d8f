// oracle/taichi_shim/taichi/common/util.h — TEST INFRASTRUCTURE, never shipped, never on the product path.
//
// A one-file stand-in for the part of the LEGACY taichi C++ core (un-vendored; pins found in the reference tree:
// README.md:215 -> taichi legacy @ 5ab90f03..., scripts/mls-cpic/sand_paddles.py:1-2) that the reference's MPM
// sources use, written from the call sites in /root/reference/src/*.{h,cpp} so that those sources compile WHERE
// THEY LIE into oracle/_ref/libmpm_ref.so (oracle/Makefile: ref_mpm).  Every <taichi/...> header in this directory
// forwards here.  Nothing of the reference is copied; this file holds only the generic plumbing the reference
// expects from its host library:
//   * small fixed-size vectors / matrices (column-major, Vector3f/Vector4f on SSE registers, as the reference's
//     SIMD code requires: `p.pos.v`, `reinterpret_cast<Matrix &>(__m128[3])`, src/transfer.cpp:490-545,929)
//   * Config (string dictionary), the Unit / interface factory macros, serialization macros (no-ops)
//   * RegionND / IndexND / ArrayND, Simulation<dim> base, analytic DynamicLevelSet, functional RigidBody (rigid_body_shim.h), inert Texture / Mesh
//   * ThreadedTaskManager / tbb::parallel_* on OpenMP, Profiler (accumulates per-name wall time)
//   * svd / polar_decomp: OURS (see below) — the one numerical routine of the hot path that lives in the absent
//     library.  Computed in double precision by a Jacobi eigen-solve of A^T A (3D) / closed form (2D) and rounded to
//     float, with the convention det U = det V = +1, |sigma| descending, sign on the last sigma.  Everything the
//     reference derives from it (U f(S) V^T, R = U V^T, prod sigma) is convention-free for det F > 0.
// What the compiled reference therefore PINS: kernel.h, mpm_fwd.h (friction_project), particles.cpp (all eight
// constitutive models and return maps), transfer.cpp (P2G / G2P, generic and SIMD-optimised), mpm.cpp (sort, grid
// normalisation, boundary conditions, deletion, substep sequence).  What it does not pin: the SVD algorithm itself
// and the SAMPLED level set of the taichi core (here: exact distance functions of the same shapes).
#pragma once

#include <immintrin.h>

#include <algorithm>
#include <array>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include <omp.h>
#include <parallel/algorithm>

#define TC_NAMESPACE_BEGIN namespace taichi {
#define TC_NAMESPACE_END }
#define TC_FORCE_INLINE inline __attribute__((always_inline))
#define TC_ALIGNED(x) __attribute__((aligned(x)))

namespace taichi {

using real = float;
using float32 = float;
using float64 = double;
using int8 = std::int8_t;
using int16 = std::int16_t;
using int32 = std::int32_t;
using int64 = std::int64_t;
using uint8 = std::uint8_t;
using uint16 = std::uint16_t;
using uint32 = std::uint32_t;
using uint64 = std::uint64_t;
using uint = unsigned int;

constexpr real operator"" _f(long double v) { return (real)v; }
constexpr real operator"" _f(unsigned long long v) { return (real)v; }
constexpr float32 operator"" _f32(long double v) { return (float32)v; }
constexpr float64 operator"" _f64(long double v) { return (float64)v; }
constexpr real eps = 1e-6f;

using std::abs;
using std::exp;
using std::log;
using std::max;
using std::min;
using std::sqrt;
using std::cos;
using std::sin;
using std::tan;
using std::floor;
using std::pow;

template <int n, typename T>
constexpr T pow(T a) {
  T r = 1;
  for (int i = 0; i < n; i++) r *= a;
  return r;
}
template <typename T>
constexpr T sqr(T a) { return a * a; }
template <typename T>
inline T clamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline real fract(real x) { return x - std::floor(x); }
inline real rand() { return (real)(::rand() / (RAND_MAX + 1.0)); }
inline int rand_int() { return ::rand(); }
template <typename T>
inline void trash(T &&) {}
inline std::string absolute_path(const std::string &s) { return s; }

struct ShimError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] inline void shim_fail(const char *what, const char *file, int line) {
  char buf[1024];
  std::snprintf(buf, sizeof buf, "%s (%s:%d)", what, file, line);
  throw ShimError(buf);
}

#define TC_P(...) ((void)0)
#define TC_TRACE(...) ((void)0)
#define TC_DEBUG(...) ((void)0)
#define TC_INFO(...) ((void)0)
#define TC_WARN(...) ((void)0)
#define TC_TAG ((void)0)
#define TC_ERROR(...) ::taichi::shim_fail("TC_ERROR: " #__VA_ARGS__, __FILE__, __LINE__)
#define TC_NOT_IMPLEMENTED ::taichi::shim_fail("TC_NOT_IMPLEMENTED", __FILE__, __LINE__);
#define TC_STOP ::taichi::shim_fail("TC_STOP", __FILE__, __LINE__)
#define TC_ASSERT(x) \
  do { if (!(x)) ::taichi::shim_fail("TC_ASSERT failed: " #x, __FILE__, __LINE__); } while (0)
#define TC_ASSERT_INFO(x, ...) /* a complete statement: src/mpm_rigid_body.cpp:22-23 uses it without a semicolon */ \
  { if (!(x)) ::taichi::shim_fail("TC_ASSERT_INFO failed: " #x, __FILE__, __LINE__); }
#define TC_STATIC_IF(x) if constexpr (x)
#define TC_STATIC_ELSE else
#define TC_STATIC_END_IF
#define TC_REPEAT27(F) \
  F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17) F(18) F(19) F(20) F(21) F(22) F(23) F(24) F(25) F(26)
#define TC_LOAD_CONFIG(name, dflt) name = config.get(#name, dflt)
#define TC_ERROR_IF(cond, ...) do { if (cond) ::taichi::shim_fail("TC_ERROR_IF: " #cond, __FILE__, __LINE__); } while (0)

// serialization: not part of the path; the declarations only have to parse
struct BinaryOutputSerializer {
  void initialize() {}
  void finalize() {}
  void write_to_file(const std::string &) {}
  template <typename T> void operator()(const T &) {}
};
struct BinaryInputSerializer {
  void initialize(const std::string &) {}
  void finalize() {}
  template <typename T> void operator()(T &) {}
};
#define TC_IO(...) ((void)0)
#define TC_IO_DEF(...)
#define TC_IO_DEF_VIRT(...)
#define TC_IO_DEF_WITH_BASE(...)
#define TC_IO_DECL template <typename TC_SERIALIZER_> void io(TC_SERIALIZER_ &serializer) const
#define TC_IO_DECL_VIRT template <typename TC_SERIALIZER_> void io_virt_(TC_SERIALIZER_ &serializer) const
#define TC_SERIALIZER_IS(T) (std::is_same<std::decay_t<decltype(serializer)>, T>::value)
template <typename T> void write_to_binary_file(const T &, const std::string &) {}
template <typename T> void read_from_binary_file(T &, const std::string &) {}
template <int dim> struct Element {};

namespace bit {
constexpr bool is_power_of_two(long long x) { return x > 0 && (x & (x - 1)) == 0; }
inline int log2int(unsigned long long x) { int r = 0; while (x > 1) { x >>= 1; r++; } return r; }
}  // namespace bit
namespace math {
inline real radians(real deg) { return deg * (real)(M_PI / 180.0); }
inline real degrees(real rad) { return rad * (real)(180.0 / M_PI); }
}  // namespace math

struct Spinlock {
  std::uint16_t flag = 0;
  void lock() {}
  void unlock() {}
};
static_assert(sizeof(Spinlock) == 2, "GridState<3> must stay 32 bytes (src/mpm_fwd.h:69-119)");

// ------------------------------------------------------------------------------------------------ vectors
template <int dim, typename T> struct VecStore { T d[dim]; };
template <typename T> struct VecStore<1, T> { union { T d[1]; struct { T x; }; }; };
template <typename T> struct VecStore<2, T> { union { T d[2]; struct { T x, y; }; }; };
template <typename T> struct VecStore<3, T> { union { T d[3]; struct { T x, y, z; }; }; };
template <typename T> struct VecStore<4, T> { union { T d[4]; struct { T x, y, z, w; }; }; };
template <> struct alignas(16) VecStore<3, float> { union { __m128 v; float d[4]; struct { float x, y, z, pad_; }; }; };
template <> struct alignas(16) VecStore<4, float> { union { __m128 v; float d[4]; struct { float x, y, z, w; }; }; };

template <int dim, typename T>
struct VectorND : public VecStore<dim, T> {
  static constexpr bool simd = std::is_same<T, float>::value && (dim == 3 || dim == 4);
  static constexpr int D = dim;
  using VecStore<dim, T>::d;

  TC_FORCE_INLINE VectorND() { zero(); }
  TC_FORCE_INLINE void zero() {
    if constexpr (simd) this->v = _mm_setzero_ps();
    else for (int i = 0; i < dim; i++) d[i] = T(0);
  }
  template <typename S, std::enable_if_t<std::is_arithmetic<S>::value, int> = 0>
  TC_FORCE_INLINE explicit VectorND(S a) {
    if constexpr (simd) this->v = _mm_set1_ps((float)a);
    else for (int i = 0; i < dim; i++) d[i] = T(a);
  }
  TC_FORCE_INLINE VectorND(T a, T b) { static_assert(dim == 2, ""); zero(); d[0] = a; d[1] = b; }
  TC_FORCE_INLINE VectorND(T a, T b, T c) { static_assert(dim == 3, ""); zero(); d[0] = a; d[1] = b; d[2] = c; }
  TC_FORCE_INLINE VectorND(T a, T b, T c, T e) { static_assert(dim == 4, ""); d[0] = a; d[1] = b; d[2] = c; d[3] = e; }
  template <typename F, std::enable_if_t<std::is_convertible<decltype(std::declval<F>()(0)), T>::value, int> = 0>
  TC_FORCE_INLINE explicit VectorND(const F &f) { zero(); for (int i = 0; i < dim; i++) d[i] = f(i); }
  // (shorter, last): VectorP(v, m)
  TC_FORCE_INLINE VectorND(const VectorND<dim - 1, T> &o, T last) { zero(); for (int i = 0; i < dim - 1; i++) d[i] = o.d[i]; d[dim - 1] = last; }
  // truncate or zero-pad: Vector(v_and_m), VectorP(v)
  template <int d2, std::enable_if_t<d2 != dim, int> = 0>
  TC_FORCE_INLINE explicit VectorND(const VectorND<d2, T> &o) { zero(); for (int i = 0; i < (d2 < dim ? d2 : dim); i++) d[i] = o.d[i]; }
  template <typename U, std::enable_if_t<!std::is_same<U, T>::value, int> = 0>
  TC_FORCE_INLINE explicit VectorND(const VectorND<dim, U> &o) { zero(); for (int i = 0; i < dim; i++) d[i] = (T)o.d[i]; }
  template <typename U, std::enable_if_t<std::is_arithmetic<U>::value, int> = 0>
  TC_FORCE_INLINE VectorND(const std::array<U, dim> &o) { zero(); for (int i = 0; i < dim; i++) d[i] = (T)o[i]; }
  template <bool S = simd, std::enable_if_t<S, int> = 0>
  TC_FORCE_INLINE VectorND(const __m128 &m) { this->v = m; }
  template <bool S = simd, std::enable_if_t<S, int> = 0>
  TC_FORCE_INLINE operator __m128() const { return this->v; }
  TC_FORCE_INLINE operator std::array<T, dim>() const { std::array<T, dim> a; for (int i = 0; i < dim; i++) a[i] = d[i]; return a; }

  TC_FORCE_INLINE T &operator[](int i) { return d[i]; }
  TC_FORCE_INLINE const T &operator[](int i) const { return d[i]; }

#define TC_SHIM_VEC_OP(op, sse)                                                                        \
  TC_FORCE_INLINE VectorND operator op(const VectorND &o) const {                                      \
    VectorND r;                                                                                        \
    if constexpr (simd) r.v = sse(this->v, o.v);                                                       \
    else for (int i = 0; i < dim; i++) r.d[i] = d[i] op o.d[i];                                        \
    return r;                                                                                          \
  }                                                                                                    \
  TC_FORCE_INLINE VectorND &operator op##=(const VectorND &o) { *this = *this op o; return *this; }
  TC_SHIM_VEC_OP(+, _mm_add_ps)
  TC_SHIM_VEC_OP(-, _mm_sub_ps)
  TC_SHIM_VEC_OP(*, _mm_mul_ps)
#undef TC_SHIM_VEC_OP
  TC_FORCE_INLINE VectorND operator/(const VectorND &o) const { VectorND r; for (int i = 0; i < dim; i++) r.d[i] = d[i] / o.d[i]; return r; }
  TC_FORCE_INLINE VectorND &operator/=(const VectorND &o) { *this = *this / o; return *this; }
  TC_FORCE_INLINE VectorND operator-() const { VectorND r; for (int i = 0; i < dim; i++) r.d[i] = -d[i]; return r; }
  TC_FORCE_INLINE VectorND operator*(T s) const { return *this * VectorND(s); }
  TC_FORCE_INLINE VectorND operator/(T s) const { VectorND r; for (int i = 0; i < dim; i++) r.d[i] = d[i] / s; return r; }
  TC_FORCE_INLINE VectorND &operator*=(T s) { *this = *this * s; return *this; }
  TC_FORCE_INLINE VectorND &operator/=(T s) { *this = *this / s; return *this; }
  TC_FORCE_INLINE bool operator==(const VectorND &o) const { for (int i = 0; i < dim; i++) if (d[i] != o.d[i]) return false; return true; }
  TC_FORCE_INLINE bool operator!=(const VectorND &o) const { return !(*this == o); }
  TC_FORCE_INLINE bool operator<(const VectorND &o) const { for (int i = 0; i < dim; i++) if (!(d[i] < o.d[i])) return false; return true; }
  TC_FORCE_INLINE bool operator<=(const VectorND &o) const { for (int i = 0; i < dim; i++) if (!(d[i] <= o.d[i])) return false; return true; }

  TC_FORCE_INLINE T min() const { T r = d[0]; for (int i = 1; i < dim; i++) r = std::min(r, d[i]); return r; }
  TC_FORCE_INLINE T max() const { T r = d[0]; for (int i = 1; i < dim; i++) r = std::max(r, d[i]); return r; }
  TC_FORCE_INLINE T sum() const { T r = d[0]; for (int i = 1; i < dim; i++) r += d[i]; return r; }
  TC_FORCE_INLINE T prod() const { T r = d[0]; for (int i = 1; i < dim; i++) r *= d[i]; return r; }
  TC_FORCE_INLINE T dot(const VectorND &o) const { T r = d[0] * o.d[0]; for (int i = 1; i < dim; i++) r += d[i] * o.d[i]; return r; }
  TC_FORCE_INLINE T length2() const { return dot(*this); }
  TC_FORCE_INLINE T length() const { return std::sqrt(length2()); }
  TC_FORCE_INLINE T abs_max() const { T r = std::abs(d[0]); for (int i = 1; i < dim; i++) r = std::max(r, std::abs(d[i])); return r; }
  TC_FORCE_INLINE VectorND abs() const { VectorND r; for (int i = 0; i < dim; i++) r.d[i] = std::abs(d[i]); return r; }
  template <typename F> TC_FORCE_INLINE VectorND map(F f) const { VectorND r; for (int i = 0; i < dim; i++) r.d[i] = f(d[i]); return r; }
  TC_FORCE_INLINE VectorND clamp(const VectorND &lo, const VectorND &hi) const {
    VectorND r; for (int i = 0; i < dim; i++) r.d[i] = std::min(std::max(d[i], lo.d[i]), hi.d[i]); return r;
  }
  TC_FORCE_INLINE bool abnormal() const { for (int i = 0; i < dim; i++) if (!std::isfinite((double)d[i])) return true; return false; }
  template <typename U> TC_FORCE_INLINE VectorND<dim, U> cast() const { VectorND<dim, U> r; for (int i = 0; i < dim; i++) r.d[i] = (U)d[i]; return r; }
  static TC_FORCE_INLINE VectorND axis(int a) { VectorND r; r.d[a] = T(1); return r; }
  static VectorND rand() { VectorND r; for (int i = 0; i < dim; i++) r.d[i] = (T)taichi::rand(); return r; }
};
template <int dim, typename T> TC_FORCE_INLINE VectorND<dim, T> operator*(T s, const VectorND<dim, T> &v) { return v * s; }
template <int dim> TC_FORCE_INLINE VectorND<dim, float> operator*(double s, const VectorND<dim, float> &v) { return v * (float)s; }
template <int dim> TC_FORCE_INLINE VectorND<dim, float> operator*(const VectorND<dim, float> &v, double s) { return v * (float)s; }
template <int dim> TC_FORCE_INLINE VectorND<dim, float> operator*(int s, const VectorND<dim, float> &v) { return v * (float)s; }
template <int dim, typename T> TC_FORCE_INLINE T dot(const VectorND<dim, T> &a, const VectorND<dim, T> &b) { return a.dot(b); }
template <int dim, typename T> TC_FORCE_INLINE T length(const VectorND<dim, T> &a) { return a.length(); }
template <int dim, typename T> TC_FORCE_INLINE T length2(const VectorND<dim, T> &a) { return a.length2(); }
template <int dim, typename T> TC_FORCE_INLINE VectorND<dim, T> min(const VectorND<dim, T> &a, const VectorND<dim, T> &b) {
  VectorND<dim, T> r; for (int i = 0; i < dim; i++) r.d[i] = std::min(a.d[i], b.d[i]); return r;
}
template <int dim, typename T> TC_FORCE_INLINE VectorND<dim, T> max(const VectorND<dim, T> &a, const VectorND<dim, T> &b) {
  VectorND<dim, T> r; for (int i = 0; i < dim; i++) r.d[i] = std::max(a.d[i], b.d[i]); return r;
}
template <int dim, typename T> TC_FORCE_INLINE VectorND<dim, T> fract(const VectorND<dim, T> &a) {
  VectorND<dim, T> r; for (int i = 0; i < dim; i++) r.d[i] = a.d[i] - std::floor(a.d[i]); return r;
}
template <int dim, typename T> TC_FORCE_INLINE VectorND<dim, T> normalized(const VectorND<dim, T> &a) { return a / a.length(); }
// a * b + c, fused where the type lives in an SSE register (the reference is built with FMA)
template <int dim, typename T>
TC_FORCE_INLINE VectorND<dim, T> fused_mul_add(const VectorND<dim, T> &a, const VectorND<dim, T> &b, const VectorND<dim, T> &c) {
  if constexpr (VectorND<dim, T>::simd) return VectorND<dim, T>(_mm_fmadd_ps(a.v, b.v, c.v));
  else return a * b + c;
}
template <int dim, typename T> std::array<T, dim> to_std_array(const VectorND<dim, T> &v) { return (std::array<T, dim>)v; }

using Vector2 = VectorND<2, real>; using Vector3 = VectorND<3, real>; using Vector4 = VectorND<4, real>;
using Vector2f = VectorND<2, float>; using Vector3f = VectorND<3, float>; using Vector4f = VectorND<4, float>;
using Vector2i = VectorND<2, int>; using Vector3i = VectorND<3, int>; using Vector4i = VectorND<4, int>;
using Vector2d = VectorND<2, double>; using Vector3d = VectorND<3, double>;

// ------------------------------------------------------------------------------------------------ matrices (column major)
template <int dim, typename T>
struct MatrixND {
  using Vector = VectorND<dim, T>;
  Vector d[dim];  // columns
  TC_FORCE_INLINE MatrixND() {}
  // implicit on purpose: the legacy core converts scalars to diagonal matrices silently, and the reference relies on
  // it where it passes a block index to damp_affine_momemtum (src/transfer.cpp:925-926)
  template <typename S, std::enable_if_t<std::is_arithmetic<S>::value, int> = 0>
  TC_FORCE_INLINE MatrixND(S a) { for (int i = 0; i < dim; i++) d[i][i] = (T)a; }
  // a smaller matrix into the top-left corner (MatrixP(rotation_matrix), src/rigid_body_solver.h:131)
  template <int d2, std::enable_if_t<(d2 < dim), int> = 0>
  TC_FORCE_INLINE explicit MatrixND(const MatrixND<d2, T> &o) { for (int c = 0; c < d2; c++) for (int r = 0; r < d2; r++) d[c][r] = o[c][r]; }
  TC_FORCE_INLINE explicit MatrixND(const Vector &diag) { for (int i = 0; i < dim; i++) d[i][i] = diag[i]; }
  TC_FORCE_INLINE MatrixND(const Vector &c0, const Vector &c1) { static_assert(dim == 2, ""); d[0] = c0; d[1] = c1; }
  TC_FORCE_INLINE MatrixND(const Vector &c0, const Vector &c1, const Vector &c2) { static_assert(dim == 3, ""); d[0] = c0; d[1] = c1; d[2] = c2; }
  TC_FORCE_INLINE Vector &operator[](int i) { return d[i]; }
  TC_FORCE_INLINE const Vector &operator[](int i) const { return d[i]; }
  TC_FORCE_INLINE MatrixND operator+(const MatrixND &o) const { MatrixND r; for (int i = 0; i < dim; i++) r.d[i] = d[i] + o.d[i]; return r; }
  TC_FORCE_INLINE MatrixND operator-(const MatrixND &o) const { MatrixND r; for (int i = 0; i < dim; i++) r.d[i] = d[i] - o.d[i]; return r; }
  TC_FORCE_INLINE MatrixND operator-() const { MatrixND r; for (int i = 0; i < dim; i++) r.d[i] = -d[i]; return r; }
  TC_FORCE_INLINE MatrixND &operator+=(const MatrixND &o) { for (int i = 0; i < dim; i++) d[i] += o.d[i]; return *this; }
  TC_FORCE_INLINE MatrixND &operator-=(const MatrixND &o) { for (int i = 0; i < dim; i++) d[i] -= o.d[i]; return *this; }
  TC_FORCE_INLINE MatrixND operator*(T s) const { MatrixND r; for (int i = 0; i < dim; i++) r.d[i] = d[i] * s; return r; }
  TC_FORCE_INLINE Vector operator*(const Vector &v) const {
    Vector r = d[0] * v[0];
    for (int i = 1; i < dim; i++) r += d[i] * v[i];
    return r;
  }
  TC_FORCE_INLINE MatrixND operator*(const MatrixND &o) const { MatrixND r; for (int j = 0; j < dim; j++) r.d[j] = (*this) * o.d[j]; return r; }
  TC_FORCE_INLINE bool operator==(const MatrixND &o) const { for (int i = 0; i < dim; i++) if (d[i] != o.d[i]) return false; return true; }
  TC_FORCE_INLINE MatrixND transposed() const { MatrixND r; for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) r.d[i][j] = d[j][i]; return r; }
  TC_FORCE_INLINE Vector diag() const { Vector r; for (int i = 0; i < dim; i++) r[i] = d[i][i]; return r; }
  TC_FORCE_INLINE T trace() const { return diag().sum(); }
  TC_FORCE_INLINE T sum() const { T s = 0; for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) s += d[i][j]; return s; }
  TC_FORCE_INLINE T frobenius_norm2() const { T s = 0; for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) s += d[i][j] * d[i][j]; return s; }
  TC_FORCE_INLINE T frobenius_norm() const { return std::sqrt(frobenius_norm2()); }
  TC_FORCE_INLINE MatrixND elementwise_product(const MatrixND &o) const { MatrixND r; for (int i = 0; i < dim; i++) r.d[i] = d[i] * o.d[i]; return r; }
  TC_FORCE_INLINE bool abnormal() const { for (int i = 0; i < dim; i++) if (d[i].abnormal()) return true; return false; }
  // column c = a * b[c]:  (a (x) b)(row r, col c) = a[r] b[c]
  static TC_FORCE_INLINE MatrixND outer_product(const Vector &a, const Vector &b) { MatrixND r; for (int c = 0; c < dim; c++) r.d[c] = a * b[c]; return r; }
};
template <int dim, typename T> TC_FORCE_INLINE MatrixND<dim, T> operator*(T s, const MatrixND<dim, T> &m) { return m * s; }
template <int dim> TC_FORCE_INLINE MatrixND<dim, float> operator*(double s, const MatrixND<dim, float> &m) { return m * (float)s; }
template <int dim> TC_FORCE_INLINE MatrixND<dim, float> operator*(int s, const MatrixND<dim, float> &m) { return m * (float)s; }
template <int dim> TC_FORCE_INLINE MatrixND<dim, float> operator*(const MatrixND<dim, float> &m, double s) { return m * (float)s; }
template <int dim, typename T> TC_FORCE_INLINE MatrixND<dim, T> transposed(const MatrixND<dim, T> &m) { return m.transposed(); }
template <int dim, typename T> TC_FORCE_INLINE MatrixND<dim, T> transpose(const MatrixND<dim, T> &m) { return m.transposed(); }
template <typename T> TC_FORCE_INLINE T determinant(const MatrixND<2, T> &m) { return m[0][0] * m[1][1] - m[1][0] * m[0][1]; }
template <typename T> TC_FORCE_INLINE T determinant(const MatrixND<3, T> &m) {
  return m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
         m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}
template <typename T> inline MatrixND<2, T> inversed(const MatrixND<2, T> &m) {
  const T id = T(1) / determinant(m);
  return MatrixND<2, T>(VectorND<2, T>(m[1][1] * id, -m[0][1] * id), VectorND<2, T>(-m[1][0] * id, m[0][0] * id));
}
template <typename T> inline MatrixND<3, T> inversed(const MatrixND<3, T> &m) {
  // adjugate / determinant; entry (row r, col c) = m[c][r]
  auto a = [&](int r, int c) { return m[c][r]; };
  const T det = determinant(m), id = T(1) / det;
  MatrixND<3, T> o;
  o[0][0] = (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) * id; o[1][0] = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id; o[2][0] = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
  o[0][1] = (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) * id; o[1][1] = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id; o[2][1] = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
  o[0][2] = (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)) * id; o[1][2] = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id; o[2][2] = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
  return o;
}
// 4x4 (the least-squares fit of src/rigid_transfer.cpp:250-252): OURS like svd — Gauss-Jordan with partial pivoting in
// double precision, rounded to T
template <typename T> inline T determinant(const MatrixND<4, T> &m) {
  double a[4][4], det = 1;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) a[r][c] = m[c][r];
  for (int k = 0; k < 4; k++) {
    int p = k;
    for (int r = k + 1; r < 4; r++) if (std::abs(a[r][k]) > std::abs(a[p][k])) p = r;
    if (a[p][k] == 0) return T(0);
    if (p != k) { for (int c = 0; c < 4; c++) std::swap(a[p][c], a[k][c]); det = -det; }
    det *= a[k][k];
    for (int r = k + 1; r < 4; r++) { const double f = a[r][k] / a[k][k]; for (int c = k; c < 4; c++) a[r][c] -= f * a[k][c]; }
  }
  return (T)det;
}
template <typename T> inline MatrixND<4, T> inversed(const MatrixND<4, T> &m) {
  double a[4][8];
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = m[c][r]; a[r][4 + c] = r == c; }
  for (int k = 0; k < 4; k++) {
    int p = k;
    for (int r = k + 1; r < 4; r++) if (std::abs(a[r][k]) > std::abs(a[p][k])) p = r;
    if (p != k) for (int c = 0; c < 8; c++) std::swap(a[p][c], a[k][c]);
    const double inv = 1.0 / a[k][k];
    for (int c = 0; c < 8; c++) a[k][c] *= inv;
    for (int r = 0; r < 4; r++) if (r != k) { const double f = a[r][k]; for (int c = 0; c < 8; c++) a[r][c] -= f * a[k][c]; }
  }
  MatrixND<4, T> o;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o[c][r] = (T)a[r][4 + c];
  return o;
}
template <int dim, typename T> inline MatrixND<dim, T> inverse(const MatrixND<dim, T> &m) { return inversed(m); }
using Matrix2 = MatrixND<2, real>; using Matrix3 = MatrixND<3, real>;
using Matrix2f = MatrixND<2, float>; using Matrix3f = MatrixND<3, float>;

// ------------------------------------------------------------------------------------------------ svd / polar (OURS)
namespace shim_svd {
// A = U diag(s) V^T in double; det U = det V = +1; |s| descending; the sign of det A on the last s.
inline void svd3(const double A[3][3], double U[3][3], double s[3], double V[3][3]) {
  double S[3][3];  // A^T A
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { S[i][j] = 0; for (int k = 0; k < 3; k++) S[i][j] += A[k][i] * A[k][j]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = i == j;
  for (int sweep = 0; sweep < 30; sweep++) {
    const double off = std::abs(S[0][1]) + std::abs(S[0][2]) + std::abs(S[1][2]);
    if (off <= 1e-300 || off < 1e-17 * (std::abs(S[0][0]) + std::abs(S[1][1]) + std::abs(S[2][2]))) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (S[p][q] == 0.0) continue;
      const double theta = (S[q][q] - S[p][p]) / (2 * S[p][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1));
      const double c = 1 / std::sqrt(t * t + 1), sn = t * c;
      for (int k = 0; k < 3; k++) { const double a = S[k][p], b = S[k][q]; S[k][p] = c * a - sn * b; S[k][q] = sn * a + c * b; }
      for (int k = 0; k < 3; k++) { const double a = S[p][k], b = S[q][k]; S[p][k] = c * a - sn * b; S[q][k] = sn * a + c * b; }
      for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
    }
  }
  int ord[3] = {0, 1, 2};
  std::sort(ord, ord + 3, [&](int a, int b) { return S[a][a] > S[b][b]; });
  double Vs[3][3];
  for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) Vs[i][j] = V[i][ord[j]];
  const double detV = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) - Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                      Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
  if (detV < 0) for (int i = 0; i < 3; i++) Vs[i][2] = -Vs[i][2];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = Vs[i][j];
  // columns of A V, the first two normalised (Gram-Schmidt), the third = their cross product (det U = +1)
  double B[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { B[i][j] = 0; for (int k = 0; k < 3; k++) B[i][j] += A[i][k] * V[k][j]; }
  auto col_norm = [&](int j) { return std::sqrt(B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j]); };
  double n0 = col_norm(0);
  if (n0 < 1e-300) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i][j] = i == j; s[0] = s[1] = s[2] = 0; return; }
  for (int i = 0; i < 3; i++) U[i][0] = B[i][0] / n0;
  double proj = 0;
  for (int i = 0; i < 3; i++) proj += U[i][0] * B[i][1];
  double w[3];
  for (int i = 0; i < 3; i++) w[i] = B[i][1] - proj * U[i][0];
  double n1 = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (n1 < 1e-150 * n0 || n1 < 1e-300) {  // rank one: any unit vector orthogonal to U0
    int k = std::abs(U[0][0]) < std::abs(U[1][0]) ? (std::abs(U[0][0]) < std::abs(U[2][0]) ? 0 : 2) : (std::abs(U[1][0]) < std::abs(U[2][0]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[k] = 1;
    double pe = U[k][0];
    for (int i = 0; i < 3; i++) w[i] = e[i] - pe * U[i][0];
    n1 = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  }
  for (int i = 0; i < 3; i++) U[i][1] = w[i] / n1;
  U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
  U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
  U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  for (int j = 0; j < 3; j++) { s[j] = 0; for (int i = 0; i < 3; i++) s[j] += U[i][j] * B[i][j]; }
}
}  // namespace shim_svd

// 3D: m = u * sig * v^T (sig diagonal)
inline void svd(const MatrixND<3, real> &m, MatrixND<3, real> &u, MatrixND<3, real> &sig, MatrixND<3, real> &v) {
  double A[3][3], U[3][3], s[3], V[3][3];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = m[c][r];
  shim_svd::svd3(A, U, s, V);
  sig = MatrixND<3, real>();
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { u[c][r] = (real)U[r][c]; v[c][r] = (real)V[r][c]; }
  for (int i = 0; i < 3; i++) sig[i][i] = (real)s[i];
}
inline void polar_decomp(const MatrixND<3, real> &m, MatrixND<3, real> &R, MatrixND<3, real> &S) {
  double A[3][3], U[3][3], s[3], V[3][3];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = m[c][r];
  shim_svd::svd3(A, U, s, V);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
    double rr = 0, ss = 0;
    for (int k = 0; k < 3; k++) { rr += U[r][k] * V[c][k]; ss += V[r][k] * s[k] * V[c][k]; }
    R[c][r] = (real)rr; S[c][r] = (real)ss;
  }
}
// 2D: closed forms in double.  R = [[c,-s],[s,c]] with (c,s) ~ (m00 + m11, m10 - m01); S = R^T m.
inline void polar_decomp(const MatrixND<2, real> &m, MatrixND<2, real> &R, MatrixND<2, real> &S) {
  const double a = m[0][0], b = m[1][0], c = m[0][1], d = m[1][1];  // rows: [a b; c d]
  double x = a + d, y = c - b, den = std::sqrt(x * x + y * y);
  double co = 1, si = 0;
  if (den > 0) { co = x / den; si = y / den; }
  R = MatrixND<2, real>(VectorND<2, real>((real)co, (real)si), VectorND<2, real>((real)-si, (real)co));
  // S = R^T m
  const double s00 = co * a + si * c, s01 = co * b + si * d, s10 = -si * a + co * c, s11 = -si * b + co * d;
  S = MatrixND<2, real>(VectorND<2, real>((real)s00, (real)s10), VectorND<2, real>((real)s01, (real)s11));
}
inline void svd(const MatrixND<2, real> &m, MatrixND<2, real> &u, MatrixND<2, real> &sig, MatrixND<2, real> &v) {
  const double a = m[0][0], b = m[1][0], c = m[0][1], d = m[1][1];
  double x = a + d, y = c - b, den = std::sqrt(x * x + y * y), co = 1, si = 0;
  if (den > 0) { co = x / den; si = y / den; }
  const double s00 = co * a + si * c, s01 = co * b + si * d, s11 = -si * b + co * d;  // symmetric S
  double cv = 1, sv = 0, l0 = s00, l1 = s11;
  if (std::abs(s01) > 0) {
    const double tau = 0.5 * (s00 - s11), w = std::sqrt(tau * tau + s01 * s01);
    const double t = tau > 0 ? s01 / (tau + w) : s01 / (tau - w);
    cv = 1 / std::sqrt(t * t + 1); sv = -t * cv;
    // V = [[cv, sv], [-sv, cv]] (rows) diagonalises S
    l0 = cv * cv * s00 - 2 * cv * sv * s01 + sv * sv * s11;
    l1 = sv * sv * s00 + 2 * cv * sv * s01 + cv * cv * s11;
  }
  if (std::abs(l0) < std::abs(l1)) { std::swap(l0, l1); const double t = cv; cv = -sv; sv = t; }  // |s0| >= |s1|, V stays a rotation
  if (l0 < 0) { l0 = -l0; l1 = -l1; cv = -cv; sv = -sv; }
  // V (rows) = [[cv, sv], [-sv, cv]];  U = R V
  const double u00 = co * cv + si * sv, u01 = co * sv - si * cv, u10 = si * cv - co * sv, u11 = si * sv + co * cv;
  v = MatrixND<2, real>(VectorND<2, real>((real)cv, (real)-sv), VectorND<2, real>((real)sv, (real)cv));
  u = MatrixND<2, real>(VectorND<2, real>((real)u00, (real)u10), VectorND<2, real>((real)u01, (real)u11));
  sig = MatrixND<2, real>(VectorND<2, real>((real)l0, (real)l1));
}

// ------------------------------------------------------------------------------------------------ Config
class Config {
  std::map<std::string, std::string> data;
  template <typename T> static std::string to_s(const T &v) { std::ostringstream o; o.precision(9); o << v; return o.str(); }
  template <int dim, typename T> static std::string to_s(const VectorND<dim, T> &v) {
    std::ostringstream o; o.precision(9); o << "(";
    for (int i = 0; i < dim; i++) o << (i ? "," : "") << v[i];
    o << ")"; return o.str();
  }
  static std::string to_s(const std::string &v) { return v; }
  static std::string to_s(const char *v) { return v; }
  static std::string to_s(bool v) { return v ? "true" : "false"; }
  template <typename T> static void parse(const std::string &s, T &out) {
    if constexpr (std::is_same<T, bool>::value) {
      out = !(s == "false" || s == "False" || s == "0" || s == "");
    } else if constexpr (std::is_floating_point<T>::value) {
      out = (T)std::strtod(s.c_str(), nullptr);
    } else if constexpr (std::is_integral<T>::value) {
      if (s == "true" || s == "True") out = 1; else if (s == "false" || s == "False") out = 0;
      else out = (T)std::strtoll(s.c_str(), nullptr, 10);
    } else {
      std::istringstream i(s); i >> out;
    }
  }
  static void parse(const std::string &s, std::string &out) { out = s; }
  template <int dim, typename T> static void parse(const std::string &s, VectorND<dim, T> &out) {
    std::string t = s;
    for (auto &c : t) if (c == '(' || c == ')' || c == ',' || c == '[' || c == ']') c = ' ';
    std::istringstream i(t);
    for (int k = 0; k < dim; k++) { double x = 0; i >> x; out[k] = (T)x; }
  }

 public:
  bool has_key(const std::string &k) const { return data.count(k) > 0; }
  template <typename T> Config &set(const std::string &k, const T &v) { data[k] = to_s(v); return *this; }
  template <typename T> Config &set(const std::string &k, T *v) { data[k] = std::to_string((unsigned long long)(uintptr_t)v); return *this; }
  template <typename T> T get(const std::string &k) const {
    auto it = data.find(k);
    if (it == data.end()) shim_fail(("Config: missing key " + k).c_str(), __FILE__, __LINE__);
    T out{};
    parse(it->second, out);
    return out;
  }
  template <typename T> T get(const std::string &k, const T &dflt) const { return has_key(k) ? get<T>(k) : dflt; }
  std::string get(const std::string &k, const char *dflt) const { return has_key(k) ? get<std::string>(k) : std::string(dflt); }
  std::string get_string(const std::string &k) const { return get<std::string>(k); }
  template <typename T> T *get_ptr(const std::string &k) const { return (T *)(uintptr_t)get<unsigned long long>(k); }
  // "key=value;key=value"
  static Config from_string(const std::string &s) {
    Config c;
    size_t p = 0;
    while (p < s.size()) {
      size_t e = s.find(';', p);
      if (e == std::string::npos) e = s.size();
      const std::string kv = s.substr(p, e - p);
      const size_t q = kv.find('=');
      if (q != std::string::npos) c.data[kv.substr(0, q)] = kv.substr(q + 1);
      p = e + 1;
    }
    return c;
  }
};

// ------------------------------------------------------------------------------------------------ Unit + factory
class Unit {
 public:
  virtual void initialize(const Config &) {}
  virtual bool test() const { return true; }
  virtual std::string get_name() const { return "unit"; }
  virtual ~Unit() {}
  virtual void binary_io(BinaryOutputSerializer &) const {}
  virtual void binary_io(BinaryInputSerializer &) const {}
};
template <typename T>
struct InterfaceHolder {
  using Placement = std::function<T *(void *)>;
  using Make = std::function<std::unique_ptr<T>()>;
  static std::map<std::string, Placement> &placement() { static std::map<std::string, Placement> m; return m; }
  static std::map<std::string, Make> &make() { static std::map<std::string, Make> m; return m; }
};
template <typename T> T *create_instance_placement(const std::string &alias, void *where) {
  auto &m = InterfaceHolder<T>::placement();
  auto it = m.find(alias);
  if (it == m.end()) shim_fail(("no implementation registered under '" + alias + "'").c_str(), __FILE__, __LINE__);
  return it->second(where);
}
template <typename T> std::unique_ptr<T> create_instance_unique(const std::string &alias) {
  auto &m = InterfaceHolder<T>::make();
  auto it = m.find(alias);
  if (it == m.end()) shim_fail(("no implementation registered under '" + alias + "'").c_str(), __FILE__, __LINE__);
  return it->second();
}
template <typename T> std::unique_ptr<T> create_instance_unique(const std::string &alias, const Config &cfg) {
  auto p = create_instance_unique<T>(alias);
  p->initialize(cfg);
  return p;
}
#define TC_INTERFACE(T)
#define TC_INTERFACE_DEF(T, name)
#define TC_SHIM_CAT_(a, b) a##b
#define TC_SHIM_CAT(a, b) TC_SHIM_CAT_(a, b)
#define TC_IMPLEMENTATION(B, D, alias)                                                                  \
  static struct TC_SHIM_CAT(ShimRegister_##B##_##D##_, __LINE__) {                                      \
    TC_SHIM_CAT(ShimRegister_##B##_##D##_, __LINE__)() {                                                \
      ::taichi::InterfaceHolder<B>::placement()[alias] = [](void *p) -> B * { return new (p) D(); };    \
      ::taichi::InterfaceHolder<B>::make()[alias] = []() -> std::unique_ptr<B> { return std::make_unique<D>(); }; \
    }                                                                                                   \
  } TC_SHIM_CAT(shim_register_instance_##B##_##D##_, __LINE__);

// tests of the reference (TC_TEST bodies) are compiled as never-called templates
#define TC_TEST(name) template <typename TC_TEST_T_> static void TC_SHIM_CAT(shim_tc_test_, __LINE__)()
#define CHECK(x) ((void)(x))
#define TC_CHECK_EQUAL(a, b, tol) ((void)0)

// ------------------------------------------------------------------------------------------------ regions / arrays
template <int dim>
struct IndexND {
  using Vectori = VectorND<dim, int>;
  using Vector = VectorND<dim, real>;
  Vectori i, lo, hi;
  Vector storage_offset;
  IndexND() {}
  IndexND(const Vectori &lo, const Vectori &hi, const Vector &off) : i(lo), lo(lo), hi(hi), storage_offset(off) {}
  Vectori get_ipos() const { return i; }
  Vector get_pos() const { return i.template cast<real>() + storage_offset; }
  int &operator[](int k) { return i[k]; }
  int operator[](int k) const { return i[k]; }
  operator Vectori() const { return i; }
  IndexND operator+(const Vectori &o) const { IndexND r = *this; r.i = i + o; return r; }
  void next() {  // last axis fastest
    for (int k = dim - 1; k >= 0; k--) {
      if (++i[k] < hi[k]) return;
      if (k > 0) i[k] = lo[k];
    }
  }
  bool operator!=(const IndexND &o) const { return i != o.i; }
  IndexND &operator++() { next(); return *this; }
  IndexND &operator*() { return *this; }
  int x_() const { return i[0]; }
};
template <int dim>
struct RegionND {
  using Vectori = VectorND<dim, int>;
  using Vector = VectorND<dim, real>;
  Vectori lo, hi;
  Vector storage_offset;
  RegionND() {}
  RegionND(const Vectori &lo, const Vectori &hi, const Vector &off = Vector(0.5f)) : lo(lo), hi(hi), storage_offset(off) {}
  IndexND<dim> begin() const {
    for (int k = 0; k < dim; k++) if (hi[k] <= lo[k]) return end();
    return IndexND<dim>(lo, hi, storage_offset);
  }
  IndexND<dim> end() const { IndexND<dim> e(lo, hi, storage_offset); e.i = lo; e.i[0] = hi[0]; return e; }
};
template <int dim, typename T>
struct ArrayND {
  using Vectori = VectorND<dim, int>;
  Vectori res;
  std::vector<T> data;
  ArrayND() {}
  ArrayND(const Vectori &r, T init = T()) { initialize(r, init); }
  void initialize(const Vectori &r, T init = T()) { res = r; size_t n = 1; for (int k = 0; k < dim; k++) n *= (size_t)std::max(r[k], 0); data.assign(n, init); }
  const Vectori &get_res() const { return res; }
  size_t lin(const Vectori &i) const { size_t o = 0; for (int k = 0; k < dim; k++) o = o * res[k] + i[k]; return o; }
  T &operator[](const Vectori &i) { return data[lin(i)]; }
  const T &operator[](const Vectori &i) const { return data[lin(i)]; }
  T &operator[](const IndexND<dim> &i) { return data[lin(i.i)]; }
  const T &operator[](const IndexND<dim> &i) const { return data[lin(i.i)]; }
  bool inside(const Vectori &i) const { for (int k = 0; k < dim; k++) if (i[k] < 0 || i[k] >= res[k]) return false; return true; }
  RegionND<dim> get_region() const { return RegionND<dim>(Vectori(0), res); }
  void write_as_image(const std::string &) const {}
  void reset_zero() { std::fill(data.begin(), data.end(), T()); }
};
template <typename T> using Array2D = ArrayND<2, T>;
template <typename T> using Array3D = ArrayND<3, T>;

// ------------------------------------------------------------------------------------------------ inert neighbours
struct Texture { Vector4 sample(const Vector2 &) const { TC_NOT_IMPLEMENTED } Vector4 sample(const Vector3 &) const { TC_NOT_IMPLEMENTED } };
struct Mesh { std::vector<Vector3> vertices; void initialize(const Config &) { TC_NOT_IMPLEMENTED } };
struct AssetManager { template <typename T> static std::shared_ptr<T> get_asset(int) { TC_NOT_IMPLEMENTED } };
struct RenderParticle {};
namespace fmt {
inline void format_into(std::ostringstream &o, const std::string &f, size_t p) { o << f.substr(p); }
template <typename T, typename... A> inline void format_into(std::ostringstream &o, const std::string &f, size_t p, const T &v, const A &...rest) {
  const size_t q = f.find("{}", p);
  if (q == std::string::npos) { o << f.substr(p); return; }
  o << f.substr(p, q - p) << v;
  format_into(o, f, q + 2, rest...);
}
// "{}" placeholders filled in order (src/articulation.cpp:109 builds the keys offset0 / offset1 this way)
template <typename... A> inline std::string format(const std::string &f, const A &...a) { std::ostringstream o; format_into(o, f, 0, a...); return o.str(); }
template <typename... A> inline void print(FILE *, const char *, A &&...) {}
}  // namespace fmt

}  // namespace taichi
#include <taichi/dynamics/rigid_body_shim.h>  // RigidBody / mesh elements / rotations for the CPIC coupling (ours)
namespace taichi {

// analytic level set (taichi core: a SAMPLED signed-distance array built by add_plane / add_sphere / add_cuboid,
// and DynamicLevelSet = two such arrays at times t0 < t1 blended linearly in time — scripts/async/async_mpm.py:119-127
// builds one per frame from levelset_generator(t0), levelset_generator(t1)).  Here each key frame is a list of exact
// shapes; the time blending follows the same model: phi = lerp(phi0, phi1), gradient = normalised lerp of the two
// gradients, d phi / dt = (phi1 - phi0) / (t1 - t0)   [our reading of the absent core: a design choice, not a
// reference fact].  Arguments of sample() etc. are in GRID units like the reference's call sites
// (src/mpm.cpp:323-342,416-421); shapes are kept in world units; phi is returned in grid units.
struct ShimShape { int type = 0, inside_out = 0; float p[6] = {0, 0, 0, 0, 0, 0}; };
template <int dim>
struct LevelSet {
  using Vector = VectorND<dim, real>;
  real friction = 1.0f;
  std::vector<ShimShape> shapes;
  real delta_x = 1.0f;
  // phi (grid units) and unit gradient of the nearest shape at grid position pos
  real eval(const Vector &pos, Vector *grad) const {
    const real dx = delta_x, idx = 1.0f / dx;
    real best = 1e30f;
    Vector g(0.0f);
    for (const ShimShape &s : shapes) {
      real x[3] = {0, 0, 0};
      for (int k = 0; k < dim; k++) x[k] = pos[k] * dx;
      real ph, gr[3] = {0, 0, 0};
      if (s.type == 0) {
        ph = s.p[0] * x[0] + s.p[1] * x[1] + s.p[2] * x[2] + s.p[3];
        gr[0] = s.p[0]; gr[1] = s.p[1]; gr[2] = s.p[2];
      } else if (s.type == 1) {
        const real d0 = x[0] - s.p[0], d1 = x[1] - s.p[1], d2 = x[2] - s.p[2];
        const real len = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2), inv = len > 0 ? 1.0f / len : 0.0f;
        ph = len - s.p[3];
        gr[0] = d0 * inv; gr[1] = d1 * inv; gr[2] = d2 * inv;
      } else {
        bool inside = true;
        real near[3];
        for (int k = 0; k < 3; k++) { inside = inside && s.p[k] <= x[k] && x[k] <= s.p[3 + k]; near[k] = std::min(std::max(x[k], s.p[k]), s.p[3 + k]); }
        if (inside) {
          real b = 1e30f;
          for (int k = 0; k < 3; k++) {
            const real dlo = x[k] - s.p[k], dhi = s.p[3 + k] - x[k];
            if (dlo < b) { b = dlo; gr[0] = gr[1] = gr[2] = 0; gr[k] = -1; }
            if (dhi < b) { b = dhi; gr[0] = gr[1] = gr[2] = 0; gr[k] = 1; }
          }
          ph = -b;
        } else {
          const real d0 = x[0] - near[0], d1 = x[1] - near[1], d2 = x[2] - near[2];
          const real len = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
          ph = len; gr[0] = d0 / len; gr[1] = d1 / len; gr[2] = d2 / len;
        }
      }
      if (s.type != 0 && s.inside_out) { ph = -ph; for (int k = 0; k < 3; k++) gr[k] = -gr[k]; }
      ph *= idx;
      if (ph < best) { best = ph; for (int k = 0; k < dim; k++) g[k] = gr[k]; }
    }
    if (grad) *grad = g;
    return best;
  }
};
template <int dim>
struct DynamicLevelSet {
  using Vector = VectorND<dim, real>;
  std::shared_ptr<LevelSet<dim>> levelset0, levelset1;  // levelset1 == nullptr: static
  real t0 = 0.0f, t1 = 1.0f;
  void initialize(real t0_, real t1_, const std::shared_ptr<LevelSet<dim>> &l0, const std::shared_ptr<LevelSet<dim>> &l1) {
    t0 = t0_; t1 = t1_; levelset0 = l0; levelset1 = l1;
  }
  real sample(const Vector &pos, real t) const {
    const real p0 = levelset0->eval(pos, nullptr);
    if (!levelset1) return p0;
    const real a = (t - t0) / (t1 - t0);
    return (1.0f - a) * p0 + a * levelset1->eval(pos, nullptr);
  }
  Vector get_spatial_gradient(const Vector &pos, real t) const {
    Vector g0;
    levelset0->eval(pos, &g0);
    if (!levelset1) return g0;
    Vector g1;
    levelset1->eval(pos, &g1);
    const real a = (t - t0) / (t1 - t0);
    Vector g = g0 * (1.0f - a) + g1 * a;
    const real len = g.length();
    return len < 1e-10f ? Vector(0.0f) : g / len;
  }
  real get_temporal_derivative(const Vector &pos, real t) const {
    if (!levelset1) return 0.0f;
    return (levelset1->eval(pos, nullptr) - levelset0->eval(pos, nullptr)) / (t1 - t0);
  }
  bool inside(const Vector &) const { return true; }
};

// ------------------------------------------------------------------------------------------------ threads, timers
struct ShimRuntime {
  int threads = 1;  // OpenMP threads for every parallel construct of the reference
  std::map<std::string, double> seconds;
  std::mutex mu;
  static ShimRuntime &get() { static ShimRuntime r; return r; }
};
class ThreadedTaskManager {
 public:
  template <typename F> static void run(int n, int num_threads, const F &f) {
    int th = num_threads > 0 ? num_threads : ShimRuntime::get().threads;
    if (th <= 1) { for (int i = 0; i < n; i++) f(i); return; }
#pragma omp parallel for schedule(dynamic, 4) num_threads(th)
    for (int i = 0; i < n; i++) f(i);
  }
};
class Profiler {
  std::string name;
  std::chrono::steady_clock::time_point t0;
 public:
  explicit Profiler(const std::string &n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~Profiler() {
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto &r = ShimRuntime::get();
    std::lock_guard<std::mutex> g(r.mu);
    r.seconds[name] += s;
  }
  static void disable() {}
  static void enable() {}
};
#define TC_PROFILER(name) ::taichi::Profiler TC_SHIM_CAT(shim_profiler_scope_, __LINE__)(name);
#define TC_PROFILE(name, stmt) { ::taichi::Profiler shim_profiler_(name); stmt; }
#define TC_PROFILE_TPE(name, stmt, n) { ::taichi::Profiler shim_profiler_(name); stmt; }
namespace Time {
struct Timer { explicit Timer(const std::string &) {} };
}  // namespace Time

// ------------------------------------------------------------------------------------------------ Simulation base
template <int DIM>
class Simulation : public Unit {
 public:
  static constexpr int dim = DIM;  // (visible in the explicit specialisations of the derived class, src/mpm.cpp:683)
  using Vector = VectorND<dim, real>;
  using VectorP = VectorND<dim + 1, real>;
  using VectorI = VectorND<dim, int>;
  using Vectori = VectorND<dim, int>;
  using Matrix = MatrixND<dim, real>;
  using MatrixP = MatrixND<dim + 1, real>;
  real current_t = 0.0f;
  int num_threads = 1;
  int frame = 0;
  DynamicLevelSet<dim> levelset;
  void initialize(const Config &config) override { num_threads = config.get("num_threads", 1); }
  virtual std::string add_particles(const Config &) { return ""; }
  virtual void step(real) {}
  virtual std::vector<RenderParticle> get_render_particles() const { return {}; }
  virtual void visualize() const {}
  virtual std::string get_debug_information() { return ""; }
  virtual std::string general_action(const Config &) { return ""; }
  virtual void set_levelset(const DynamicLevelSet<dim> &l) { levelset = l; }
  real get_current_time() const { return current_t; }
  template <typename S> void io(S &) const {}
};
using Simulation2D = Simulation<2>;
using Simulation3D = Simulation<3>;

}  // namespace taichi

namespace tbb {
// the containers src/async/async_mpm.{h,cpp} uses.  The reference runs its block loops on TBB threads; here every tbb
// construct with a blocked_range runs serially (AsyncMPM is a checker in this build, not a timed path), so plain vectors do.
template <typename T>
class concurrent_vector : public std::vector<T> {
 public:
  using std::vector<T>::vector;
};
template <typename T>
struct blocked_range {
  T b, e;
  blocked_range(T b_, T e_) : b(b_), e(e_) {}
  T begin() const { return b; }
  T end() const { return e; }
};
template <typename T>
class enumerable_thread_specific {
  std::vector<T> v = std::vector<T>(1);
 public:
  T &local() { return v[0]; }
  typename std::vector<T>::iterator begin() { return v.begin(); }
  typename std::vector<T>::iterator end() { return v.end(); }
};
template <typename T, typename F> inline void parallel_for(const blocked_range<T> &r, const F &f) { f(r); }
template <typename F> inline void parallel_for(int b, int e, const F &f) {
  const int th = ::taichi::ShimRuntime::get().threads;
  if (th <= 1) { for (int i = b; i < e; i++) f(i); return; }
#pragma omp parallel for schedule(static) num_threads(th)
  for (int i = b; i < e; i++) f(i);
}
template <typename It> inline void parallel_sort(It b, It e) {
  const int th = ::taichi::ShimRuntime::get().threads;
  if (th <= 1) { std::sort(b, e); return; }
  omp_set_num_threads(th);
  __gnu_parallel::sort(b, e);
}
}  // namespace tbb
