// taichi_mpm_amd/csrc/mpm_math.h — device-side 3x3 math and constitutive models (gfx950, VALU only).
//
// The 3x3 tensor work of MLS-MPM stays in vector registers (no MFMA: nothing here is GEMM-shaped).
// Design note: every constitutive model of the reference (src/particles.cpp) is isotropic, so both
// `calculate_force()` (= -V0 * P F^T, a function of the LEFT stretch only) and the return-mapping
// part of `plasticity()` (F <- U f(S) V^T = U diag(f(s_i)/s_i) U^T F) can be written with U and the
// singular values alone.  U and s_i^2 are the eigen-pairs of the symmetric matrix F F^T, so the
// device never forms V: one cyclic-Jacobi eigen-solve of a symmetric 3x3 per call instead of a full
// SVD.  This is mathematically identical to the reference's svd()/polar_decomp() route
// (src/particles.cpp:76,212,227,394,630,642) and is checked against the CPU oracle in tests/.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mpmhip.h"

namespace mpm {

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32): the IEEE-exact expansions hipcc emits for
// `/` and sqrtf cost ~10 VALU instructions each, and there are ~20 of them per particle in the eigen-solve.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// packed fp32: two lanes of a 64-bit register pair per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float a) { return (f2){a, a}; }

struct mat3 {
  float m[9];  // row-major
  __device__ __forceinline__ float &operator()(int r, int c) { return m[3 * r + c]; }
  __device__ __forceinline__ float operator()(int r, int c) const { return m[3 * r + c]; }
};

__device__ __forceinline__ mat3 mat_identity() {
  mat3 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.m[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  return r;
}
__device__ __forceinline__ mat3 mat_mul(const mat3 &A, const mat3 &B) {
  mat3 C;  // row r of C = sum_k A(r,k) * row k of B: the (c0, c1) pair of every row goes through packed fp32
  const f2 b0 = {B(0, 0), B(0, 1)}, b1 = {B(1, 0), B(1, 1)}, b2 = {B(2, 0), B(2, 1)};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const f2 c01 = fma2(splat2(A(r, 0)), b0, fma2(splat2(A(r, 1)), b1, splat2(A(r, 2)) * b2));
    C(r, 0) = c01.x; C(r, 1) = c01.y;
    C(r, 2) = fmaf(A(r, 0), B(0, 2), fmaf(A(r, 1), B(1, 2), A(r, 2) * B(2, 2)));
  }
  return C;
}
// A * B^T
__device__ __forceinline__ mat3 mat_mul_bt(const mat3 &A, const mat3 &B) {
  mat3 C;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) C(r, c) = fmaf(A(r, 0), B(c, 0), fmaf(A(r, 1), B(c, 1), A(r, 2) * B(c, 2)));
  return C;
}
__device__ __forceinline__ float mat_det(const mat3 &m) {
  return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
         m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}

// One Jacobi rotation annihilating a_pq of the symmetric matrix {app,aqq,arr,apq,arp,arq}; r is the third index.
// Columns p,q of U are rotated along (U is kept by columns: (U0k, U1k) as a register pair + U2k, so that the three rows
// of a rotation go through packed fp32).  With d = aqq - app, x = 2 a_pq, h = sqrt(d^2 + x^2) and den = |d| + h the
// classical small-angle root is t = tan(phi) = sgn(d) x / den, hence
//     (cos, sin) = (den, sgn(d) x) / sqrt(den^2 + x^2),        {app', aqq'} = (app + aqq -+ sgn(d) h) / 2
// (the rotated diagonal is the eigenvalue pair of the 2x2 block): one v_sqrt and one v_rsq per rotation, no division.
// d = x = 0 (to within 1e-15 of nothing) leaves everything unchanged.
__device__ __forceinline__ void jacobi_rotate(float &app, float &aqq, float &apq, float &arp, float &arq, f2 &up01,
                                              float &up2, f2 &uq01, float &uq2) {
  const float d = aqq - app, x = apq + apq;
  const float xx = x * x;
  const float h = fast_sqrt(fmaf(d, d, xx));
  const float den = fabsf(d) + h;
  const float n2 = fmaf(den, den, xx);
  const float r = __builtin_amdgcn_rsqf(n2);
  const bool live = n2 > 1e-30f;  // (below: d = x = 0 up to denormals, which v_rsq_f32 does not take — nothing to rotate)
  const uint32_t sd = __float_as_uint(d) & 0x80000000u;  // sgn(d), +1 at d = 0
  const float c = live ? den * r : 1.0f;
  const float sn = live ? __uint_as_float(__float_as_uint(x * r) ^ sd) : 0.0f;
  const float sum = app + aqq, hs = __uint_as_float(__float_as_uint(h) | sd);
  app = 0.5f * (sum - hs);
  aqq = 0.5f * (sum + hs);
  apq = 0.0f;
  const float nrp = c * arp - sn * arq, nrq = sn * arp + c * arq;
  arp = nrp; arq = nrq;
  const f2 c2 = splat2(c), s2 = splat2(sn);
  const f2 np01 = fma2(c2, up01, -(s2 * uq01));
  uq01 = fma2(s2, up01, c2 * uq01);
  up01 = np01;
  const float np2 = c * up2 - sn * uq2;
  uq2 = sn * up2 + c * uq2;
  up2 = np2;
}

// Ill-conditioned F: the decomposition is finished on F itself, one-sided (Hestenes), preconditioned by the U the eigen-solve of
// F F^T found (see sym_eig3_FFt): B = U^T F has nearly orthogonal rows sigma_k v_k^T; two cyclic sweeps of plane rotations make
// them orthogonal, U's columns rotate along, lam_k = |b_k|^2.  One-sided Jacobi loses nothing of what its input holds (Demmel &
// Veselic 1992), and B = U^T F formed in fp32 holds sigma_min to ~ eps cond(F) — the entries of F are of size sigma_max —
// instead of the eps cond(F)^2 of sqrt(eig(F F^T)); measured relative error of sigma_min: 2e-6 at cond 1e2, 1e-5 at 1e3,
// 2e-4 at 1e4, against 8e-4 / 4e-2 / 5.0 (profiles/r05_c_illcond_head.txt, r05_e_illcond_default.txt).
// (Re-measuring |F^T u_k| with the eigenvectors as they are — no rotations — was built first and is not enough: the eigenvectors
// of the two small singular values mix by ~ eps cond^2 / gap, and the larger of the two then leaks into the smaller.)
// Register budget: k_g2p keeps three workgroups per CU only below 168 VGPRs, and U, B and F in registers on top of a particle's
// other state cost it 4..14 more than it has (k_g2p_packed: 163 -> 175).  With `lds` (20 floats of LDS private to the lane: its
// row of the wave's store-staging slab, idle while a particle is computed) U and B live THERE while the rotations run: a
// rotation holds two rows, two columns and its own scalars, F is formed again as U B at the end (its rounding, eps |F|, is what
// F_new = R F carries anyway) — the rare path then needs fewer registers than the common one.  Without `lds` (k_affine, the
// colour-aware kernels, the debug entry points): the same arithmetic in registers.
__device__ __forceinline__ void plane_rotation(const float app, const float aqq, const float apq, float &c, float &sn) {
  // (cos, sin) that annihilate the off-diagonal of the 2x2 Gram block {app, apq; apq, aqq}: the small-angle root of jacobi_rotate
  const float d = aqq - app, x = apq + apq;
  const float xx = x * x;
  const float h = fast_sqrt(fmaf(d, d, xx));
  const float den = fabsf(d) + h;
  const float n2 = fmaf(den, den, xx);
  const float r = __builtin_amdgcn_rsqf(n2);
  const bool live = n2 > 1e-30f;
  const uint32_t sd = __float_as_uint(d) & 0x80000000u;
  c = live ? den * r : 1.0f;
  sn = live ? __uint_as_float(__float_as_uint(x * r) ^ sd) : 0.0f;
}
// (the empty asm statements end the compiler's scheduling regions: left alone it hoists the LDS loads of all six rotations to the
// front and the kernel's allocation grows by 50 registers — __launch_bounds__(256, 2) only tells it that 256 are free)
#define MPM_SCHED_FENCE() asm volatile("" ::: "memory")
template <bool IN_LDS>
__device__ __forceinline__ void sym_eig3_refine(mat3 &F, mat3 &U, float lam[3], float *lds) {
  if constexpr (IN_LDS) {
    // lds[3 k + r] = U(r, k) (columns), lds[9 + 3 k + c] = B(k, c) (rows)
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int c = 0; c < 3; c++) lds[9 + 3 * k + c] = fmaf(U(0, k), F(0, c), fmaf(U(1, k), F(1, c), U(2, k) * F(2, c)));
#pragma unroll
      for (int r = 0; r < 3; r++) lds[3 * k + r] = U(r, k);
    }
    MPM_SCHED_FENCE();
#pragma unroll
    for (int sweep = 0; sweep < 2; sweep++)
#pragma unroll
      for (int pair = 0; pair < 3; pair++) {
        const int p = pair == 2 ? 1 : 0, q = pair == 0 ? 1 : 2;
        float bp[3], bq[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { bp[k] = lds[9 + 3 * p + k]; bq[k] = lds[9 + 3 * q + k]; }
        float c, sn;
        plane_rotation(fmaf(bp[0], bp[0], fmaf(bp[1], bp[1], bp[2] * bp[2])), fmaf(bq[0], bq[0], fmaf(bq[1], bq[1], bq[2] * bq[2])),
                       fmaf(bp[0], bq[0], fmaf(bp[1], bq[1], bp[2] * bq[2])), c, sn);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          lds[9 + 3 * p + k] = c * bp[k] - sn * bq[k];
          lds[9 + 3 * q + k] = sn * bp[k] + c * bq[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float up = lds[3 * p + k], uq = lds[3 * q + k];
          lds[3 * p + k] = c * up - sn * uq;
          lds[3 * q + k] = sn * up + c * uq;
        }
        MPM_SCHED_FENCE();
      }
    float B[9];
#pragma unroll
    for (int k = 0; k < 9; k++) B[k] = lds[9 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) lam[k] = fmaf(B[3 * k], B[3 * k], fmaf(B[3 * k + 1], B[3 * k + 1], B[3 * k + 2] * B[3 * k + 2]));
#pragma unroll
    for (int r = 0; r < 3; r++) {  // F = U B, row by row
      const float u0 = lds[r], u1 = lds[3 + r], u2 = lds[6 + r];
#pragma unroll
      for (int c = 0; c < 3; c++) F(r, c) = fmaf(u0, B[c], fmaf(u1, B[3 + c], u2 * B[6 + c]));
    }
    MPM_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int r = 0; r < 3; r++) U(r, k) = lds[3 * k + r];
    MPM_SCHED_FENCE();
    return;
  }
  float b[3][3];  // B = U^T F
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) b[k][c] = fmaf(U(0, k), F(0, c), fmaf(U(1, k), F(1, c), U(2, k) * F(2, c)));
#pragma unroll
  for (int sweep = 0; sweep < 2; sweep++)
#pragma unroll
    for (int pair = 0; pair < 3; pair++) {
      const int p = pair == 2 ? 1 : 0, q = pair == 0 ? 1 : 2;
      float c, sn;
      plane_rotation(fmaf(b[p][0], b[p][0], fmaf(b[p][1], b[p][1], b[p][2] * b[p][2])),
                     fmaf(b[q][0], b[q][0], fmaf(b[q][1], b[q][1], b[q][2] * b[q][2])),
                     fmaf(b[p][0], b[q][0], fmaf(b[p][1], b[q][1], b[p][2] * b[q][2])), c, sn);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float np = c * b[p][k] - sn * b[q][k], nq = sn * b[p][k] + c * b[q][k];
        b[p][k] = np; b[q][k] = nq;
        const float up = c * U(k, p) - sn * U(k, q), uq = sn * U(k, p) + c * U(k, q);
        U(k, p) = up; U(k, q) = uq;
      }
    }
#pragma unroll
  for (int k = 0; k < 3; k++) lam[k] = fmaf(b[k][0], b[k][0], fmaf(b[k][1], b[k][1], b[k][2] * b[k][2]));
}

constexpr int kJacobiSweeps = 4;
#ifndef MPM_JACOBI_TOL
#define MPM_JACOBI_TOL 1e-7f
#endif

// Eigen-decomposition of the symmetric positive semi-definite A = F F^T:  A = U diag(lam) U^T.
// U is a proper rotation (product of Givens rotations).  Unsorted.  Cyclic Jacobi converges quadratically; the
// wavefront stops as soon as every lane's off-diagonal is below 1e-7 * trace (all users are isotropic functions
// U f(lam) U^T, whose error is ~f' * |off-diagonal|, also for near-equal eigenvalues).  The test runs before every
// sweep: material at rest or in free fall (F F^T diagonal to rounding) costs no sweep at all.
// (F is const in effect: the refinement of an ill-conditioned F may rebuild it from its factors, equal to rounding.)
// lds: see sym_eig3_refine.
template <bool IN_LDS = false>
__device__ __forceinline__ void sym_eig3_FFt(mat3 &F, mat3 &U, float lam[3], float *lds = nullptr) {
  float a00 = fmaf(F(0, 0), F(0, 0), fmaf(F(0, 1), F(0, 1), F(0, 2) * F(0, 2)));
  float a11 = fmaf(F(1, 0), F(1, 0), fmaf(F(1, 1), F(1, 1), F(1, 2) * F(1, 2)));
  float a22 = fmaf(F(2, 0), F(2, 0), fmaf(F(2, 1), F(2, 1), F(2, 2) * F(2, 2)));
  float a01 = fmaf(F(0, 0), F(1, 0), fmaf(F(0, 1), F(1, 1), F(0, 2) * F(1, 2)));
  float a02 = fmaf(F(0, 0), F(2, 0), fmaf(F(0, 1), F(2, 1), F(0, 2) * F(2, 2)));
  float a12 = fmaf(F(1, 0), F(2, 0), fmaf(F(1, 1), F(2, 1), F(1, 2) * F(2, 2)));
  f2 u0 = {1.0f, 0.0f}, u1 = {0.0f, 1.0f}, u2 = {0.0f, 0.0f};  // columns of U: rows 0,1
  float w0 = 0.0f, w1 = 0.0f, w2 = 1.0f;                        //               row 2
  const float tol = MPM_JACOBI_TOL * (a00 + a11 + a22);
#pragma unroll
  for (int sweep = 0; sweep < kJacobiSweeps; sweep++) {
    // the wave leaves when every lane has converged; a lane that HAS converged sits the sweep out, so that a particle's result
    // does not depend on which other particles share its wave (the per-block and the packed G2P walk, two tilings of one scene)
    const bool more = fmaxf(fabsf(a01), fmaxf(fabsf(a02), fabsf(a12))) > tol;
    if (!__any(more)) break;
    if (more) {
      jacobi_rotate(a00, a11, a01, a02, a12, u0, w0, u1, w1);  // (p,q,r)=(0,1,2)
      jacobi_rotate(a00, a22, a02, a01, a12, u0, w0, u2, w2);  // (0,2,1)
      jacobi_rotate(a11, a22, a12, a01, a02, u1, w1, u2, w2);  // (1,2,0)
    }
  }
  U(0, 0) = u0.x; U(1, 0) = u0.y; U(2, 0) = w0;
  U(0, 1) = u1.x; U(1, 1) = u1.y; U(2, 1) = w1;
  U(0, 2) = u2.x; U(1, 2) = u2.y; U(2, 2) = w2;
  lam[0] = a00; lam[1] = a11; lam[2] = a22;
  // Ill-conditioned F.  The eigenvalues of F F^T carry an ABSOLUTE error ~ eps sigma_max^2, i.e. sigma_min loses relative accuracy
  // like eps cond(F)^2 (8e-4 at cond 1e2, nothing left at 1e3: profiles/r05_c_illcond_head.txt), while the reference takes
  // svd(F) — and the Hencky models take log(sigma).  A lane that holds lam_min < lam_max / 64 (cond > 8: never on the benchmark
  // states) finishes the decomposition on F itself (sym_eig3_refine) — decided per lane, like the sweeps above: until round 5 the
  // whole wave followed one such lane, and results depended on who shared a wave.
  const float lmax = fmaxf(a00, fmaxf(a11, a22)), lmin = fminf(a00, fminf(a11, a22));
  if (lmin < (1.0f / 64.0f) * lmax) sym_eig3_refine<IN_LDS>(F, U, lam, lds);
}

// Signed singular values in U's (unsorted) column order: s_i = sqrt(lam_i); if det F < 0 the sign goes
// on the smallest one (the oracle's / taichi's convention: U,V rotations, sign on the last sigma).
__device__ __forceinline__ void signed_sigma(const float lam[3], float detF, float s[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++) s[i] = fast_sqrt(fmaxf(lam[i], 0.0f));
  if (detF < 0.0f) {
    int k = (s[0] <= s[1]) ? ((s[0] <= s[2]) ? 0 : 2) : ((s[1] <= s[2]) ? 1 : 2);
    if (k == 0) s[0] = -s[0];
    else if (k == 1) s[1] = -s[1];
    else s[2] = -s[2];
  }
}

// U diag(d) U^T
__device__ __forceinline__ mat3 sandwich(const mat3 &U, const float d[3]) {
  mat3 R;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = r; c < 3; c++) {
      float v = fmaf(U(r, 0) * d[0], U(c, 0), fmaf(U(r, 1) * d[1], U(c, 1), U(r, 2) * d[2] * U(c, 2)));
      R(r, c) = v;
      R(c, r) = v;
    }
  return R;
}
// (U diag(d1) U^T, U diag(d2) U^T) together: the dyads U(r,k) U(c,k) are formed once and each entry pair goes through
// packed fp32
__device__ __forceinline__ void sandwich2(const mat3 &U, const float d1[3], const float d2[3], mat3 &R1, mat3 &R2) {
  const f2 e0 = {d1[0], d2[0]}, e1 = {d1[1], d2[1]}, e2 = {d1[2], d2[2]};
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = r; c < 3; c++) {
      const f2 v = fma2(splat2(U(r, 0) * U(c, 0)), e0, fma2(splat2(U(r, 1) * U(c, 1)), e1, splat2(U(r, 2) * U(c, 2)) * e2));
      R1(r, c) = v.x; R1(c, r) = v.x;
      R2(r, c) = v.y; R2(c, r) = v.y;
    }
}

struct GroupParams {
  float p[MPMHIP_NPARAM];
  int32_t type;
  int32_t pad[3];
};

// calculate_force(): returns -vol * P(F) * F^T  (src/particles.h:134-137; bodies in src/particles.cpp)
// MATS_: the material set the caller is compiled for (MAT_ALL below); a set of ONE material needs no type dispatch
template <uint32_t MATS_ = 0x1FEu>
__device__ __forceinline__ mat3 calculate_force(const GroupParams &g, const mat3 &F, float aux) {
  const float vol = g.p[1];
  mat3 out;
  const int type = ((MATS_ & (MATS_ - 1u)) == 0u) ? (int)__builtin_ctz(MATS_) : g.type;
  switch (type) {
    case MPMHIP_VISCO:   // src/particles.cpp:72-85 (same fixed-corotated energy)
    case MPMHIP_JELLY:   // src/particles.cpp:391-411  P = 2mu(F-R) + lambda (J-1) J F^-T
    case MPMHIP_SNOW: {  // src/particles.cpp:207-220, 244-252 (mu,lambda scaled by exp(h(1-Jp)))
      float mu = g.p[2], la = g.p[3];
      if (type == MPMHIP_SNOW) {
        const float e = expf(g.p[4] * (1.0f - aux));
        mu *= e; la *= e;
      }
      mat3 U, Fm = F; float lam[3], s[3];
      sym_eig3_FFt(Fm, U, lam);
      const float J = mat_det(F);
      signed_sigma(lam, J, s);
      // (F-R)F^T = U (S^2 - S) U^T ;  lambda (J-1) J F^-T F^T = lambda (J-1) J I
      const float vol_l = la * (J - 1.0f) * J;
      float d[3];
#pragma unroll
      for (int i = 0; i < 3; i++) d[i] = -vol * fmaf(2.0f * mu, lam[i] - s[i], vol_l);
      out = sandwich(U, d);
      break;
    }
    case MPMHIP_LINEAR: {  // src/particles.cpp:329-336
      const float mu = g.p[2], la = g.p[3];
      const float tr = la * (F(0, 0) + F(1, 1) + F(2, 2) - 3.0f);
      mat3 P;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) P(r, c) = mu * (F(r, c) + F(c, r) - ((r == c) ? 2.0f : 0.0f)) + ((r == c) ? tr : 0.0f);
      out = mat_mul_bt(P, F);
#pragma unroll
      for (int i = 0; i < 9; i++) out.m[i] *= -vol;
      break;
    }
    case MPMHIP_WATER: {  // src/particles.cpp:463-467: -vol * j * (-p I), p = k (j^-gamma - 1)
      const float j = aux;
      const float p = g.p[2] * (powf(j, -g.p[3]) - 1.0f);
      const float dd = vol * j * p;
#pragma unroll
      for (int i = 0; i < 9; i++) out.m[i] = (i % 4 == 0) ? dd : 0.0f;
      break;
    }
    case MPMHIP_SAND:       // src/particles.cpp:628-637
    case MPMHIP_VON_MISES:  // :701-711
    case MPMHIP_ELASTIC: {  // :798-807   P F^T = U (2 mu ln S + lambda tr(ln S) I) U^T
      const float mu = g.p[2], la = g.p[3];
      mat3 U, Fm = F; float lam[3], s[3];
      sym_eig3_FFt(Fm, U, lam);
      signed_sigma(lam, mat_det(F), s);
      float ls[3];
#pragma unroll
      for (int i = 0; i < 3; i++) ls[i] = logf(s[i]);  // log of a negative sigma is NaN, as in the reference (:631)
      const float tr = ls[0] + ls[1] + ls[2];
      float d[3];
#pragma unroll
      for (int i = 0; i < 3; i++) d[i] = -vol * fmaf(2.0f * mu, ls[i], la * tr);
      out = sandwich(U, d);
      break;
    }
    default:
#pragma unroll
      for (int i = 0; i < 9; i++) out.m[i] = 0.0f;
  }
  return out;
}

// ViscoParticle::plasticity (src/particles.cpp:87-134): F_hat = approximate_exponent(cdg - I) F; the Frobenius
// norm of the first Piola-Kirchhoff stress of the OLD F drives gamma; sigma_d <- clamp(sigma_d / (sigma_d *
// det^-1/3)^gamma, 0.1, 10); tau += kappa gamma |P|.  Both SVDs of the reference share U and V, so one
// eigen-solve of F_hat F_hat^T gives everything (plus one of the old F for |P|, which is rotation invariant).
// On return F = F_hat, U/s = its left singular vectors/values, sn = the new singular values.
template <bool IN_LDS = false>
__device__ __forceinline__ void visco_return(const GroupParams &g, const mat3 &cdg, mat3 &F, float &aux, mat3 &U,
                                             float s[3], float sn[3], float *lds = nullptr) {
  const float mu = g.p[2], la = g.p[3], vnu = g.p[4], kappa = g.p[5], dt = g.p[6];
  float pnorm;
  {
    mat3 U0; float lam0[3], s0[3];
    sym_eig3_FFt<IN_LDS>(F, U0, lam0, lds);
    const float J0 = mat_det(F);
    signed_sigma(lam0, J0, s0);
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const float pd = fmaf(2.0f * mu, s0[d] - 1.0f, la * (J0 - 1.0f) * J0 / s0[d]);
      acc = fmaf(pd, pd, acc);
    }
    pnorm = sqrtf(acc);
  }
  mat3 sm;  // approximate_exponent(dt, (cdg - I)/dt): r = I + s + s^2/2 with s halved until det r > 0, then squared back
#pragma unroll
  for (int i = 0; i < 9; i++) sm.m[i] = cdg.m[i] - ((i % 4 == 0) ? 1.0f : 0.0f);
  mat3 r;
  int halvings = 0;
  for (;;) {
    mat3 h = sm;
#pragma unroll
    for (int i = 0; i < 9; i++) h.m[i] = 0.5f * sm.m[i] + ((i % 4 == 0) ? 1.0f : 0.0f);
    r = mat_mul(h, sm);
#pragma unroll
    for (int i = 0; i < 3; i++) r.m[4 * i] += 1.0f;
    if (mat_det(r) > 0.0f || halvings > 20) break;
#pragma unroll
    for (int i = 0; i < 9; i++) sm.m[i] *= 0.5f;
    halvings++;
  }
  for (int i = 0; i < halvings; i++) r = mat_mul(r, r);
  F = mat_mul(r, F);
  float lam[3];
  sym_eig3_FFt<IN_LDS>(F, U, lam, lds);
  signed_sigma(lam, mat_det(F), s);
  float gamma = 0.0f;
  if (pnorm > 1e-5f) gamma = fminf(fmaxf(dt * vnu * (pnorm - aux) / pnorm, 0.0f), 1.0f);
  const float dets = s[0] * s[1] * s[2];
  const float scale = fabsf(dets) > 1e-5f ? 1.0f / powf(dets, 1.0f / 3.0f) : 1.0f;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const float md = powf(s[d] * scale, gamma);
    const float inv = fabsf(md) > 1e-5f ? 1.0f / md : 1.0f;
    sn[d] = fminf(fmaxf(s[d] * inv, 0.1f), 10.0f);
  }
  aux = fmaf(kappa * gamma, pnorm, aux);
}

// plasticity(cdg): F <- cdg F, then the material's return mapping (src/particles.h:139-141).
__device__ __forceinline__ void plasticity(const GroupParams &g, const mat3 &cdg, mat3 &F, float &aux) {
  if (g.type == MPMHIP_WATER) {  // src/particles.cpp:469-478 (dg_e is never touched for water)
    float j = aux * (cdg(0, 0) + cdg(1, 1) + cdg(2, 2) - 2.0f);
    aux = (j < 0.1f) ? 0.1f : j;
    return;
  }
  if (g.type == MPMHIP_VISCO) {
    mat3 U; float s[3], sn[3], ratio[3];
    visco_return(g, cdg, F, aux, U, s, sn);
#pragma unroll
    for (int i = 0; i < 3; i++) ratio[i] = sn[i] / s[i];
    F = mat_mul(sandwich(U, ratio), F);
    return;
  }
  F = mat_mul(cdg, F);
  if (g.type == MPMHIP_JELLY || g.type == MPMHIP_LINEAR || g.type == MPMHIP_ELASTIC) return;  // :413-416,:338-341,:809-812
  mat3 U; float lam[3], s[3];
  sym_eig3_FFt(F, U, lam);
  signed_sigma(lam, mat_det(F), s);
  float ratio[3];  // f(s_i) / s_i
  if (g.type == MPMHIP_SNOW) {  // src/particles.cpp:222-242
    const float lo = 1.0f - g.p[5], hi = 1.0f + g.p[6];
    float det_o = 1.0f, det_n = 1.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float c = fminf(fmaxf(s[i], lo), hi);
      det_o *= s[i];
      det_n *= c;
      ratio[i] = c / s[i];
    }
    float Jp = aux * det_o / det_n;
    if (!(Jp <= g.p[8])) Jp = g.p[8];
    if (!(Jp >= g.p[7])) Jp = g.p[7];
    aux = Jp;
  } else if (g.type == MPMHIP_SAND) {  // src/particles.cpp:599-626, 639-647
    const float mu = g.p[2], la = g.p[3], alpha = g.p[4], coh = g.p[5], beta = g.p[6];
    float eps[3];
#pragma unroll
    for (int i = 0; i < 3; i++) eps[i] = logf(fmaxf(fabsf(s[i]), 1e-4f)) - coh;
    const float sum = eps[0] + eps[1] + eps[2];
    const float tr = sum + aux;
    float eh[3];
#pragma unroll
    for (int i = 0; i < 3; i++) eh[i] = eps[i] - tr * (1.0f / 3.0f);
    const float ehn = sqrtf(fmaf(eh[0], eh[0], fmaf(eh[1], eh[1], eh[2] * eh[2])));
    float ns[3];
    if (tr >= 0.0f) {
      const float e = expf(coh);
      ns[0] = e; ns[1] = e; ns[2] = e;
      aux = fmaf(beta, sum, aux);
    } else {
      aux = 0.0f;
      const float dg = ehn + (3.0f * la + 2.0f * mu) / (2.0f * mu) * tr * alpha;
      const float k = (dg <= 0.0f) ? 0.0f : dg / ehn;
#pragma unroll
      for (int i = 0; i < 3; i++) ns[i] = expf(eps[i] - k * eh[i] + coh);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) ratio[i] = ns[i] / s[i];
  } else if (g.type == MPMHIP_VON_MISES) {  // src/particles.cpp:713-732
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; i++) e[i] = logf(s[i]);
    const float tr = e[0] + e[1] + e[2];
    float eh[3] = {e[0] - tr * (1.0f / 3.0f), e[1] - tr * (1.0f / 3.0f), e[2] - tr * (1.0f / 3.0f)};
    const float n2 = fmaf(eh[0], eh[0], fmaf(eh[1], eh[1], eh[2] * eh[2]));  // frobenius_norm2 (squared, as in the reference)
    const float dg = n2 - g.p[4] / (2.0f * g.p[2]);
    if (dg <= 0.0f) return;
#pragma unroll
    for (int i = 0; i < 3; i++) ratio[i] = expf(e[i] - (dg / n2) * eh[i]) / s[i];
  } else {
    return;
  }
  // F <- U diag(ratio) U^T F
  F = mat_mul(sandwich(U, ratio), F);
}

// Fused G2P tail: plasticity(cdg) of THIS substep followed by calculate_force() of the NEXT substep's P2G
// (src/particles.h:134-141).  The reference evaluates them in two separate passes with an svd/polar_decomp
// each; both are functions of the same U and singular values (F_new = U S' V^T keeps U), so the device does
// ONE symmetric eigen-solve per particle per substep and derives both from it.  `stress` = -vol P(F_new) F_new^T.
//
// MATS: bit t set = material type t can occur (compile time).  k_g2p is instantiated for the full set and for each single
// material: a scene made of one material (the benchmark configurations, most scene scripts) runs a kernel that carries only
// that material's code — no type dispatch, a third of the instructions, and a register budget set by the material in use
// instead of by the hungriest one of the eight (visco: two eigen-solves and a matrix exponential).
constexpr uint32_t MAT_ALL = 0x1FEu;  // MPMHIP_VISCO (1) .. MPMHIP_ELASTIC (8)
template <uint32_t MATS>
__device__ __forceinline__ bool mat_is(const GroupParams &g, int t) {
  if (!((MATS >> t) & 1u)) return false;       // not in this kernel's set
  if ((MATS & (MATS - 1u)) == 0u) return true;  // the only one in the set: no run-time test
  return g.type == t;
}
template <uint32_t MATS = MAT_ALL, bool IN_LDS = false>
__device__ __forceinline__ void plasticity_and_force(const GroupParams &g, const mat3 &cdg, mat3 &F, float &aux,
                                                     mat3 &stress, float *lds = nullptr) {
  const float vol = g.p[1];
  if (mat_is<MATS>(g, MPMHIP_WATER)) {  // src/particles.cpp:463-478
    float j = aux * (cdg(0, 0) + cdg(1, 1) + cdg(2, 2) - 2.0f);
    j = (j < 0.1f) ? 0.1f : j;
    aux = j;
    const float p = g.p[2] * (powf(j, -g.p[3]) - 1.0f);
    const float dd = vol * j * p;
#pragma unroll
    for (int i = 0; i < 9; i++) stress.m[i] = (i % 4 == 0) ? dd : 0.0f;
    return;
  }
  if (mat_is<MATS>(g, MPMHIP_VISCO)) {  // src/particles.cpp:72-134
    mat3 U; float s[3], sn[3], ratio[3], d[3];
    visco_return<IN_LDS>(g, cdg, F, aux, U, s, sn, lds);
    const float Jn = sn[0] * sn[1] * sn[2];
    const float vol_l = g.p[3] * (Jn - 1.0f) * Jn;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      ratio[i] = sn[i] / s[i];
      d[i] = -vol * fmaf(2.0f * g.p[2], sn[i] * sn[i] - sn[i], vol_l);
    }
    mat3 Rr;
    sandwich2(U, ratio, d, Rr, stress);
    F = mat_mul(Rr, F);
    return;
  }
  F = mat_mul(cdg, F);
  if (mat_is<MATS>(g, MPMHIP_LINEAR)) {  // src/particles.cpp:329-341 (no eigen-solve)
    const float mu = g.p[2], la = g.p[3];
    const float tr = la * (F(0, 0) + F(1, 1) + F(2, 2) - 3.0f);
    mat3 Pk;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Pk(r, c) = mu * (F(r, c) + F(c, r) - ((r == c) ? 2.0f : 0.0f)) + ((r == c) ? tr : 0.0f);
    stress = mat_mul_bt(Pk, F);
#pragma unroll
    for (int i = 0; i < 9; i++) stress.m[i] *= -vol;
    return;
  }
  mat3 U; float lam[3], s[3];
  sym_eig3_FFt<IN_LDS>(F, U, lam, lds);
  const float detF = mat_det(F);
  signed_sigma(lam, detF, s);
  const float mu0 = g.p[2], la0 = g.p[3];
  float d[3];
  if (mat_is<MATS>(g, MPMHIP_JELLY)) {  // src/particles.cpp:391-416
    const float vol_l = la0 * (detF - 1.0f) * detF;
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = -vol * fmaf(2.0f * mu0, lam[i] - s[i], vol_l);
    stress = sandwich(U, d);
    return;
  }
  if (mat_is<MATS>(g, MPMHIP_ELASTIC)) {  // src/particles.cpp:798-812
    float ls[3];
#pragma unroll
    for (int i = 0; i < 3; i++) ls[i] = __logf(s[i]);
    const float tr = ls[0] + ls[1] + ls[2];
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = -vol * fmaf(2.0f * mu0, ls[i], la0 * tr);
    stress = sandwich(U, d);
    return;
  }
  float ratio[3];  // s'_i / s_i
  if (mat_is<MATS>(g, MPMHIP_SNOW)) {  // src/particles.cpp:207-252
    const float lo = 1.0f - g.p[5], hi = 1.0f + g.p[6];
    float det_o = 1.0f, det_n = 1.0f, sn[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      sn[i] = fminf(fmaxf(s[i], lo), hi);
      det_o *= s[i];
      det_n *= sn[i];
      ratio[i] = sn[i] * fast_rcp(s[i]);
    }
    float Jp = aux * det_o / det_n;
    if (!(Jp <= g.p[8])) Jp = g.p[8];
    if (!(Jp >= g.p[7])) Jp = g.p[7];
    aux = Jp;
    const float e = __expf(g.p[4] * (1.0f - Jp));
    const float mu = mu0 * e, la = la0 * e;
    const float vol_l = la * (det_n - 1.0f) * det_n;
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = -vol * fmaf(2.0f * mu, sn[i] * sn[i] - sn[i], vol_l);
  } else if (mat_is<MATS>(g, MPMHIP_SAND)) {  // src/particles.cpp:599-647
    const float alpha = g.p[4], coh = g.p[5], beta = g.p[6];
    float eps[3];
#pragma unroll
    for (int i = 0; i < 3; i++) eps[i] = __logf(fmaxf(fabsf(s[i]), 1e-4f)) - coh;
    const float sum = eps[0] + eps[1] + eps[2];
    const float tr = sum + aux;
    float eh[3];
#pragma unroll
    for (int i = 0; i < 3; i++) eh[i] = eps[i] - tr * (1.0f / 3.0f);
    const float ehn = fast_sqrt(fmaf(eh[0], eh[0], fmaf(eh[1], eh[1], eh[2] * eh[2])));
    float h[3];  // log of the projected singular values
    if (tr >= 0.0f) {
      h[0] = coh; h[1] = coh; h[2] = coh;
      aux = fmaf(beta, sum, aux);
    } else {
      aux = 0.0f;
      const float dg = ehn + (3.0f * la0 + 2.0f * mu0) / (2.0f * mu0) * tr * alpha;
      const float k = (dg <= 0.0f) ? 0.0f : dg * fast_rcp(ehn);
#pragma unroll
      for (int i = 0; i < 3; i++) h[i] = eps[i] - k * eh[i] + coh;
    }
    const float trh = h[0] + h[1] + h[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      ratio[i] = __expf(h[i]) * fast_rcp(s[i]);
      d[i] = -vol * fmaf(2.0f * mu0, h[i], la0 * trh);
    }
  } else if (mat_is<MATS>(g, MPMHIP_VON_MISES)) {  // src/particles.cpp:701-732
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; i++) e[i] = __logf(s[i]);
    const float tr = e[0] + e[1] + e[2];
    float eh[3] = {e[0] - tr * (1.0f / 3.0f), e[1] - tr * (1.0f / 3.0f), e[2] - tr * (1.0f / 3.0f)};
    const float n2 = fmaf(eh[0], eh[0], fmaf(eh[1], eh[1], eh[2] * eh[2]));
    const float dg = n2 - g.p[4] / (2.0f * mu0);
    float h[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      h[i] = (dg <= 0.0f) ? e[i] : e[i] - (dg / n2) * eh[i];
      ratio[i] = (dg <= 0.0f) ? 1.0f : __expf(h[i]) * fast_rcp(s[i]);
    }
    const float trh = h[0] + h[1] + h[2];
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = -vol * fmaf(2.0f * mu0, h[i], la0 * trh);
    if (dg <= 0.0f) { stress = sandwich(U, d); return; }
  } else {  // (a type outside this kernel's set, or an unknown type id: nothing to project, no stress)
#pragma unroll
    for (int i = 0; i < 3; i++) { ratio[i] = 1.0f; d[i] = 0.0f; }
  }
  mat3 Rr;
  sandwich2(U, ratio, d, Rr, stress);
  F = mat_mul(Rr, F);
}

// MPMParticle::get_allowed_dt(dx) per material — the explicit-integration bound the async stepper turns into the block's
// strength_dt_limit (src/async/async_mpm.cpp:105-111): dx / (c + |v|) with the material's sound speed c
//   visco / sand / von_mises / elastic (src/particles.cpp:136-155,649-665,734-750,814-830):
//        J = det F, rho = rho0 / J, K = 2 mu / 3 + lambda, c^2 = max(4 mu / (3 rho) + K (1 - log J) / rho0, 1e-20)
//   snow (:254-278):  J = det F * Jp, (mu, lambda) hardened by exp(h (1 - Jp)), c = sqrt((lambda + 2 mu) / rho)
//   water (:480-490): c^2 = k gamma / j^(gamma - 1)
//   linear / jelly (:343-345,418-420): 0 — the reference's async stepper stops on them (tmp_limit < 1)
__device__ __forceinline__ float allowed_dt(const GroupParams &g, const mat3 &F, float aux, const float v[3], float dx) {
  const float u = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float rho0 = g.p[0] / g.p[1];
  float c;
  switch (g.type) {
    case MPMHIP_LINEAR:
    case MPMHIP_JELLY:
      return 0.0f;
    case MPMHIP_WATER:
      c = sqrtf(g.p[2] * g.p[3] / powf(aux, g.p[3] - 1.0f));
      break;
    case MPMHIP_SNOW: {
      const float J = mat_det(F) * aux, rho = rho0 / J, e = expf(g.p[4] * (1.0f - aux));
      c = sqrtf((g.p[3] * e + 2.0f * g.p[2] * e) / rho);
      break;
    }
    default: {
      const float J = mat_det(F), rho = rho0 / J, K = 2.0f * g.p[2] / 3.0f + g.p[3];
      c = sqrtf(fmaxf(4.0f * g.p[2] / (3.0f * rho) + K * (1.0f - logf(J)) / rho0, 1e-20f));
    }
  }
  return dx / (c + u);
}

// friction_project — src/mpm_fwd.h:25-57
__device__ __forceinline__ void friction_project(float v[3], const float vb[3], const float n[3], float friction) {
  if (friction == -1.0f) { v[0] = vb[0]; v[1] = vb[1]; v[2] = vb[2]; return; }
  const bool slip = friction <= -2.0f;
  if (slip) friction = -friction - 2.0f;
  const float r0 = v[0] - vb[0], r1 = v[1] - vb[1], r2 = v[2] - vb[2];
  const float nn = n[0] * r0 + n[1] * r1 + n[2] * r2;
  const float t0 = r0 - nn * n[0], t1 = r1 - nn * n[1], t2 = r2 - nn * n[2];
  const float tn = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
  const float ts = fmaxf(tn + fminf(nn, 0.0f) * friction, 0.0f) / fmaxf(1e-30f, tn);
  const float keep = slip ? 0.0f : fmaxf(0.0f, nn);
  v[0] = ts * t0 + keep * n[0] + vb[0];
  v[1] = ts * t1 + keep * n[1] + vb[1];
  v[2] = ts * t2 + keep * n[2] + vb[2];
}

// Analytic level set: union of solids, phi = min over primitives, negative inside a solid (reference: taichi's
// sampled LevelSet built by add_plane / add_sphere / add_cuboid in the scene scripts; sample() and
// get_spatial_gradient() as used in src/mpm.cpp:323-326 and :416-421).  Returns phi in grid units and the unit
// gradient of the active primitive.  A DYNAMIC level set (taichi's DynamicLevelSet, which the python driver rebuilds
// every frame from levelset_generator(t0), levelset_generator(t1): scripts/async/async_mpm.py:119-127) is two key
// frames blended linearly in time: phi = lerp, gradient = normalised lerp of the two gradients,
// d phi / dt = (phi1 - phi0) / (t1 - t0), which gives the boundary velocity of src/mpm.cpp:340-342.
struct ShapeDev { int type, inside_out; float p[6]; };  // type 0 plane {n, d}, 1 sphere, 2 cuboid
struct LevelSetDev {
  int n;
  float friction;
  int particle_collision;
  int dynamic;  // 1: s1 / n1 hold the key frame at t1, s / n the one at t0
  int n1;
  float t0, t1;
  int dirichlet;  // MPM<3>::apply_dirichlet_boundary_conditions (src/mpm.cpp:401-412): grid nodes above y = 0.525 are held at rest
  ShapeDev s[MPMHIP_MAX_SHAPES];
  ShapeDev s1[MPMHIP_MAX_SHAPES];
};

__device__ __forceinline__ bool levelset_eval_key(const ShapeDev *S, int count, const float x[3], float idx, float &phi,
                                                  float n[3]) {
  if (count <= 0) return false;
  phi = 1e30f;
  for (int i = 0; i < count; i++) {
    const float *q = S[i].p;
    float ph, g[3];
    if (S[i].type == 0) {
      ph = q[0] * x[0] + q[1] * x[1] + q[2] * x[2] + q[3];
      g[0] = q[0]; g[1] = q[1]; g[2] = q[2];
    } else if (S[i].type == 1) {
      const float d0 = x[0] - q[0], d1 = x[1] - q[1], d2 = x[2] - q[2];
      const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
      const float inv = len > 0.0f ? 1.0f / len : 0.0f;
      ph = len - q[3];
      g[0] = d0 * inv; g[1] = d1 * inv; g[2] = d2 * inv;
    } else {
      bool inside = true;
      float near[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        inside = inside && q[k] <= x[k] && x[k] <= q[3 + k];
        near[k] = fminf(fmaxf(x[k], q[k]), q[3 + k]);
      }
      g[0] = g[1] = g[2] = 0.0f;
      if (inside) {
        float best = 1e30f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float dlo = x[k] - q[k], dhi = q[3 + k] - x[k];
          if (dlo < best) { best = dlo; g[0] = g[1] = g[2] = 0.0f; g[k] = -1.0f; }
          if (dhi < best) { best = dhi; g[0] = g[1] = g[2] = 0.0f; g[k] = 1.0f; }
        }
        ph = -best;
      } else {
        const float d0 = x[0] - near[0], d1 = x[1] - near[1], d2 = x[2] - near[2];
        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        ph = len;
        g[0] = d0 / len; g[1] = d1 / len; g[2] = d2 / len;
      }
    }
    if (S[i].type != 0 && S[i].inside_out) { ph = -ph; g[0] = -g[0]; g[1] = -g[1]; g[2] = -g[2]; }
    ph *= idx;
    if (ph < phi) { phi = ph; n[0] = g[0]; n[1] = g[1]; n[2] = g[2]; }
  }
  return true;
}

// the level set at time t: phi (grid units), unit gradient, and — when asked — d phi / dt in grid units per second
__device__ __forceinline__ bool levelset_eval(const LevelSetDev &L, float t, const float x[3], float idx, float &phi,
                                              float n[3], float *dphidt = nullptr) {
  if (dphidt) *dphidt = 0.0f;
  if (!levelset_eval_key(L.s, L.n, x, idx, phi, n)) return false;
  if (!L.dynamic) return true;
  float phi1, n1[3] = {0.0f, 0.0f, 0.0f};
  if (!levelset_eval_key(L.s1, L.n1, x, idx, phi1, n1)) return true;
  const float a = (t - L.t0) / (L.t1 - L.t0);
  if (dphidt) *dphidt = (phi1 - phi) / (L.t1 - L.t0);
  phi = (1.0f - a) * phi + a * phi1;
  const float g0 = n[0] * (1.0f - a) + n1[0] * a, g1 = n[1] * (1.0f - a) + n1[1] * a, g2 = n[2] * (1.0f - a) + n1[2] * a;
  const float len = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
  const float inv = len < 1e-10f ? 0.0f : 1.0f / len;
  n[0] = g0 * inv; n[1] = g1 * inv; n[2] = g2 * inv;
  return true;
}

// quadratic B-spline weights of MLSMPMFastKernel32 (src/transfer.cpp:168-186; src/kernel.h:126-130):
// p = rel_pos - 0.5 in [0,1);  t = p - (-0.5, 0.5, 1.5);  w = fma(c2, t*t, fma(c1, t, c0))
__device__ __forceinline__ void bspline_weights(float rel, float w[3]) {
  const float p = rel - 0.5f;
  const float t0 = p + 0.5f, t1 = p - 0.5f, t2 = p - 1.5f;
  w[0] = fmaf(0.5f, t0 * t0, fmaf(-1.5f, t0, 1.125f));
  w[1] = fmaf(-1.0f, t1 * t1, 0.75f);
  w[2] = fmaf(0.5f, t2 * t2, fmaf(1.5f, t2, 1.125f));
}

}  // namespace mpm
