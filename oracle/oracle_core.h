// oracle/oracle_core.h — shared math of the CPU oracle (TEST INFRASTRUCTURE ONLY).
// See mpm_oracle.h for the parity status ("parity unpinned" except kernel weights).
#pragma once
#include "mpm_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {


typedef float real;

// ---------------------------------------------------------------- tiny linalg
struct M3 {
  real a[9];
  real &operator()(int r, int c) { return a[3 * r + c]; }
  real operator()(int r, int c) const { return a[3 * r + c]; }
};
inline M3 m3_zero() { M3 m; for (int i = 0; i < 9; i++) m.a[i] = 0; return m; }
inline M3 m3_diag(real x, real y, real z) { M3 m = m3_zero(); m(0,0) = x; m(1,1) = y; m(2,2) = z; return m; }
inline M3 m3_id(real s = 1) { return m3_diag(s, s, s); }
inline M3 load3(const float *p) { M3 m; for (int i = 0; i < 9; i++) m.a[i] = p[i]; return m; }
inline void store3(const M3 &m, float *p) { for (int i = 0; i < 9; i++) p[i] = m.a[i]; }
inline M3 operator*(const M3 &A, const M3 &B) {
  M3 C;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      real s = 0;
      for (int k = 0; k < 3; k++) s += A(r, k) * B(k, c);
      C(r, c) = s;
    }
  return C;
}
inline M3 operator*(real s, const M3 &A) { M3 C; for (int i = 0; i < 9; i++) C.a[i] = s * A.a[i]; return C; }
inline M3 operator+(const M3 &A, const M3 &B) { M3 C; for (int i = 0; i < 9; i++) C.a[i] = A.a[i] + B.a[i]; return C; }
inline M3 operator-(const M3 &A, const M3 &B) { M3 C; for (int i = 0; i < 9; i++) C.a[i] = A.a[i] - B.a[i]; return C; }
inline M3 transposed(const M3 &A) { M3 C; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C(r, c) = A(c, r); return C; }
inline real determinant(const M3 &m) {
  return m(0,0) * (m(1,1) * m(2,2) - m(1,2) * m(2,1)) - m(0,1) * (m(1,0) * m(2,2) - m(1,2) * m(2,0)) +
         m(0,2) * (m(1,0) * m(2,1) - m(1,1) * m(2,0));
}
inline M3 inversed(const M3 &m) {
  real det = determinant(m);
  real id = 1.0f / det;
  M3 r;
  r(0,0) = (m(1,1) * m(2,2) - m(1,2) * m(2,1)) * id;
  r(0,1) = (m(0,2) * m(2,1) - m(0,1) * m(2,2)) * id;
  r(0,2) = (m(0,1) * m(1,2) - m(0,2) * m(1,1)) * id;
  r(1,0) = (m(1,2) * m(2,0) - m(1,0) * m(2,2)) * id;
  r(1,1) = (m(0,0) * m(2,2) - m(0,2) * m(2,0)) * id;
  r(1,2) = (m(0,2) * m(1,0) - m(0,0) * m(1,2)) * id;
  r(2,0) = (m(1,0) * m(2,1) - m(1,1) * m(2,0)) * id;
  r(2,1) = (m(0,1) * m(2,0) - m(0,0) * m(2,1)) * id;
  r(2,2) = (m(0,0) * m(1,1) - m(0,1) * m(1,0)) * id;
  return r;
}

// ---------------------------------------------------------------- SVD / polar
// taichi core `svd` / `polar_decomp` (<taichi/math/svd.h>, un-vendored; call
// sites src/particles.cpp:76,107,212,227,394,630,642).  Convention adopted
// (SURVEY §8c): U, V proper rotations, |sigma| sorted descending, the sign of
// det(F) carried by the last singular value.  Internals in double (Jacobi on
// F^T F iterated to convergence) so that the oracle is the accurate side of
// every comparison; inputs/outputs are fp32 like the reference's.
inline void jacobi_eig3(double S[3][3], double V[3][3]) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    double diag = S[0][0] * S[0][0] + S[1][1] * S[1][1] + S[2][2] * S[2][2];
    if (off <= 1e-34 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = S[p][q];
        if (apq == 0.0) continue;
        double theta = (S[q][q] - S[p][p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        double J[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        J[p][p] = c; J[q][q] = c; J[p][q] = s; J[q][p] = -s;
        double T[3][3], N[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[i][j] = 0; for (int k = 0; k < 3; k++) T[i][j] += S[i][k] * J[k][j]; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { N[i][j] = 0; for (int k = 0; k < 3; k++) N[i][j] += J[k][i] * T[k][j]; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S[i][j] = 0.5 * (N[i][j] + N[j][i]);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[i][j] = 0; for (int k = 0; k < 3; k++) T[i][j] += V[i][k] * J[k][j]; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = T[i][j];
      }
  }
}

inline void svd3_d(const double A[3][3], double U[3][3], double sig[3], double V[3][3]) {
  double S[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { S[i][j] = 0; for (int k = 0; k < 3; k++) S[i][j] += A[k][i] * A[k][j]; }
  jacobi_eig3(S, V);
  double lam[3] = {S[0][0], S[1][1], S[2][2]};
  // sort eigenpairs descending
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (lam[j] < lam[j + 1]) {
        std::swap(lam[j], lam[j + 1]);
        for (int k = 0; k < 3; k++) std::swap(V[k][j], V[k][j + 1]);
      }
  double detV = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
  if (detV < 0) for (int k = 0; k < 3; k++) V[k][2] = -V[k][2];
  double Bm[3][3];  // columns b_i = A v_i
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Bm[i][j] = 0; for (int k = 0; k < 3; k++) Bm[i][j] += A[i][k] * V[k][j]; }
  double u1[3], u2[3], u3[3];
  double n1 = std::sqrt(Bm[0][0] * Bm[0][0] + Bm[1][0] * Bm[1][0] + Bm[2][0] * Bm[2][0]);
  if (n1 > 1e-300) { for (int k = 0; k < 3; k++) u1[k] = Bm[k][0] / n1; } else { u1[0] = 1; u1[1] = 0; u1[2] = 0; }
  double d12 = u1[0] * Bm[0][1] + u1[1] * Bm[1][1] + u1[2] * Bm[2][1];
  for (int k = 0; k < 3; k++) u2[k] = Bm[k][1] - d12 * u1[k];
  double n2 = std::sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  if (n2 > 1e-12 * (n1 + 1e-300)) { for (int k = 0; k < 3; k++) u2[k] /= n2; }
  else {  // rank <= 1: any unit vector orthogonal to u1
    int m = 0; if (std::fabs(u1[1]) < std::fabs(u1[m])) m = 1; if (std::fabs(u1[2]) < std::fabs(u1[m])) m = 2;
    double e[3] = {0, 0, 0}; e[m] = 1;
    double d = u1[m];
    for (int k = 0; k < 3; k++) u2[k] = e[k] - d * u1[k];
    double nn = std::sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int k = 0; k < 3; k++) u2[k] /= nn;
  }
  u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
  u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
  u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
  for (int k = 0; k < 3; k++) { U[k][0] = u1[k]; U[k][1] = u2[k]; U[k][2] = u3[k]; }
  for (int j = 0; j < 3; j++) sig[j] = U[0][j] * Bm[0][j] + U[1][j] * Bm[1][j] + U[2][j] * Bm[2][j];
}

inline void svd(const M3 &F, M3 &U, M3 &Sig, M3 &V) {
  double A[3][3], Ud[3][3], s[3], Vd[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = F(i, j);
  svd3_d(A, Ud, s, Vd);
  Sig = m3_diag((real)s[0], (real)s[1], (real)s[2]);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U(i, j) = (real)Ud[i][j]; V(i, j) = (real)Vd[i][j]; }
}

inline void polar_decomp(const M3 &F, M3 &R, M3 &S) {
  double A[3][3], Ud[3][3], s[3], Vd[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = F(i, j);
  svd3_d(A, Ud, s, Vd);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double r = 0, ss = 0;
      for (int k = 0; k < 3; k++) { r += Ud[i][k] * Vd[j][k]; ss += Vd[i][k] * s[k] * Vd[j][k]; }
      R(i, j) = (real)r; S(i, j) = (real)ss;
    }
}

// 2x2 (closed-form polar, then one Jacobi rotation) — used by the 88-line demo
inline void polar2_d(const double A[2][2], double R[2][2], double S[2][2]) {
  double x = A[0][0] + A[1][1], y = A[1][0] - A[0][1];
  double d = std::sqrt(x * x + y * y);
  double c = 1, s = 0;
  if (d > 0) { c = x / d; s = y / d; }
  R[0][0] = c; R[0][1] = -s; R[1][0] = s; R[1][1] = c;
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) S[i][j] = R[0][i] * A[0][j] + R[1][i] * A[1][j];
}
inline void svd2_d(const double A[2][2], double U[2][2], double sig[2], double V[2][2]) {
  // polar first (closed form), then the symmetric 2x2 eigenproblem of S; U = R V.
  double R[2][2], S[2][2];
  polar2_d(A, R, S);
  double s01 = 0.5 * (S[0][1] + S[1][0]);
  double a = 0.5 * (S[0][0] + S[1][1]), tao = 0.5 * (S[0][0] - S[1][1]);
  double w = std::sqrt(tao * tao + s01 * s01);
  sig[0] = a + w; sig[1] = a - w;
  double vx, vy;  // unit eigenvector of sig[0]
  if (w < 1e-300) { vx = 1; vy = 0; }
  else if (tao >= 0) { vx = tao + w; vy = s01; }
  else { vx = s01; vy = w - tao; }
  double nn = std::sqrt(vx * vx + vy * vy);
  if (nn < 1e-300) { vx = 1; vy = 0; nn = 1; }
  vx /= nn; vy /= nn;
  V[0][0] = vx; V[1][0] = vy; V[0][1] = -vy; V[1][1] = vx;
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) U[i][j] = R[i][0] * V[0][j] + R[i][1] * V[1][j];
}

// ---------------------------------------------------------------- kernels
// MPMKernel<dim,2>::calculate_kernel — src/kernel.h:123-134 (and :198-209).
// p_fract = fract(pos - 0.5); t = p_fract - (-0.5, 0.5, 1.5);
// w = (0.5,-1,0.5)*t*t + (-1.5,0,1.5)*t + (1.125,0.75,1.125); dw = (1,-2,1)*t + (-1.5,0,1.5)
inline real fract(real x) { return x - std::floor(x); }
inline void quad_w_dw(real p_fract, real w[3], real dw[3]) {
  const real c2[3] = {0.5f, -1.0f, 0.5f}, c1[3] = {-1.5f, 0.0f, 1.5f}, c0[3] = {1.125f, 0.75f, 1.125f};
  const real off[3] = {-0.5f, 0.5f, 1.5f}, d1[3] = {1.0f, -2.0f, 1.0f};
  for (int k = 0; k < 3; k++) {
    real t = p_fract - off[k];
    real tt = t * t;
    w[k] = c2[k] * tt + c1[k] * t + c0[k];
    dw[k] = d1[k] * t + c1[k];
  }
}
// MLSMPMFastKernel32 — src/transfer.cpp:168-186: same polynomial, evaluated with two FMAs
inline void quad_w_fma(real p_fract, real w[3]) {
  const real c2[3] = {0.5f, -1.0f, 0.5f}, c1[3] = {-1.5f, 0.0f, 1.5f}, c0[3] = {1.125f, 0.75f, 1.125f};
  const real off[3] = {-0.5f, 0.5f, 1.5f};
  for (int k = 0; k < 3; k++) {
    real t = p_fract - off[k];
    real tt = t * t;
    w[k] = std::fmaf(c2[k], tt, std::fmaf(c1[k], t, c0[k]));
  }
}
// MPMKernel<dim,2>::get_stencil_start — src/kernel.h:119-121
inline int stencil_start(real x) { return int(x - 0.5f); }

// ---------------------------------------------------------------- materials
// gp[] layout: mpm_oracle.h
inline M3 first_piola_fixed_corotated(const M3 &F, real mu, real lambda) {
  // src/particles.cpp:391-398 (jelly), :207-216 (snow), :72-80 (visco)
  real j = determinant(F);
  M3 r, s;
  polar_decomp(F, r, s);
  return 2 * mu * (F - r) + (lambda * (j - 1) * j) * inversed(transposed(F));
}
inline M3 hencky_force(const M3 &F, real mu0, real lambda0, real vol) {
  // src/particles.cpp:628-637 (sand), :701-711 (von_mises), :798-807 (elastic)
  M3 u, v, sig;
  svd(F, u, sig, v);
  real ls[3], is[3];
  for (int d = 0; d < 3; d++) { ls[d] = std::log(sig(d, d)); is[d] = 1.0f / sig(d, d); }
  real tr = ls[0] + ls[1] + ls[2];
  M3 center = m3_zero();
  for (int d = 0; d < 3; d++) center(d, d) = 2.0f * mu0 * is[d] * ls[d] + lambda0 * tr * is[d];
  return (-vol) * ((u * center * transposed(v)) * transposed(F));
}

inline M3 calculate_force(int type, const float *gp, const M3 &F, real aux) {
  const real vol = gp[1];
  switch (type) {
    case ORC_JELLY:  // src/particles.cpp:409-411
    case ORC_VISCO:  // src/particles.cpp:82-85
      return (-vol) * (first_piola_fixed_corotated(F, gp[2], gp[3]) * transposed(F));
    case ORC_SNOW: {  // src/particles.cpp:218-220, 244-252
      real e = std::exp(gp[4] * (1.0f - aux));
      return (-vol) * (first_piola_fixed_corotated(F, gp[2] * e, gp[3] * e) * transposed(F));
    }
    case ORC_LINEAR: {  // src/particles.cpp:329-336
      real mu = gp[2], lambda = gp[3];
      M3 P = mu * (F + transposed(F) - m3_id(2.0f)) + m3_id(lambda * ((F(0,0) + F(1,1) + F(2,2)) - 3));
      return (-vol) * (P * transposed(F));
    }
    case ORC_WATER: {  // src/particles.cpp:463-467
      real j = aux, k = gp[2], gamma = gp[3];
      real p = k * (std::pow(j, -gamma) - 1.f);
      M3 sigma = m3_id(-p);
      return (-vol * j) * sigma;
    }
    case ORC_SAND:
    case ORC_VON_MISES:
    case ORC_ELASTIC:
      return hencky_force(F, gp[2], gp[3], vol);
  }
  return m3_zero();
}

// SandParticle::project — src/particles.cpp:599-626
inline void sand_project(const M3 &sigma, real alpha, real cohesion, real beta, real lambda_0, real mu_0, real &logJp,
                  M3 &sigma_out) {
  const real d = 3;
  real eps[3];
  for (int i = 0; i < 3; i++) eps[i] = std::log(std::max(std::abs(sigma(i, i)), 1e-4f)) - cohesion;
  real sum = eps[0] + eps[1] + eps[2];
  real tr = sum + logJp;
  real eh[3];
  for (int i = 0; i < 3; i++) eh[i] = eps[i] - tr / d;
  real eh_for = std::sqrt(eh[0] * eh[0] + eh[1] * eh[1] + eh[2] * eh[2]);
  if (tr >= 0.0f) {
    sigma_out = m3_id(std::exp(cohesion));
    logJp = beta * sum + logJp;
  } else {
    logJp = 0;
    real delta_gamma = eh_for + (d * lambda_0 + 2 * mu_0) / (2 * mu_0) * tr * alpha;
    if (delta_gamma <= 0) {
      sigma_out = m3_diag(std::exp(eps[0] + cohesion), std::exp(eps[1] + cohesion), std::exp(eps[2] + cohesion));
    } else {
      real h[3];
      for (int i = 0; i < 3; i++) h[i] = eps[i] - delta_gamma / eh_for * eh[i] + cohesion;
      sigma_out = m3_diag(std::exp(h[0]), std::exp(h[1]), std::exp(h[2]));
    }
  }
}

inline void plasticity(int type, const float *gp, const M3 &cdg, M3 &F, real &aux) {
  switch (type) {
    case ORC_JELLY:    // src/particles.cpp:413-416
    case ORC_LINEAR:   // :338-341
    case ORC_ELASTIC:  // :809-812
      F = cdg * F;
      return;
    case ORC_SNOW: {  // src/particles.cpp:222-242
      F = cdg * F;
      real theta_c = gp[5], theta_s = gp[6], min_Jp = gp[7], max_Jp = gp[8];
      real det_orig = 1.0f, det_new = 1.0f;
      M3 u, sig, v;
      svd(F, u, sig, v);
      for (int i = 0; i < 3; i++) {
        det_orig *= sig(i, i);
        sig(i, i) = std::min(std::max(sig(i, i), 1.0f - theta_c), 1.0f + theta_s);
        det_new *= sig(i, i);
      }
      F = u * sig * transposed(v);
      real Jp_new = aux * det_orig / det_new;
      if (!(Jp_new <= max_Jp)) Jp_new = max_Jp;
      if (!(Jp_new >= min_Jp)) Jp_new = min_Jp;
      aux = Jp_new;
      return;
    }
    case ORC_WATER: {  // src/particles.cpp:469-478
      aux *= (cdg(0,0) + cdg(1,1) + cdg(2,2)) - (3 - 1);
      if (aux < 0.1f) aux = 0.1f;
      return;
    }
    case ORC_SAND: {  // src/particles.cpp:639-647
      F = cdg * F;
      M3 u, v, sig, t = m3_id();
      svd(F, u, sig, v);
      sand_project(sig, gp[4], gp[5], gp[6], gp[3], gp[2], aux, t);
      F = u * t * transposed(v);
      return;
    }
    case ORC_VON_MISES: {  // src/particles.cpp:713-732
      F = cdg * F;
      M3 U, V, sigma;
      svd(F, U, sigma, V);
      real e[3] = {std::log(sigma(0,0)), std::log(sigma(1,1)), std::log(sigma(2,2))};
      real tr = e[0] + e[1] + e[2];
      real eh[3] = {e[0] - tr / 3, e[1] - tr / 3, e[2] - tr / 3};
      real ehn = eh[0] * eh[0] + eh[1] * eh[1] + eh[2] * eh[2];  // frobenius_norm2 (sic, squared)
      real dg = ehn - gp[4] / (2.0f * gp[2]);
      if (dg <= 0) return;
      real H[3];
      for (int i = 0; i < 3; i++) H[i] = e[i] - (dg / ehn) * eh[i];
      F = U * m3_diag(std::exp(H[0]), std::exp(H[1]), std::exp(H[2])) * transposed(V);
      return;
    }
    case ORC_VISCO: {  // src/particles.cpp:87-134
      real dt = gp[6], visco_nu = gp[4], visco_kappa = gp[5];
      // approximate_exponent(dt, (cdg - I)/dt), recursion unrolled as a loop (:87-100)
      M3 m = (1.0f / dt) * (cdg - m3_id());
      int halvings = 0;
      M3 r;
      real h = dt;
      for (;;) {
        M3 s = h * m;
        r = (0.5f * s + m3_id()) * s + m3_id();
        if (determinant(r) > 0.0f || halvings > 20) break;
        h *= 0.5f; halvings++;
      }
      for (int i = 0; i < halvings; i++) r = r * r;
      M3 Fh = r * F;
      M3 u, sig, v;
      svd(Fh, u, sig, v);
      real pnorm;
      {
        M3 P = first_piola_fixed_corotated(F, gp[2], gp[3]);
        real s2 = 0; for (int i = 0; i < 9; i++) s2 += P.a[i] * P.a[i];
        pnorm = std::sqrt(s2);
      }
      real gamma = 0.0f;
      if (pnorm > 1e-5f) gamma = std::min(std::max(dt * visco_nu * (pnorm - aux) / pnorm, 0.0f), 1.0f);
      real scale = 1.0f;
      real dets = sig(0,0) * sig(1,1) * sig(2,2);
      if (std::abs(dets) > 1e-5f) scale = 1.0f / std::pow(dets, 1.0f / 3.0f);
      M3 mid_inv = m3_zero();
      for (int d = 0; d < 3; d++) {
        real md = std::pow(sig(d, d) * scale, gamma);
        mid_inv(d, d) = std::abs(md) > 1e-5f ? 1.0f / md : 1.0f;
      }
      F = u * sig * mid_inv * transposed(v);
      svd(F, u, sig, v);
      for (int d = 0; d < 3; d++) sig(d, d) = std::min(std::max(sig(d, d), 0.1f), 10.0f);
      F = u * sig * transposed(v);
      aux += visco_kappa * gamma * pnorm;
      return;
    }
  }
}

// friction_project — src/mpm_fwd.h:25-57
inline void friction_project(const real v[3], const real vb[3], const real n[3], real friction, real out[3]) {
  real r[3] = {v[0] - vb[0], v[1] - vb[1], v[2] - vb[2]};
  if (friction == -1) { out[0] = vb[0]; out[1] = vb[1]; out[2] = vb[2]; return; }
  bool slip = friction <= -2;
  if (slip) friction = -friction - 2;
  real nn = n[0] * r[0] + n[1] * r[1] + n[2] * r[2];
  real t[3] = {r[0] - nn * n[0], r[1] - nn * n[1], r[2] - nn * n[2]};
  real tn = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  real ts = std::max(tn + std::min(nn, 0.0f) * friction, 0.0f) / std::max(1e-30f, tn);
  real keep = std::max(0.0f, nn * real(!slip));
  for (int k = 0; k < 3; k++) out[k] = ts * t[k] + keep * n[k] + vb[k];
}

inline int64_t node_index(const orc_config *c, int i, int j, int k) {
  return ((int64_t)i * (c->res[1] + 1) + j) * (c->res[2] + 1) + k;
}

// one key frame of the level set, in grid units at a grid-unit position: phi and its (unit) spatial gradient
// (reference: levelset.sample(pos,t) / get_spatial_gradient(pos,t), src/mpm.cpp:323-326,416-421)
template <typename ShapeT>
inline bool levelset_eval_key(const orc_config *c, int n_planes, const float (*planes)[4], int n_shapes, const ShapeT *shapes,
                              const real pos_grid[3], real &phi, real n[3]) {
  if (n_planes <= 0 && n_shapes <= 0) return false;
  const real idx = 1.0f / c->dx;
  const real x[3] = {pos_grid[0] * c->dx, pos_grid[1] * c->dx, pos_grid[2] * c->dx};
  phi = 1e30f;
  for (int p = 0; p < n_planes; p++) {
    const float *pl = planes[p];
    real ph = (pl[0] * x[0] + pl[1] * x[1] + pl[2] * x[2] + pl[3]) * idx;
    if (ph < phi) { phi = ph; n[0] = pl[0]; n[1] = pl[1]; n[2] = pl[2]; }
  }
  for (int s = 0; s < n_shapes; s++) {
    const float *q = shapes[s].p;
    real ph, g[3];
    if (shapes[s].type == 1) {  // sphere: distance to the surface, negative inside the ball
      real d[3] = {x[0] - q[0], x[1] - q[1], x[2] - q[2]};
      real len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      real inv = len > 0 ? 1.0f / len : 0.0f;
      ph = len - q[3];
      for (int k = 0; k < 3; k++) g[k] = d[k] * inv;
    } else {  // cuboid: negative inside the box (distance to the nearest face), Euclidean distance outside
      bool inside = true;
      real near[3];
      for (int k = 0; k < 3; k++) {
        inside = inside && q[k] <= x[k] && x[k] <= q[3 + k];
        near[k] = std::min(std::max(x[k], q[k]), q[3 + k]);
      }
      g[0] = g[1] = g[2] = 0;
      if (inside) {
        real best = 1e30f;
        for (int k = 0; k < 3; k++) {
          real dlo = x[k] - q[k], dhi = q[3 + k] - x[k];
          if (dlo < best) { best = dlo; g[0] = g[1] = g[2] = 0; g[k] = -1; }
          if (dhi < best) { best = dhi; g[0] = g[1] = g[2] = 0; g[k] = 1; }
        }
        ph = -best;
      } else {
        real d[3] = {x[0] - near[0], x[1] - near[1], x[2] - near[2]};
        real len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        ph = len;
        for (int k = 0; k < 3; k++) g[k] = d[k] / len;
      }
    }
    if (shapes[s].inside_out) { ph = -ph; g[0] = -g[0]; g[1] = -g[1]; g[2] = -g[2]; }
    ph *= idx;
    if (ph < phi) { phi = ph; n[0] = g[0]; n[1] = g[1]; n[2] = g[2]; }
  }
  return true;
}

// the level set at time c->t.  Static: key frame 0.  Dynamic (DynamicLevelSet of the taichi core, as the python driver
// builds it per frame, scripts/async/async_mpm.py:119-127): the two key frames blended linearly in time; *dphidt =
// get_temporal_derivative in grid units per second (0 when static).
inline bool levelset_eval(const orc_config *c, const real pos_grid[3], real &phi, real n[3], real *dphidt = nullptr) {
  if (dphidt) *dphidt = 0.0f;
  if (!levelset_eval_key(c, c->n_planes, c->planes, c->n_shapes, c->shapes, pos_grid, phi, n)) return false;
  if (!c->dynamic) return true;
  real phi1, n1[3] = {0, 0, 0};
  if (!levelset_eval_key(c, c->n_planes1, c->planes1, c->n_shapes1, c->shapes1, pos_grid, phi1, n1)) return true;
  const real a = (c->t - c->t0) / (c->t1 - c->t0);
  if (dphidt) *dphidt = (phi1 - phi) / (c->t1 - c->t0);
  phi = (1.0f - a) * phi + a * phi1;
  real g[3], len2 = 0;
  for (int k = 0; k < 3; k++) { g[k] = n[k] * (1.0f - a) + n1[k] * a; len2 += g[k] * g[k]; }
  const real len = std::sqrt(len2);
  for (int k = 0; k < 3; k++) n[k] = len < 1e-10f ? 0.0f : g[k] / len;
  return true;
}

inline bool particle_alive(const orc_config *c, const float *x, const float *v) {
  const real idx = 1.0f / c->dx;
  for (int k = 0; k < 3; k++) {
    if (!std::isfinite(x[k]) || !std::isfinite(v[k])) return false;
  }
  real X[3] = {x[0] * idx, x[1] * idx, x[2] * idx};
  if (c->clean_boundary) {  // near_boundary: src/mpm.h:269-276
    real mn = std::min(X[0], std::min(X[1], X[2]));
    real mx = std::max(X[0] - c->res[0], std::max(X[1] - c->res[1], X[2] - c->res[2]));
    if (mn < 7.0f || mx > -7.0f) return false;
  }
  for (int k = 0; k < 3; k++) {  // stencil must stay on the grid (reference: undefined behaviour)
    if (!(X[k] >= 0.5f)) return false;
    int b = stencil_start(X[k]);
    if (b < 0 || b + 2 > c->res[k]) return false;
  }
  return true;
}


}  // namespace orc
