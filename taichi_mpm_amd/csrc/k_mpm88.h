// taichi_mpm_amd/csrc/k_mpm88.h — the 2D dense-grid MLS-MPM demo (BASELINE configs[0]; mls-mpm88.cpp:16-69,
// annotated twin mls-mpm88-explained.cpp:64-197) as three device kernels: an (n+1)^2 grid of (m v.x, m v.y, m), P2G with
// world-unit dpos and the inline snow model, grid normalise + gravity -200 + box walls, G2P with the sigma clamp.
// Small by design (n = 80, thousands of particles): one thread per particle / node, global float atomics for the
// scatter — the demo is the reference's own CPU-sized case, not a bandwidth problem.  Part of libmpmhip.
#pragma once
#include <hip/hip_runtime.h>

namespace mpm88 {

struct Params {
  int n;             // grid cells per axis (nodes: n + 1)
  float dt, dx, inv_dx;
  float mu_0, lambda_0, hardening;  // E = 1e4, nu = 0.2, hardening 10 (mls-mpm88.cpp:7-8)
  float mass, vol;   // particle_mass = vol = 1
  int plastic;
};

struct Weights { float w[3][2]; int bc[2]; float fx[2]; };
__device__ __forceinline__ Weights weights(const Params &P, float x, float y) {  // :19-21 / :48-50
  Weights W;
  const float X[2] = {x * P.inv_dx, y * P.inv_dx};
#pragma unroll
  for (int d = 0; d < 2; d++) {
    W.bc[d] = (int)(X[d] - 0.5f);
    const float f = X[d] - (float)W.bc[d];
    W.fx[d] = f;
    W.w[0][d] = 0.5f * (1.5f - f) * (1.5f - f);
    W.w[1][d] = 0.75f - (f - 1.0f) * (f - 1.0f);
    W.w[2][d] = 0.5f * (f - 0.5f) * (f - 0.5f);
  }
  return W;
}

// A = R S with R a rotation: (c, s) ~ (a00 + a11, a10 - a01)   (closed-form 2x2 polar decomposition)
__device__ __forceinline__ void polar2(const float a[4], float r[4], float s[4]) {
  const float x = a[0] + a[3], y = a[2] - a[1];
  const float d = sqrtf(x * x + y * y);
  const float c = d > 0.0f ? x / d : 1.0f, sn = d > 0.0f ? y / d : 0.0f;
  r[0] = c; r[1] = -sn; r[2] = sn; r[3] = c;
  s[0] = c * a[0] + sn * a[2]; s[1] = c * a[1] + sn * a[3];   // S = R^T A
  s[2] = -sn * a[0] + c * a[2]; s[3] = -sn * a[1] + c * a[3];
}
// A = U diag(sig) V^T: polar part + one Jacobi rotation of the symmetric factor
__device__ __forceinline__ void svd2(const float a[4], float u[4], float sig[2], float v[4]) {
  float r[4], s[4];
  polar2(a, r, s);
  const float b = 0.5f * (s[1] + s[2]);
  float c = 1.0f, sn = 0.0f;
  sig[0] = s[0]; sig[1] = s[3];
  if (fabsf(b) > 1e-30f) {
    const float tau = (s[3] - s[0]) / (2.0f * b);
    const float t = (tau >= 0.0f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
    c = 1.0f / sqrtf(1.0f + t * t);
    sn = c * t;
    sig[0] = s[0] - t * b;
    sig[1] = s[3] + t * b;
  }
  v[0] = c; v[1] = sn; v[2] = -sn; v[3] = c;                    // S = V diag(sig) V^T
  u[0] = r[0] * v[0] + r[1] * v[2]; u[1] = r[0] * v[1] + r[1] * v[3];  // U = R V
  u[2] = r[2] * v[0] + r[3] * v[2]; u[3] = r[2] * v[1] + r[3] * v[3];
}

// P2G — mls-mpm88.cpp:18-36.  grid[(i (n+1) + j) * 3 + {0,1,2}] += w (m v + affine dpos, m)
__global__ __launch_bounds__(256) void k_p2g(Params P, int64_t np, const float *__restrict__ x, const float *__restrict__ v,
                                             const float *__restrict__ F, const float *__restrict__ C,
                                             const float *__restrict__ Jp, float *__restrict__ grid) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const Weights W = weights(P, x[2 * p], x[2 * p + 1]);
  const float e = expf(P.hardening * (1.0f - Jp[p])), mu = P.mu_0 * e, lambda = P.lambda_0 * e;
  const float f[4] = {F[4 * p], F[4 * p + 1], F[4 * p + 2], F[4 * p + 3]};
  const float J = f[0] * f[3] - f[1] * f[2];
  float r[4], s[4];
  polar2(f, r, s);
  const float fr[4] = {f[0] - r[0], f[1] - r[1], f[2] - r[2], f[3] - r[3]};
  const float m[4] = {fr[0] * f[0] + fr[1] * f[1], fr[0] * f[2] + fr[1] * f[3],   // (F - R) F^T
                      fr[2] * f[0] + fr[3] * f[1], fr[2] * f[2] + fr[3] * f[3]};
  const float k = -4.0f * P.inv_dx * P.inv_dx * P.dt * P.vol;
  const float vol_term = lambda * (J - 1.0f) * J;
  const float affine[4] = {k * (2.0f * mu * m[0] + vol_term) + P.mass * C[4 * p], k * (2.0f * mu * m[1]) + P.mass * C[4 * p + 1],
                           k * (2.0f * mu * m[2]) + P.mass * C[4 * p + 2], k * (2.0f * mu * m[3] + vol_term) + P.mass * C[4 * p + 3]};
  const float mvx = v[2 * p] * P.mass, mvy = v[2 * p + 1] * P.mass;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float dpx = ((float)i - W.fx[0]) * P.dx, dpy = ((float)j - W.fx[1]) * P.dx;
      const float ww = W.w[i][0] * W.w[j][1];
      const int gi = W.bc[0] + i, gj = W.bc[1] + j;
      if ((unsigned)gi > (unsigned)P.n || (unsigned)gj > (unsigned)P.n) continue;  // (the demo's walls keep particles inside)
      float *g = grid + 3 * ((size_t)gi * (P.n + 1) + gj);
      atomicAdd(g + 0, ww * (mvx + affine[0] * dpx + affine[1] * dpy));
      atomicAdd(g + 1, ww * (mvy + affine[2] * dpx + affine[3] * dpy));
      atomicAdd(g + 2, ww * P.mass);
    }
}

// grid — mls-mpm88.cpp:37-46: v = mv / m, gravity, sticky side/top walls, separating floor
__global__ __launch_bounds__(256) void k_grid(Params P, float *__restrict__ grid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = P.n + 1;
  if (t >= nn * nn) return;
  float *g = grid + 3 * (size_t)t;
  if (!(g[2] > 0.0f)) return;
  const float m = g[2];
  float vx = g[0] / m, vy = g[1] / m, one = g[2] / m;
  vy += P.dt * -200.0f;
  const float boundary = 0.05f, xx = (float)(t / nn) / (float)P.n, yy = (float)(t % nn) / (float)P.n;
  if (xx < boundary || xx > 1.0f - boundary || yy > 1.0f - boundary) { vx = 0.0f; vy = 0.0f; one = 0.0f; }
  if (yy < boundary) vy = fmaxf(0.0f, vy);
  g[0] = vx; g[1] = vy; g[2] = one;
}

// G2P — mls-mpm88.cpp:47-68
__global__ __launch_bounds__(256) void k_g2p(Params P, int64_t np, float *__restrict__ x, float *__restrict__ v,
                                             float *__restrict__ F, float *__restrict__ C, float *__restrict__ Jp,
                                             const float *__restrict__ grid) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const Weights W = weights(P, x[2 * p], x[2 * p + 1]);
  float Cn[4] = {0, 0, 0, 0}, vn[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int gi = W.bc[0] + i, gj = W.bc[1] + j;
      if ((unsigned)gi > (unsigned)P.n || (unsigned)gj > (unsigned)P.n) continue;
      const float *g = grid + 3 * ((size_t)gi * (P.n + 1) + gj);
      const float dpx = (float)i - W.fx[0], dpy = (float)j - W.fx[1];
      const float ww = W.w[i][0] * W.w[j][1];
      const float wx = ww * g[0], wy = ww * g[1];
      vn[0] += wx; vn[1] += wy;
      Cn[0] += 4.0f * P.inv_dx * wx * dpx; Cn[1] += 4.0f * P.inv_dx * wx * dpy;
      Cn[2] += 4.0f * P.inv_dx * wy * dpx; Cn[3] += 4.0f * P.inv_dx * wy * dpy;
    }
#pragma unroll
  for (int i = 0; i < 4; i++) C[4 * p + i] = Cn[i];
  v[2 * p] = vn[0]; v[2 * p + 1] = vn[1];
  x[2 * p] += P.dt * vn[0]; x[2 * p + 1] += P.dt * vn[1];
  const float f[4] = {F[4 * p], F[4 * p + 1], F[4 * p + 2], F[4 * p + 3]};
  const float a[4] = {1.0f + P.dt * Cn[0], P.dt * Cn[1], P.dt * Cn[2], 1.0f + P.dt * Cn[3]};
  const float Fn[4] = {a[0] * f[0] + a[1] * f[2], a[0] * f[1] + a[1] * f[3], a[2] * f[0] + a[3] * f[2], a[2] * f[1] + a[3] * f[3]};
  float U[4], sg[2], V[4];
  svd2(Fn, U, sg, V);
  if (P.plastic) {
#pragma unroll
    for (int i = 0; i < 2; i++) sg[i] = fminf(fmaxf(sg[i], 1.0f - 2.5e-2f), 1.0f + 7.5e-3f);
  }
  const float oldJ = Fn[0] * Fn[3] - Fn[1] * Fn[2];
  float Fo[4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) Fo[2 * i + j] = U[2 * i] * sg[0] * V[2 * j] + U[2 * i + 1] * sg[1] * V[2 * j + 1];
  const float newJ = Fo[0] * Fo[3] - Fo[1] * Fo[2];
  Jp[p] = fminf(fmaxf(Jp[p] * oldJ / newJ, 0.6f), 20.0f);
#pragma unroll
  for (int i = 0; i < 4; i++) F[4 * p + i] = Fo[i];
}

}  // namespace mpm88
