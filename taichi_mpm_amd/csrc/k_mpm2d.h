// taichi_mpm_amd/csrc/k_mpm2d.h — MPM<2>: the reference's 2D simulation (create_simulation2('mpm')), which runs the GENERIC
// transfer path (MPM<2>::rasterize_optimized = rasterize, MPM<2>::resample_optimized = resample: src/transfer.cpp:280-283,
// 697-700; bodies :193-278, :585-687) with all eight particle types in their dim = 2 form (src/particles.cpp).
// 2D scenes are the reference's small cases (tens of thousands to a million particles): dense (res+1)^2 grid of
// (m v.x, m v.y, m), one thread per particle / node, global float atomics for the scatter — the bandwidth design of the
// 3D path (blocks, tiles, records) is not needed here and is not repeated.  Part of libmpmhip (C ABI: mpmhip2d_*).
#pragma once
#include <hip/hip_runtime.h>

#include "mpm_math.h"
#include "k_rigid2d.h"

namespace mpm2d {

using mpm::GroupParams;
using mpm::LevelSetDev;

struct Params {
  int res[2];
  float dx, idx, dt, t;
  float g[2];
  int particle_gravity;
  float apic_damping, rpic_damping;
  int clean_boundary, particle_collision;
  int clamp_pos;  // generic path: positions clamped into [0, res - eps] (src/transfer.cpp:668-670)
  // MPM<2>::apply_dirichlet_boundary_conditions (src/mpm.cpp:374-399): nodes with x < dl move with (vl, 0), nodes with
  // x > 1 - dr with (vr, 0); off unless `dirichlet`
  int dirichlet;
  float dl, dr, vl, vr;
};

struct m2 {
  float a, b, c, d;  // [a b; c d]
};
__device__ __forceinline__ m2 mul(const m2 &x, const m2 &y) {
  return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
__device__ __forceinline__ float det(const m2 &x) { return x.a * x.d - x.b * x.c; }

// eigen-decomposition of the symmetric F F^T = U diag(lam) U^T by ONE Jacobi rotation (exact in 2D); U = [c s; -s c]^T form:
// columns (c, -s)... kept as (cu, su): U = [cu -su; su cu]
__device__ __forceinline__ void eig_FFt(const m2 &F, float &cu, float &su, float lam[2]) {
  float app = F.a * F.a + F.b * F.b, aqq = F.c * F.c + F.d * F.d, apq = F.a * F.c + F.b * F.d;
  const float d = aqq - app, x = apq + apq, xx = x * x;
  const float h = sqrtf(d * d + xx), den = fabsf(d) + h, n2 = den * den + xx;
  if (n2 > 1e-30f) {
    const float r = rsqrtf(n2);
    const float c = den * r, s = (d < 0.0f ? -1.0f : 1.0f) * x * r;
    const float sum = app + aqq, hs = d < 0.0f ? -h : h;
    lam[0] = 0.5f * (sum - hs); lam[1] = 0.5f * (sum + hs);
    // columns p, q of U rotated from the identity: (1,0) -> (c, ?) ...: U = G with G[p][p] = c, G[p][q] = s, G[q][p] = -s, G[q][q] = c
    cu = c; su = -s;  // U = [c s; -s c] = [cu -su; su cu]
  } else {
    lam[0] = app; lam[1] = aqq; cu = 1.0f; su = 0.0f;
  }
}
__device__ __forceinline__ void signed_sigma(const float lam[2], float detF, float s[2]) {
  s[0] = sqrtf(fmaxf(lam[0], 0.0f)); s[1] = sqrtf(fmaxf(lam[1], 0.0f));
  if (detF < 0.0f) { if (s[0] <= s[1]) s[0] = -s[0]; else s[1] = -s[1]; }
}
// U diag(e) U^T with U = [cu -su; su cu]
__device__ __forceinline__ m2 sandwich(float cu, float su, const float e[2]) {
  const float xx = cu * cu * e[0] + su * su * e[1], yy = su * su * e[0] + cu * cu * e[1], xy = cu * su * (e[0] - e[1]);
  return {xx, xy, xy, yy};
}

// calculate_force(): -vol * P(F) * F^T for dim = 2 (src/particles.cpp; same formulas as the 3D path with d = 2)
__device__ __forceinline__ m2 calculate_force(const GroupParams &g, const m2 &F, float aux) {
  const float vol = g.p[1];
  switch (g.type) {
    case MPMHIP_VISCO:
    case MPMHIP_JELLY:
    case MPMHIP_SNOW: {
      float mu = g.p[2], la = g.p[3];
      if (g.type == MPMHIP_SNOW) { const float e = expf(g.p[4] * (1.0f - aux)); mu *= e; la *= e; }
      float cu, su, lam[2], s[2];
      eig_FFt(F, cu, su, lam);
      const float J = det(F);
      signed_sigma(lam, J, s);
      const float vl = la * (J - 1.0f) * J;
      const float e[2] = {-vol * (2.0f * mu * (lam[0] - s[0]) + vl), -vol * (2.0f * mu * (lam[1] - s[1]) + vl)};
      return sandwich(cu, su, e);
    }
    case MPMHIP_LINEAR: {
      const float mu = g.p[2], la = g.p[3];
      const float tr = la * (F.a + F.d - 2.0f);
      const m2 P = {mu * (2.0f * F.a - 2.0f) + tr, mu * (F.b + F.c), mu * (F.b + F.c), mu * (2.0f * F.d - 2.0f) + tr};
      const m2 Ft = {F.a, F.c, F.b, F.d};
      m2 o = mul(P, Ft);
      o.a *= -vol; o.b *= -vol; o.c *= -vol; o.d *= -vol;
      return o;
    }
    case MPMHIP_WATER: {
      const float p = g.p[2] * (powf(aux, -g.p[3]) - 1.0f);
      const float dd = vol * aux * p;
      return {dd, 0.0f, 0.0f, dd};
    }
    default: {  // SAND, VON_MISES, ELASTIC: P F^T = U (2 mu ln S + lambda tr(ln S) I) U^T
      const float mu = g.p[2], la = g.p[3];
      float cu, su, lam[2], s[2];
      eig_FFt(F, cu, su, lam);
      signed_sigma(lam, det(F), s);
      const float l0 = logf(s[0]), l1 = logf(s[1]), tr = l0 + l1;
      const float e[2] = {-vol * (2.0f * mu * l0 + la * tr), -vol * (2.0f * mu * l1 + la * tr)};
      return sandwich(cu, su, e);
    }
  }
}

// plasticity(cdg) for dim = 2 (src/particles.cpp)
__device__ __forceinline__ void plasticity(const GroupParams &g, const m2 &cdg, m2 &F, float &aux) {
  if (g.type == MPMHIP_WATER) {  // :469-478  j *= tr(cdg) - (dim - 1)
    const float j = aux * (cdg.a + cdg.d - 1.0f);
    aux = j < 0.1f ? 0.1f : j;
    return;
  }
  if (g.type == MPMHIP_VISCO) {  // :87-134 with dim = 2
    const float mu = g.p[2], la = g.p[3], vnu = g.p[4], kappa = g.p[5], dt = g.p[6];
    float pnorm;
    {
      float cu, su, lam[2], s[2];
      eig_FFt(F, cu, su, lam);
      const float J0 = det(F);
      signed_sigma(lam, J0, s);
      const float p0 = 2.0f * mu * (s[0] - 1.0f) + la * (J0 - 1.0f) * J0 / s[0];
      const float p1 = 2.0f * mu * (s[1] - 1.0f) + la * (J0 - 1.0f) * J0 / s[1];
      pnorm = sqrtf(p0 * p0 + p1 * p1);
    }
    m2 sm = {cdg.a - 1.0f, cdg.b, cdg.c, cdg.d - 1.0f}, r;
    int halvings = 0;
    for (;;) {
      const m2 hm = {0.5f * sm.a + 1.0f, 0.5f * sm.b, 0.5f * sm.c, 0.5f * sm.d + 1.0f};
      r = mul(hm, sm);
      r.a += 1.0f; r.d += 1.0f;
      if (det(r) > 0.0f || halvings > 20) break;
      sm.a *= 0.5f; sm.b *= 0.5f; sm.c *= 0.5f; sm.d *= 0.5f;
      halvings++;
    }
    for (int i = 0; i < halvings; i++) r = mul(r, r);
    F = mul(r, F);
    float cu, su, lam[2], s[2];
    eig_FFt(F, cu, su, lam);
    signed_sigma(lam, det(F), s);
    float gamma = 0.0f;
    if (pnorm > 1e-5f) gamma = fminf(fmaxf(dt * vnu * (pnorm - aux) / pnorm, 0.0f), 1.0f);
    const float dets = s[0] * s[1];
    const float scale = fabsf(dets) > 1e-5f ? 1.0f / powf(dets, 0.5f) : 1.0f;
    float ratio[2];
    for (int k = 0; k < 2; k++) {
      const float md = powf(s[k] * scale, gamma);
      const float inv = fabsf(md) > 1e-5f ? 1.0f / md : 1.0f;
      ratio[k] = fminf(fmaxf(s[k] * inv, 0.1f), 10.0f) / s[k];
    }
    aux = aux + kappa * gamma * pnorm;
    F = mul(sandwich(cu, su, ratio), F);
    return;
  }
  F = mul(cdg, F);
  if (g.type == MPMHIP_JELLY || g.type == MPMHIP_LINEAR || g.type == MPMHIP_ELASTIC) return;
  float cu, su, lam[2], s[2];
  eig_FFt(F, cu, su, lam);
  signed_sigma(lam, det(F), s);
  float ratio[2];
  if (g.type == MPMHIP_SNOW) {  // :222-242
    const float lo = 1.0f - g.p[5], hi = 1.0f + g.p[6];
    float det_o = 1.0f, det_n = 1.0f;
    for (int i = 0; i < 2; i++) {
      const float c = fminf(fmaxf(s[i], lo), hi);
      det_o *= s[i]; det_n *= c;
      ratio[i] = c / s[i];
    }
    float Jp = aux * det_o / det_n;
    if (!(Jp <= g.p[8])) Jp = g.p[8];
    if (!(Jp >= g.p[7])) Jp = g.p[7];
    aux = Jp;
  } else if (g.type == MPMHIP_SAND) {  // :599-626, 639-647 with d = 2
    const float mu = g.p[2], la = g.p[3], alpha = g.p[4], coh = g.p[5], beta = g.p[6];
    const float e0 = logf(fmaxf(fabsf(s[0]), 1e-4f)) - coh, e1 = logf(fmaxf(fabsf(s[1]), 1e-4f)) - coh;
    const float sum = e0 + e1, tr = sum + aux;
    const float h0 = e0 - tr * 0.5f, h1 = e1 - tr * 0.5f;
    const float ehn = sqrtf(h0 * h0 + h1 * h1);
    float n0, n1;
    if (tr >= 0.0f) {
      n0 = n1 = expf(coh);
      aux = beta * sum + aux;
    } else {
      aux = 0.0f;
      const float dg = ehn + (2.0f * la + 2.0f * mu) / (2.0f * mu) * tr * alpha;
      const float k = (dg <= 0.0f) ? 0.0f : dg / ehn;
      n0 = expf(e0 - k * h0 + coh); n1 = expf(e1 - k * h1 + coh);
    }
    ratio[0] = n0 / s[0]; ratio[1] = n1 / s[1];
  } else {  // VON_MISES :713-732
    const float e0 = logf(s[0]), e1 = logf(s[1]), tr = e0 + e1;
    const float h0 = e0 - tr * 0.5f, h1 = e1 - tr * 0.5f;
    const float n2 = h0 * h0 + h1 * h1;
    const float dg = n2 - g.p[4] / (2.0f * g.p[2]);
    if (dg <= 0.0f) return;
    ratio[0] = expf(e0 - (dg / n2) * h0) / s[0];
    ratio[1] = expf(e1 - (dg / n2) * h1) / s[1];
  }
  F = mul(sandwich(cu, su, ratio), F);
}

__device__ __forceinline__ void weights(float rel, float w[3]) {  // MPMKernel<2,2>, src/kernel.h:103-135
  const float p = rel - 0.5f;
  const float t0 = p + 0.5f, t1 = p - 0.5f, t2 = p - 1.5f;
  w[0] = 0.5f * t0 * t0 - 1.5f * t0 + 1.125f;
  w[1] = -t1 * t1 + 0.75f;
  w[2] = 0.5f * t2 * t2 + 1.5f * t2 + 1.125f;
}

__device__ __forceinline__ bool alive_pos(const Params &P, const float x[2], const float v[2], int b[2]) {
  bool ok = isfinite(x[0]) && isfinite(x[1]) && isfinite(v[0]) && isfinite(v[1]);
  float X[2] = {x[0] * P.idx, x[1] * P.idx};
  if (P.clean_boundary) {  // near_boundary, src/mpm.h:269-276
    const float mn = fminf(X[0], X[1]), mx = fmaxf(X[0] - P.res[0], X[1] - P.res[1]);
    ok = ok && !(mn < 7.0f || mx > -7.0f);
  }
  for (int k = 0; k < 2; k++) {
    ok = ok && X[k] >= 0.5f;
    b[k] = ok ? (int)(X[k] - 0.5f) : 0;
    ok = ok && b[k] + 2 <= P.res[k];  // (the reference reads / writes beyond the grid here: undefined behaviour)
  }
  return ok;
}

__device__ __forceinline__ void friction_project2(float v[2], const float vb[2], const float n[2], float friction);

// rasterize — src/transfer.cpp:193-278 (with rigid bodies: the colour test and the impulses to the bodies)
__global__ __launch_bounds__(256) void k_p2g(Params P, int64_t n, const float *__restrict__ x, float *__restrict__ v,
                                             const float *__restrict__ F, const float *__restrict__ B,
                                             const float *__restrict__ aux, const int32_t *__restrict__ gid,
                                             const int32_t *__restrict__ pid, const GroupParams *__restrict__ groups,
                                             float *__restrict__ grid, RigidArgs2 R) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || pid[p] < 0) return;
  float vv[2] = {v[2 * p], v[2 * p + 1]};
  if (P.particle_gravity) {  // :202-204 (the particle keeps the kicked velocity until G2P overwrites it)
    vv[0] += P.g[0] * P.dt; vv[1] += P.g[1] * P.dt;
    v[2 * p] = vv[0]; v[2 * p + 1] = vv[1];
  }
  const float xx[2] = {x[2 * p], x[2 * p + 1]};
  int b[2];
  if (!alive_pos(P, xx, vv, b)) return;
  const GroupParams g = groups[gid[p]];
  const float mass = g.p[0];
  const m2 Fm = {F[4 * p], F[4 * p + 1], F[4 * p + 2], F[4 * p + 3]};
  const m2 st = calculate_force(g, Fm, aux[p]);
  const float S = -4.0f * P.idx * P.dt, m4 = 4.0f * mass;
  const float A[4] = {st.a * S + B[4 * p] * m4, st.b * S + B[4 * p + 1] * m4, st.c * S + B[4 * p + 2] * m4, st.d * S + B[4 * p + 3] * m4};
  const float r0 = xx[0] * P.idx - (float)b[0], r1 = xx[1] * P.idx - (float)b[1];
  float w0[3], w1[3];
  weights(r0, w0); weights(r1, w1);
  const int ny = P.res[1] + 1;
  uint32_t pstate = 0u;
  Bnd2 bn;
  bn.n[0] = bn.n[1] = 0.0f; bn.dist = 0.0f; bn.near = 0u;
  if (R.enabled) { pstate = R.states[p]; bn = R.bnd[p]; }
  int ibody = -1;  // the particle's impulses are summed here and handed over once (all impulses of a scene hit the same words)
  float isum[2] = {0, 0}, itrq = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float d0 = r0 - (float)i, d1 = r1 - (float)j, w = w0[i] * w1[j];
      if (R.enabled) {  // the colour test of the generic rasterize (:227-254): the other side of a body receives nothing
        const uint32_t word = node_word2(R, (size_t)(b[0] + i) * ny + (b[1] + j));
        if (incompatible2(word, pstate)) {
          const int rid = (int)(word >> 24) - 1;
          if (rid < 0) continue;
          Rigid2 *Bd = R.rb + rid;
          const float gpos[2] = {(b[0] + i) * P.dx, (b[1] + j) * P.dx};
          float rv[2], pv[2] = {vv[0], vv[1]};
          velocity_at2(*Bd, gpos, rv);
          friction_project2(pv, rv, bn.n, Bd->fric[(pstate >> (2 * rid)) & 1u]);
          // gradient of the weight (dw / dx, world units): d/dr of the quadratic B-spline is (r - 1.5, -2 (r - 1), r - 0.5)
          const float t0[3] = {r0 - 1.5f, -2.0f * (r0 - 1.0f), r0 - 0.5f}, t1[3] = {r1 - 1.5f, -2.0f * (r1 - 1.0f), r1 - 0.5f};
          const float gr[2] = {t0[i] * P.idx * w1[j], w0[i] * t1[j] * P.idx};
          const float imp[2] = {mass * w * (vv[0] - pv[0]) + P.dt * (st.a * gr[0] + st.b * gr[1]),
                                mass * w * (vv[1] - pv[1]) + P.dt * (st.c * gr[0] + st.d * gr[1])};
          if (ibody != rid) {
            if (ibody >= 0) { atomicAdd(&R.rb[ibody].tmp_imp[0], isum[0]); atomicAdd(&R.rb[ibody].tmp_imp[1], isum[1]); atomicAdd(&R.rb[ibody].tmp_trq, itrq); }
            ibody = rid; isum[0] = isum[1] = itrq = 0.0f;
          }
          isum[0] += imp[0]; isum[1] += imp[1];
          itrq += (gpos[0] - Bd->pos[0]) * imp[1] - (gpos[1] - Bd->pos[1]) * imp[0];
          continue;
        }
      }
      float *gp = grid + 3 * ((size_t)(b[0] + i) * ny + (b[1] + j));
      atomicAdd(gp + 0, w * (mass * vv[0] + A[0] * d0 + A[1] * d1));
      atomicAdd(gp + 1, w * (mass * vv[1] + A[2] * d0 + A[3] * d1));
      atomicAdd(gp + 2, w * mass);
    }
  if (ibody >= 0) { atomicAdd(&R.rb[ibody].tmp_imp[0], isum[0]); atomicAdd(&R.rb[ibody].tmp_imp[1], isum[1]); atomicAdd(&R.rb[ibody].tmp_trq, itrq); }
}

__device__ __forceinline__ void friction_project2(float v[2], const float vb[2], const float n[2], float friction) {
  if (friction == -1.0f) { v[0] = vb[0]; v[1] = vb[1]; return; }
  const bool slip = friction <= -2.0f;
  if (slip) friction = -friction - 2.0f;
  const float r0 = v[0] - vb[0], r1 = v[1] - vb[1];
  const float nn = n[0] * r0 + n[1] * r1;
  const float t0 = r0 - nn * n[0], t1 = r1 - nn * n[1];
  const float tn = sqrtf(t0 * t0 + t1 * t1);
  const float ts = fmaxf(tn + fminf(nn, 0.0f) * friction, 0.0f) / fmaxf(1e-30f, tn);
  const float keep = slip ? 0.0f : fmaxf(0.0f, nn);
  v[0] = ts * t0 + keep * n[0] + vb[0];
  v[1] = ts * t1 + keep * n[1] + vb[1];
}

// ---- MPM<2>::rigid_body_levelset_collision (src/mpm_rigid_body.cpp:347-387; see k_rigid_ls_keys / k_rigid_ls_collide of
// k_rigid.h for the 3D form and why the ORDER matters: the impulses of one boundary particle change the velocity the next
// one sees).  The order is the reference's sorted particle list: by the SPGrid key of the particle's base node
// (sort_particles_and_populate_grid, src/mpm.cpp:785-790; SPGrid_Mask<5, 5, 2>: 8 x 16-node blocks, page bits per level y
// above x, inside a block x above y), ties by the position in the previous order.
__device__ __forceinline__ uint32_t spread_every_second(uint32_t v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__device__ __forceinline__ uint32_t ref_node_key2(int i, int j) {
  const uint32_t blk = spread_every_second((uint32_t)(i >> 3)) | (spread_every_second((uint32_t)(j >> 4)) << 1);
  return (blk << 7) | (uint32_t)(((i & 7) << 4) | (j & 15));
}
__device__ __forceinline__ void sample_world2(const Rigid2 &B, const Sample2 &s, float p[2]) {
  rot2(B.angle, s.off, p);
  p[0] += B.pos[0]; p[1] += B.pos[1];
}
__global__ __launch_bounds__(256) void k2_ls_keys(float idx, const Rigid2 *__restrict__ rb, const Sample2 *__restrict__ smp, uint32_t n,
                                                  const uint32_t *__restrict__ rank, unsigned long long *__restrict__ keys,
                                                  uint32_t *__restrict__ vals) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    float p[2];
    sample_world2(rb[smp[s].body], smp[s], p);
    const int b0 = max((int)(p[0] * idx - 0.5f), 0), b1 = max((int)(p[1] * idx - 0.5f), 0);  // get_grid_base_pos (src/mpm.h:252-255)
    keys[s] = ((unsigned long long)ref_node_key2(b0, b1) << 32) | rank[s];
    vals[s] = s;
  }
}
struct Restitution2 { float e[MAX_RIGID2]; };
// one workgroup: batches of 1024 boundary particles are tested against the level set in parallel, the hits of a batch are
// compacted IN ORDER, and one lane applies their impulses to the bodies (kept in LDS) one after the other
__global__ __launch_bounds__(1024) void k2_ls_collide(Params P, LevelSetDev LS, Rigid2 *rb, int nb, const Sample2 *__restrict__ smp,
                                                      const uint32_t *__restrict__ sorted, uint32_t n, uint32_t *__restrict__ rank,
                                                      Restitution2 rest) {
  struct Hit { int body; float r[2], g[2]; };
  __shared__ Rigid2 sb[MAX_RIGID2];
  __shared__ Hit hits[1024];
  __shared__ int wave_n[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < nb) sb[t] = rb[t];
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t j = base + t;
    bool hit = false;
    Hit h;
    h.body = 0;
    if (j < n) {
      const uint32_t s = sorted[j];
      rank[s] = j;  // the position in this substep's order is the next substep's tie-break
      const Sample2 sm = smp[s];
      float p[2];
      sample_world2(rb[sm.body], sm, p);
      const float xw[3] = {p[0], p[1], 0.0f};
      float phi, g[3] = {0, 0, 0};
      if (mpm::levelset_eval(LS, P.t, xw, P.idx, phi, g) && phi < 0.0f) {
        hit = true;
        h.body = sm.body;
        h.r[0] = p[0] - rb[sm.body].pos[0]; h.r[1] = p[1] - rb[sm.body].pos[1];
        h.g[0] = g[0]; h.g[1] = g[1];
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_n[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) off += wave_n[w]; total += wave_n[w]; }
    if (hit) hits[off + __popcll(m & ((1ull << lane) - 1ull))] = h;
    __syncthreads();
    if (t == 0) {
      for (int e = 0; e < total; e++) {
        const Hit &H = hits[e];
        Rigid2 &B = sb[H.body];
        auto vel_at = [&](float v[2]) { v[0] = B.vel[0] - B.omega * H.r[1]; v[1] = B.vel[1] + B.omega * H.r[0]; };
        auto contribution = [&](const float d[2]) {  // get_impulse_contribution: inv_mass + inv_inertia * cross(r, d)^2
          const float c = H.r[0] * d[1] - H.r[1] * d[0];
          return B.inv_mass + B.inv_I * c * c;
        };
        auto apply = [&](const float imp[2]) {
          B.vel[0] += imp[0] * B.inv_mass; B.vel[1] += imp[1] * B.inv_mass;
          B.omega += B.inv_I * (H.r[0] * imp[1] - H.r[1] * imp[0]);
        };
        float v10[2];
        vel_at(v10);
        const float v0 = H.g[0] * v10[0] + H.g[1] * v10[1];
        const float J = -((1.0f + rest.e[H.body]) * v0) * (1.0f / contribution(H.g));
        if (!(J >= 0.0f)) continue;  // (J < 0: separating; a body of infinite mass and inertia gives 0 / 0)
        const float imp[2] = {J * H.g[0], J * H.g[1]};
        apply(imp);
        vel_at(v10);
        const float vn = H.g[0] * v10[0] + H.g[1] * v10[1];
        float tao[2] = {v10[0] - H.g[0] * vn, v10[1] - H.g[1] * vn};
        if (fmaxf(fabsf(tao[0]), fabsf(tao[1])) > 1e-7f) {
          const float il = 1.0f / sqrtf(tao[0] * tao[0] + tao[1] * tao[1]);
          tao[0] *= il; tao[1] *= il;
          const float fr = B.fric[0];
          float jj = -(v10[0] * tao[0] + v10[1] * tao[1]) / contribution(tao);
          jj = fminf(fmaxf(jj, fr * -J), fr * J);
          const float fi[2] = {jj * tao[0], jj * tao[1]};
          apply(fi);
        }
      }
    }
    __syncthreads();
  }
  if (t >= 1 && t < nb) { rb[t].vel[0] = sb[t].vel[0]; rb[t].vel[1] = sb[t].vel[1]; rb[t].omega = sb[t].omega; }
}

// normalize_grid_and_apply_external_force + apply_grid_boundary_conditions — src/mpm.cpp:277-372
__global__ __launch_bounds__(256) void k_grid(Params P, LevelSetDev LS, float *__restrict__ grid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ny = P.res[1] + 1;
  if (t >= (P.res[0] + 1) * ny) return;
  float *g = grid + 3 * (size_t)t;
  const float m = g[2];
  float v[2] = {g[0], g[1]};
  if (m > 0.0f) {
    const float im = 1.0f / m;
    v[0] = v[0] * im + (P.particle_gravity ? 0.0f : P.g[0] * P.dt);
    v[1] = v[1] * im + (P.particle_gravity ? 0.0f : P.g[1] * P.dt);
  }
  if (m != 0.0f && LS.n > 0) {
    const float xw[3] = {(float)(t / ny) * P.dx, (float)(t % ny) * P.dx, 0.0f};
    float phi, dphidt, nrm[3] = {0, 0, 0};
    mpm::levelset_eval(LS, P.t, xw, P.idx, phi, nrm, &dphidt);
    if (!(phi < -3.0f || 0.0f < phi)) {
      const float vb[2] = {-dphidt * nrm[0] * P.dx, -dphidt * nrm[1] * P.dx};
      friction_project2(v, vb, nrm, LS.friction);
    }
  }
  if (P.dirichlet) {  // src/mpm.cpp:389-397, behind the boundary condition (:541-544); every node of the grid region
    const float px = (float)(t / ny) * P.dx;
    if (px < P.dl) { v[0] = P.vl; v[1] = 0.0f; }
    else if (px > 1.0f - P.dr) { v[0] = P.vr; v[1] = 0.0f; }
  }
  g[0] = v[0]; g[1] = v[1];
}

// resample — src/transfer.cpp:585-687 (same-colour branch), then clear_boundary_particles (src/mpm.cpp:582-633) and
// particle_collision_resolution (:414-426)
__global__ __launch_bounds__(256) void k_g2p(Params P, LevelSetDev LS, int64_t n, float *__restrict__ x, float *__restrict__ v,
                                             float *__restrict__ F, float *__restrict__ B, float *__restrict__ aux,
                                             const int32_t *__restrict__ gid, int32_t *__restrict__ pid,
                                             const GroupParams *__restrict__ groups, const float *__restrict__ grid,
                                             unsigned int *__restrict__ n_dead, RigidArgs2 R) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || pid[p] < 0) return;
  float xx[2] = {x[2 * p], x[2 * p + 1]}, vv[2] = {v[2 * p], v[2 * p + 1]};
  int b[2];
  if (!alive_pos(P, xx, vv, b)) {  // (only reachable for freshly uploaded particles: deleted like the reference's clean-up)
    pid[p] = -1;
    atomicAdd(n_dead, 1u);
    return;
  }
  const float r0 = xx[0] * P.idx - (float)b[0], r1 = xx[1] * P.idx - (float)b[1];
  float w0[3], w1[3];
  weights(r0, w0); weights(r1, w1);
  const int ny = P.res[1] + 1;
  float nv[2] = {0, 0}, bb[4] = {0, 0, 0, 0};
  uint32_t pstate = 0u;
  Bnd2 bn;
  bn.n[0] = bn.n[1] = 0.0f; bn.dist = 0.0f; bn.near = 0u;
  int rigid_id = -1;
  if (R.enabled) { pstate = R.states[p]; bn = R.bnd[p]; }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float *gp = grid + 3 * ((size_t)(b[0] + i) * ny + (b[1] + j));
      const float d0 = r0 - (float)i, d1 = r1 - (float)j, w = w0[i] * w1[j];
      float gv[2] = {gp[0], gp[1]};
      if (R.enabled) {  // :611-642: a node of the other colour contributes the particle's own (projected) velocity
        const uint32_t word = node_word2(R, (size_t)(b[0] + i) * ny + (b[1] + j));
        if (incompatible2(word, pstate)) {
          float fake[2] = {vv[0], vv[1]}, vg[2] = {0, 0}, friction = 0.0f;
          const int rid = (int)(word >> 24) - 1;
          if (rid >= 0) {
            const float gpos[2] = {(b[0] + i) * P.dx, (b[1] + j) * P.dx};
            velocity_at2(R.rb[rid], gpos, vg);
            rigid_id = rid;
            friction = R.rb[rid].fric[(pstate >> (2 * rid)) & 1u];
          }
          if (bn.near) {
            friction_project2(fake, vg, bn.n, friction);
            const float push = P.dt * P.dx * R.pushing_force;
            fake[0] += bn.n[0] * push; fake[1] += bn.n[1] * push;
          }
          gv[0] = fake[0]; gv[1] = fake[1];
        }
      }
      const float a0 = w * gv[0], a1 = w * gv[1];
      nv[0] += a0; nv[1] += a1;
      bb[0] += a0 * d0; bb[1] += a0 * d1; bb[2] += a1 * d0; bb[3] += a1 * d1;  // b = sum (w g_v) (x) dpos, :644
    }
  const float scale = -4.0f * P.idx * P.dt;
  const m2 cdg = {1.0f + scale * bb[0], scale * bb[1], scale * bb[2], 1.0f + scale * bb[3]};  // :659-661
  if (bn.near) {  // p.apic_b = Matrix(0), :649-653
    bb[0] = bb[1] = bb[2] = bb[3] = 0.0f;
  } else if (P.rpic_damping != 0.0f || P.apic_damping != 0.0f) {  // damp_affine_momemtum, src/mpm.h:465-469 (:654)
    const float ks = 1.0f - P.rpic_damping, ka = 1.0f - P.apic_damping;
    const float sym = 0.5f * (bb[1] + bb[2]), skew = 0.5f * (bb[1] - bb[2]);
    bb[0] *= ks; bb[3] *= ks;
    bb[1] = ks * sym + ka * skew; bb[2] = ks * sym - ka * skew;
  }
  const GroupParams g = groups[gid[p]];
  m2 Fm = {F[4 * p], F[4 * p + 1], F[4 * p + 2], F[4 * p + 3]};
  float a = aux[p];
  plasticity(g, cdg, Fm, a);
  xx[0] += P.dt * nv[0]; xx[1] += P.dt * nv[1];  // :664
  if (P.clamp_pos) {  // :668-670
    xx[0] = fminf(fmaxf(xx[0] * P.idx, 0.0f), (float)P.res[0] - 1e-6f) * P.dx;
    xx[1] = fminf(fmaxf(xx[1] * P.idx, 0.0f), (float)P.res[1] - 1e-6f) * P.dx;
  }
  if (bn.near && bn.dist < -0.05f * P.dx && bn.dist > -P.dx * 0.3f) {  // penalty, :671-682
    const float dv[2] = {bn.dist * bn.n[0] * R.penalty, bn.dist * bn.n[1] * R.penalty};
    nv[0] -= dv[0]; nv[1] -= dv[1];
    if (rigid_id != -1) {
      const float imp[2] = {dv[0] * g.p[0], dv[1] * g.p[0]};
      tmp_impulse2(R.rb + rigid_id, imp, xx);
    }
  }
  int nb[2];
  const bool keep = alive_pos(P, xx, nv, nb);  // clear_boundary_particles sees the advected position
  if (keep && P.particle_collision && LS.n > 0) {
    const float xw[3] = {xx[0], xx[1], 0.0f};
    float phi, gr[3] = {0, 0, 0};
    if (mpm::levelset_eval(LS, P.t, xw, P.idx, phi, gr) && phi < 0.0f) {
      const float vn = gr[0] * nv[0] + gr[1] * nv[1];
      xx[0] -= gr[0] * phi * P.dx; xx[1] -= gr[1] * phi * P.dx;
      nv[0] -= vn * gr[0]; nv[1] -= vn * gr[1];
    }
  }
  x[2 * p] = xx[0]; x[2 * p + 1] = xx[1];
  v[2 * p] = nv[0]; v[2 * p + 1] = nv[1];
#pragma unroll
  for (int k = 0; k < 4; k++) B[4 * p + k] = bb[k];
  if (g.type != MPMHIP_WATER) { F[4 * p] = Fm.a; F[4 * p + 1] = Fm.b; F[4 * p + 2] = Fm.c; F[4 * p + 3] = Fm.d; }
  aux[p] = a;
  if (!keep) {
    pid[p] = -1;
    atomicAdd(n_dead, 1u);
  }
}

}  // namespace mpm2d
