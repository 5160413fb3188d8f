// oracle/mpm_oracle.cpp — CPU oracle for the MLS-MPM sub-step.
//
// TEST INFRASTRUCTURE ONLY (see mpm_oracle.h).  A restatement, in plain fp32
// C++17, of the algorithm in yuanming-hu/taichi_mpm; every function cites the
// reference `file:line` it follows.  PARITY UNPINNED except for the kernel
// weights (reference has no golden vectors and cannot be built here).
//
// Conventions: 3x3 matrices are row-major float[9] (m[3*r+c]); the reference
// stores column-major `Matrix[i]` = column i (README.md:314) — every formula
// below is written in index-free matrix algebra so the storage order is
// immaterial.

#include "mpm_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_core.h"
using namespace orc;

// ============================================================================ C ABI
extern "C" {

void orc_kernel3_dw_w(const float pos[3], float inv_dx, float out[27 * 4]) {
  // MPMFastKernel32 ctor — src/kernel.h:174-188: kernels[i][j][k] = ((1*ws[0][i])*ws[1][j])*ws[2][k]
  real w[3][3], dw[3][3];
  for (int d = 0; d < 3; d++) quad_w_dw(fract(pos[d] - 0.5f), w[d], dw[d]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) {
        float *o = out + 4 * (i * 9 + j * 3 + k);
        // w_stages[axis][k][comp] = (comp==axis) ? dw*inv_dx : w   (kernel.h:29-42)
        for (int comp = 0; comp < 4; comp++) {
          real a = (comp == 0) ? dw[0][i] * inv_dx : w[0][i];
          real b = (comp == 1) ? dw[1][j] * inv_dx : w[1][j];
          real cc = (comp == 2) ? dw[2][k] * inv_dx : w[2][k];
          o[comp] = ((1.0f * a) * b) * cc;
        }
      }
}

void orc_kernel3_dw_w_slow(const float pos[3], float inv_dx, float out[27 * 4]) {
  // MPMKernelBase::get_dw_w — src/kernel.h:44-50: ret = ws[0][k0]; ret *= ws[1][k1]; ret *= ws[2][k2]
  real w[3][3], dw[3][3];
  for (int d = 0; d < 3; d++) quad_w_dw(fract(pos[d] - 0.5f), w[d], dw[d]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) {
        float *o = out + 4 * (i * 9 + j * 3 + k);
        for (int comp = 0; comp < 4; comp++) {
          real ret = (comp == 0) ? dw[0][i] * inv_dx : w[0][i];
          ret *= (comp == 1) ? dw[1][j] * inv_dx : w[1][j];
          ret *= (comp == 2) ? dw[2][k] * inv_dx : w[2][k];
          o[comp] = ret;
        }
      }
}

void orc_mls_kernel3_w(const float rel_pos[3], float out[27]) {
  // MLSMPMFastKernel32 — src/transfer.cpp:168-186 (no fract: the caller passes pos - base cell)
  real w[3][3];
  for (int d = 0; d < 3; d++) quad_w_fma(rel_pos[d] - 0.5f, w[d]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) out[i * 9 + j * 3 + k] = (w[0][i] * w[1][j]) * w[2][k];
}

void orc_kernel2_dw_w(const float pos[2], float inv_dx, float out[9 * 3]) {
  real w[2][3], dw[2][3];
  for (int d = 0; d < 2; d++) quad_w_dw(fract(pos[d] - 0.5f), w[d], dw[d]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float *o = out + 3 * (i * 3 + j);
      for (int comp = 0; comp < 3; comp++) {
        real ret = (comp == 0) ? dw[0][i] * inv_dx : w[0][i];
        ret *= (comp == 1) ? dw[1][j] * inv_dx : w[1][j];
        o[comp] = ret;
      }
    }
}

void orc_kernel_cubic_dw_w(int dim, const float *pos, float inv_dx, float *out) {
  // MPMKernel<dim,3>::calculate_kernel — src/kernel.h:152-164
  real w[3][4], dw[3][4];
  const real a3[4] = {-1 / 6.0f, 0.5f, -0.5f, 1 / 6.0f}, a2[4] = {1, -1, -1, 1}, a1[4] = {-2, 0, 0, 2},
             a0[4] = {4 / 3.0f, 2 / 3.0f, 2 / 3.0f, 4 / 3.0f};
  const real b2[4] = {-0.5f, 1.5f, -1.5f, 0.5f}, b1[4] = {2, -2, -2, 2}, b0[4] = {-2, 0, 0, 2};
  const real off[4] = {-1, 0, 1, 2};
  for (int d = 0; d < dim; d++) {
    real pf = fract(pos[d]);
    for (int k = 0; k < 4; k++) {
      real t = pf - off[k], tt = t * t, ttt = tt * t;
      w[d][k] = a3[k] * ttt + a2[k] * tt + a1[k] * t + a0[k];
      dw[d][k] = b2[k] * tt + b1[k] * t + b0[k];
    }
  }
  int total = dim == 2 ? 16 : 64;
  for (int id = 0; id < total; id++) {
    int idx[3] = {dim == 2 ? id / 4 : id / 16, dim == 2 ? id % 4 : (id / 4) % 4, id % 4};
    for (int comp = 0; comp <= dim; comp++) {
      real ret = 1;
      for (int d = 0; d < dim; d++) ret *= (comp == d) ? dw[d][idx[d]] * inv_dx : w[d][idx[d]];
      out[id * (dim + 1) + comp] = ret;
    }
  }
}

void orc_svd3(const float F[9], float U[9], float S[3], float V[9]) {
  M3 u, s, v;
  svd(load3(F), u, s, v);
  store3(u, U); store3(v, V);
  S[0] = s(0,0); S[1] = s(1,1); S[2] = s(2,2);
}
void orc_polar3(const float F[9], float R[9], float Ssym[9]) {
  M3 r, s;
  polar_decomp(load3(F), r, s);
  store3(r, R); store3(s, Ssym);
}
void orc_svd2(const float F[4], float U[4], float S[2], float V[4]) {
  double A[2][2] = {{F[0], F[1]}, {F[2], F[3]}}, Ud[2][2], s[2], Vd[2][2];
  svd2_d(A, Ud, s, Vd);
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { U[2 * i + j] = (float)Ud[i][j]; V[2 * i + j] = (float)Vd[i][j]; }
  S[0] = (float)s[0]; S[1] = (float)s[1];
}
void orc_polar2(const float F[4], float R[4], float Ssym[4]) {
  double A[2][2] = {{F[0], F[1]}, {F[2], F[3]}}, Rd[2][2], Sd[2][2];
  polar2_d(A, Rd, Sd);
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { R[2 * i + j] = (float)Rd[i][j]; Ssym[2 * i + j] = (float)Sd[i][j]; }
}

void orc_calculate_force(int type, const float *gp, const float F[9], float aux, float out[9]) {
  store3(calculate_force(type, gp, load3(F), aux), out);
}
void orc_plasticity(int type, const float *gp, const float cdg[9], float F[9], float *aux) {
  M3 f = load3(F);
  real a = *aux;
  plasticity(type, gp, load3(cdg), f, a);
  store3(f, F);
  *aux = a;
}
void orc_friction_project(const float v[3], const float vb[3], const float n[3], float mu, float out[3]) {
  friction_project(v, vb, n, mu, out);
}

// ---------------------------------------------------------------------------- P2G
// MPM<3>::rasterize_optimized / block_op_normal — src/transfer.cpp:467-569.
// Per particle, in the reference's operation order:
//   v += gravity*dt                                      (:485-487, particle_gravity)
//   pos_ = pos * inv_dx ; rela_pos = pos_ - base          (:490,518)
//   weights = MLSMPMFastKernel32(rela_pos)                (:493)
//   affine[i] = fma(stress[i], S, apic_b[i]*(4*mass)), S = -4*inv_dx*dt   (:465,507,521-522)
//   per node: dpos = rela_pos - offset; contrib = (fma chain affine*dpos + mass*v, mass);
//             g += weight * contrib                       (:526-545)
// Particles are visited in array order (the reference visits them in sorted
// block/cell order; only the fp32 summation order differs).
void orc_p2g(const orc_config *c, int64_t n, const float *x, float *v, const float *B, const float *F,
             const float *aux, const int32_t *gid, const float *gparams, const int32_t *gtype, float *grid) {
  const real idx = 1.0f / c->dx, dt = c->dt;
  const real S = -4.0f * idx * dt;
  const int64_t nn = (int64_t)(c->res[0] + 1) * (c->res[1] + 1) * (c->res[2] + 1);
  std::memset(grid, 0, sizeof(float) * 4 * nn);  // src/mpm.cpp:867-874
  for (int64_t p = 0; p < n; p++) {
    if (!particle_alive(c, x + 3 * p, v + 3 * p)) continue;
    const float *gp = gparams + ORC_NPARAM * gid[p];
    const int type = gtype[gid[p]];
    if (c->particle_gravity)
      for (int k = 0; k < 3; k++) v[3 * p + k] = v[3 * p + k] + c->gravity[k] * dt;
    real pos[3], rela[3];
    int base[3];
    for (int k = 0; k < 3; k++) {
      pos[k] = x[3 * p + k] * idx;
      base[k] = stencil_start(pos[k]);
      rela[k] = pos[k] - (real)base[k];
    }
    real w[3][3];
    for (int d = 0; d < 3; d++) quad_w_fma(rela[d] - 0.5f, w[d]);
    const real mass = gp[0];
    M3 stress = calculate_force(type, gp, load3(F + 9 * p), aux[p]);
    M3 Bm = load3(B + 9 * p);
    M3 affine;
    for (int i = 0; i < 9; i++) affine.a[i] = std::fmaf(stress.a[i], S, Bm.a[i] * (4.0f * mass));
    real mass_v[3] = {mass * v[3 * p], mass * v[3 * p + 1], mass * v[3 * p + 2]};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) {
          real dpos[3] = {rela[0] - i, rela[1] - j, rela[2] - k};
          real weight = (w[0][i] * w[1][j]) * w[2][k];
          float *g = grid + 4 * node_index(c, base[0] + i, base[1] + j, base[2] + k);
          for (int r = 0; r < 3; r++) {
            // fmadd(affine[2],d2, fmadd(affine[1],d1, fmadd(affine[0],d0, mass_v)))  (columns of affine)
            real ap = std::fmaf(affine(r, 2), dpos[2], std::fmaf(affine(r, 1), dpos[1], std::fmaf(affine(r, 0), dpos[0], mass_v[r])));
            g[r] = g[r] + weight * ap;
          }
          g[3] = g[3] + weight * mass;
        }
  }
}

// ---------------------------------------------------------------------------- grid
// normalize_grid_and_apply_external_force — src/mpm.cpp:277-294 (velocity increment is
// gravity*dt only when !particle_gravity, :526-530), then
// apply_grid_boundary_conditions — src/mpm.cpp:296-372 with an analytic level set.
void orc_grid_update(const orc_config *c, float *grid) {
  real inc[3] = {0, 0, 0};
  if (!c->particle_gravity) for (int k = 0; k < 3; k++) inc[k] = c->gravity[k] * c->dt;
  for (int i = 0; i <= c->res[0]; i++)
    for (int j = 0; j <= c->res[1]; j++)
      for (int k = 0; k <= c->res[2]; k++) {
        float *g = grid + 4 * node_index(c, i, j, k);
        real mass = g[3];
        if (mass > 0) {
          real inv_mass = 1.0f / mass;
          for (int r = 0; r < 3; r++) g[r] = std::fmaf(g[r], inv_mass, inc[r]);
        }
        if (g[3] == 0.0f) continue;  // src/mpm.cpp:313-315
        real pos[3] = {(real)i, (real)j, (real)k}, phi, nrm[3] = {0, 0, 0};
        real dphidt;
        if (!levelset_eval(c, pos, phi, nrm, &dphidt)) continue;
        if (phi < -3 || 0 < phi) continue;  // src/mpm.cpp:324-325
        // boundary_velocity = -levelset.get_temporal_derivative(pos, t) * n * delta_x   (src/mpm.cpp:340-342)
        real vb[3] = {-dphidt * nrm[0] * c->dx, -dphidt * nrm[1] * c->dx, -dphidt * nrm[2] * c->dx}, out[3];
        real vel[3] = {g[0], g[1], g[2]};
        friction_project(vel, vb, nrm, c->friction, out);
        g[0] = out[0]; g[1] = out[1]; g[2] = out[2];
      }
}

// ---------------------------------------------------------------------------- G2P
// MPM<3>::resample_optimized / block_op_normal — src/transfer.cpp:837-954.
//   v_ = Σ fma(grid_vel, w, v_) ; b_[r] = Σ fma(w*grid_vel, dpos[r], b_[r])       (:888-904)
//   apic_b = damp(b)  (reference bug on this path documented in DESIGN.md; :925-931, mpm.h:465-469)
//   cdg = I + (-4*inv_dx*dt) * b                                                   (:936-942)
//   plasticity(cdg) ; pos += v_*dt                                                 (:950-951)
void orc_g2p(const orc_config *c, int64_t n, float *x, float *v, float *B, float *F, float *aux,
             const int32_t *gid, const float *gparams, const int32_t *gtype, const float *grid) {
  const real idx = 1.0f / c->dx, dt = c->dt;
  const real scale = -4.0f * idx * dt;
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < n; p++) {
    if (!particle_alive(c, x + 3 * p, v + 3 * p)) continue;
    const float *gp = gparams + ORC_NPARAM * gid[p];
    const int type = gtype[gid[p]];
    real pos[3], rela[3];
    int base[3];
    for (int k = 0; k < 3; k++) {
      pos[k] = x[3 * p + k] * idx;
      base[k] = stencil_start(pos[k]);
      rela[k] = pos[k] - (real)base[k];
    }
    real w[3][3];
    for (int d = 0; d < 3; d++) quad_w_fma(rela[d] - 0.5f, w[d]);
    real v_[3] = {0, 0, 0};
    M3 b_ = m3_zero();
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) {
          real dpos[3] = {rela[0] - i, rela[1] - j, rela[2] - k};
          real weight = (w[0][i] * w[1][j]) * w[2][k];
          const float *g = grid + 4 * node_index(c, base[0] + i, base[1] + j, base[2] + k);
          for (int r = 0; r < 3; r++) {
            v_[r] = std::fmaf(g[r], weight, v_[r]);
            real wgv = weight * g[r];
            for (int cc = 0; cc < 3; cc++) b_(r, cc) = std::fmaf(wgv, dpos[cc], b_(r, cc));
          }
        }
    M3 bd = b_;
    if (c->rpic_damping != 0 || c->apic_damping != 0) {  // damp_affine_momemtum — src/mpm.h:465-469
      M3 sym = 0.5f * (b_ + transposed(b_));
      M3 skew = b_ - sym;
      bd = (1 - c->rpic_damping) * sym + (1 - c->apic_damping) * skew;
    }
    store3(bd, B + 9 * p);
    for (int k = 0; k < 3; k++) v[3 * p + k] = v_[k];
    M3 cdg;
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) cdg(r, cc) = std::fmaf(scale, b_(r, cc), (r == cc) ? 1.0f : 0.0f);
    M3 Fm = load3(F + 9 * p);
    real a = aux[p];
    plasticity(type, gp, cdg, Fm, a);
    store3(Fm, F + 9 * p);
    aux[p] = a;
    for (int k = 0; k < 3; k++) x[3 * p + k] = std::fmaf(v_[k], dt, x[3 * p + k]);
  }
}

// MPM<dim>::particle_collision_resolution — src/mpm.cpp:414-426: a particle inside the level set (phi < 0) is
// moved back to the surface along the gradient and loses the normal component of its velocity.
void orc_particle_collision(const orc_config *c, int64_t n, float *x, float *v) {
  const real idx = 1.0f / c->dx;
  for (int64_t p = 0; p < n; p++) {
    real pos[3] = {x[3 * p] * idx, x[3 * p + 1] * idx, x[3 * p + 2] * idx}, phi, g[3] = {0, 0, 0};
    if (!levelset_eval(c, pos, phi, g) || !(phi < 0)) continue;
    real vn = g[0] * v[3 * p] + g[1] * v[3 * p + 1] + g[2] * v[3 * p + 2];
    for (int k = 0; k < 3; k++) {
      x[3 * p + k] -= g[k] * phi * c->dx;
      v[3 * p + k] -= vn * g[k];
    }
  }
}

// general_action "delete_particles_inside_level_set" — src/mpm.cpp:962-974: a particle whose level-set sample at its
// position is negative is dropped.  keep[p] = 0 for those; returns the number deleted.
int64_t orc_delete_inside_levelset(const orc_config *c, int64_t n, const float *x, uint8_t *keep) {
  const real idx = 1.0f / c->dx;
  int64_t deleted = 0;
  for (int64_t p = 0; p < n; p++) {
    real pos[3] = {x[3 * p] * idx, x[3 * p + 1] * idx, x[3 * p + 2] * idx}, phi, g[3] = {0, 0, 0};
    const bool inside = levelset_eval(c, pos, phi, g) && phi < 0;
    keep[p] = inside ? 0 : 1;
    deleted += inside;
  }
  return deleted;
}

int64_t orc_clear_boundary(const orc_config *c, int64_t n, const float *x, const float *v, uint8_t *keep) {
  int64_t cnt = 0;
  for (int64_t p = 0; p < n; p++) {
    keep[p] = particle_alive(c, x + 3 * p, v + 3 * p) ? 1 : 0;
    cnt += keep[p];
  }
  return cnt;
}

// MPM<dim>::substep — src/mpm.cpp:452-575 (no rigid bodies): [sort] P2G, grid, G2P, clean.
int64_t orc_substep(const orc_config *c, int64_t n, float *x, float *v, float *B, float *F, float *aux,
                    int32_t *gid, int32_t *ids, const float *gparams, const int32_t *gtype, float *grid) {
  orc_p2g(c, n, x, v, B, F, aux, gid, gparams, gtype, grid);
  orc_grid_update(c, grid);
  orc_g2p(c, n, x, v, B, F, aux, gid, gparams, gtype, grid);
  // clear_boundary_particles (src/mpm.cpp:582-633): stable compaction
  int64_t m = 0;
  for (int64_t p = 0; p < n; p++) {
    if (!particle_alive(c, x + 3 * p, v + 3 * p)) continue;
    if (m != p) {
      std::memcpy(x + 3 * m, x + 3 * p, 12); std::memcpy(v + 3 * m, v + 3 * p, 12);
      std::memcpy(B + 9 * m, B + 9 * p, 36); std::memcpy(F + 9 * m, F + 9 * p, 36);
      aux[m] = aux[p]; gid[m] = gid[p];
      if (ids) ids[m] = ids[p];
    }
    m++;
  }
  if (c->particle_collision) orc_particle_collision(c, m, x, v);  // src/mpm.cpp:566-569 (after the clean-up)
  return m;
}

// ---------------------------------------------------------------------------- 2D demo
// advance(dt) — mls-mpm88.cpp:16-69 (n = grid cells; grid is (n+1)^2 Vector3).
void orc_mpm88_advance(int n_grid, float dt, int64_t n, float *x, float *v, float *F, float *C, float *Jp,
                       float *grid, int plastic) {
  const int ng = n_grid;
  const real dx = 1.0f / ng, inv_dx = 1.0f / dx;
  const real particle_mass = 1.0f, vol = 1.0f, hardening = 10.0f, E = 1e4f, nu = 0.2f;  // mls-mpm88.cpp:7-8
  const real mu_0 = E / (2 * (1 + nu)), lambda_0 = E * nu / ((1 + nu) * (1 - 2 * nu));
  std::memset(grid, 0, sizeof(float) * 3 * (ng + 1) * (ng + 1));
  auto G = [&](int i, int j) { return grid + 3 * (i * (ng + 1) + j); };
  for (int64_t p = 0; p < n; p++) {  // P2G — mls-mpm88.cpp:18-36
    int bc[2] = {int(x[2 * p] * inv_dx - 0.5f), int(x[2 * p + 1] * inv_dx - 0.5f)};
    real fx[2] = {x[2 * p] * inv_dx - bc[0], x[2 * p + 1] * inv_dx - bc[1]};
    real w[3][2];
    for (int d = 0; d < 2; d++) {
      w[0][d] = 0.5f * (1.5f - fx[d]) * (1.5f - fx[d]);
      w[1][d] = 0.75f - (fx[d] - 1.0f) * (fx[d] - 1.0f);
      w[2][d] = 0.5f * (fx[d] - 0.5f) * (fx[d] - 0.5f);
    }
    real e = std::exp(hardening * (1.0f - Jp[p])), mu = mu_0 * e, lambda = lambda_0 * e;
    const float *f = F + 4 * p;
    real J = f[0] * f[3] - f[1] * f[2];
    double A[2][2] = {{f[0], f[1]}, {f[2], f[3]}}, Rd[2][2], Sd[2][2];
    polar2_d(A, Rd, Sd);
    real r[4] = {(real)Rd[0][0], (real)Rd[0][1], (real)Rd[1][0], (real)Rd[1][1]};
    real fr[4] = {f[0] - r[0], f[1] - r[1], f[2] - r[2], f[3] - r[3]};
    // (F-r)*F^T
    real m[4] = {fr[0] * f[0] + fr[1] * f[1], fr[0] * f[2] + fr[1] * f[3], fr[2] * f[0] + fr[3] * f[1], fr[2] * f[2] + fr[3] * f[3]};
    real k = -4 * inv_dx * inv_dx * dt * vol;
    real stress[4] = {k * (2 * mu * m[0] + lambda * (J - 1) * J), k * (2 * mu * m[1]), k * (2 * mu * m[2]),
                      k * (2 * mu * m[3] + lambda * (J - 1) * J)};
    real affine[4];
    for (int i = 0; i < 4; i++) affine[i] = stress[i] + particle_mass * C[4 * p + i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        real dpos[2] = {(i - fx[0]) * dx, (j - fx[1]) * dx};
        real ww = w[i][0] * w[j][1];
        float *g = G(bc[0] + i, bc[1] + j);
        g[0] += ww * (v[2 * p] * particle_mass + affine[0] * dpos[0] + affine[1] * dpos[1]);
        g[1] += ww * (v[2 * p + 1] * particle_mass + affine[2] * dpos[0] + affine[3] * dpos[1]);
        g[2] += ww * particle_mass;
      }
  }
  for (int i = 0; i <= ng; i++)  // grid — mls-mpm88.cpp:37-46
    for (int j = 0; j <= ng; j++) {
      float *g = G(i, j);
      if (g[2] > 0) {
        real m = g[2];
        g[0] /= m; g[1] /= m; g[2] /= m;
        g[1] += dt * -200.0f;
        real boundary = 0.05f, xx = (real)i / ng, yy = (real)j / ng;
        if (xx < boundary || xx > 1 - boundary || yy > 1 - boundary) { g[0] = 0; g[1] = 0; g[2] = 0; }
        if (yy < boundary) g[1] = std::max(0.0f, g[1]);
      }
    }
  for (int64_t p = 0; p < n; p++) {  // G2P — mls-mpm88.cpp:47-68
    int bc[2] = {int(x[2 * p] * inv_dx - 0.5f), int(x[2 * p + 1] * inv_dx - 0.5f)};
    real fx[2] = {x[2 * p] * inv_dx - bc[0], x[2 * p + 1] * inv_dx - bc[1]};
    real w[3][2];
    for (int d = 0; d < 2; d++) {
      w[0][d] = 0.5f * (1.5f - fx[d]) * (1.5f - fx[d]);
      w[1][d] = 0.75f - (fx[d] - 1.0f) * (fx[d] - 1.0f);
      w[2][d] = 0.5f * (fx[d] - 0.5f) * (fx[d] - 0.5f);
    }
    real Cn[4] = {0, 0, 0, 0}, vn[2] = {0, 0};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        real dpos[2] = {i - fx[0], j - fx[1]};
        const float *g = G(bc[0] + i, bc[1] + j);
        real ww = w[i][0] * w[j][1];
        vn[0] += ww * g[0]; vn[1] += ww * g[1];
        Cn[0] += 4 * inv_dx * (ww * g[0]) * dpos[0]; Cn[1] += 4 * inv_dx * (ww * g[0]) * dpos[1];
        Cn[2] += 4 * inv_dx * (ww * g[1]) * dpos[0]; Cn[3] += 4 * inv_dx * (ww * g[1]) * dpos[1];
      }
    for (int i = 0; i < 4; i++) C[4 * p + i] = Cn[i];
    v[2 * p] = vn[0]; v[2 * p + 1] = vn[1];
    x[2 * p] += dt * vn[0]; x[2 * p + 1] += dt * vn[1];
    float *f = F + 4 * p;
    real a[4] = {1 + dt * Cn[0], dt * Cn[1], dt * Cn[2], 1 + dt * Cn[3]};
    real Fn[4] = {a[0] * f[0] + a[1] * f[2], a[0] * f[1] + a[1] * f[3], a[2] * f[0] + a[3] * f[2], a[2] * f[1] + a[3] * f[3]};
    double A[2][2] = {{Fn[0], Fn[1]}, {Fn[2], Fn[3]}}, U[2][2], s[2], V[2][2];
    svd2_d(A, U, s, V);
    real sg[2] = {(real)s[0], (real)s[1]};
    if (plastic) for (int i = 0; i < 2; i++) sg[i] = std::min(std::max(sg[i], 1.0f - 2.5e-2f), 1.0f + 7.5e-3f);
    real oldJ = Fn[0] * Fn[3] - Fn[1] * Fn[2];
    real Fo[4];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) Fo[2 * i + j] = (real)(U[i][0] * sg[0] * V[j][0] + U[i][1] * sg[1] * V[j][1]);
    real newJ = Fo[0] * Fo[3] - Fo[1] * Fo[2];
    real Jp_new = std::min(std::max(Jp[p] * oldJ / newJ, 0.6f), 20.0f);
    Jp[p] = Jp_new;
    for (int i = 0; i < 4; i++) f[i] = Fo[i];
  }
}

}  // extern "C"
