// taichi_mpm_amd/csrc/k_particles.h — per-particle state kernels off the hot path: affine matrix (re)build, apic_b recovery, potential energy
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

// ------------------------------------------------------------------------------------------------ affine
// A = stress * (-4 inv_dx dt) + apic_b * (4 m)   (src/transfer.cpp:465,507,521-522) for every live particle,
// from (F, aux, apic_b).  Needed only when the state did not come out of k_g2p (first substep, uploads).
__global__ __launch_bounds__(256) void k_affine(Params P, const RecG *__restrict__ rg, RecP *__restrict__ rp,
                                                const float *__restrict__ rb, const GroupParams *__restrict__ groups) {
  const float S = -4.0f * P.idx * P.dt;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const RecG r = rg[i];
    if (r.pid < 0) continue;
    const GroupParams g = groups[r.gid];
    mat3 F;
#pragma unroll
    for (int k = 0; k < 9; k++) F.m[k] = r.F[k];
    const mat3 stress = calculate_force(g, F, r.aux);
    const float m4 = 4.0f * g.p[0];
#pragma unroll
    for (int k = 0; k < 9; k++) rp[i].A[k] = fmaf(stress.m[k], S, rb[(size_t)i * BW + k] * m4);
  }
}

// inverse of k_affine for ctxs that fold apic_b into A (discard_apic_b): apic_b = (A - stress * S) / (4 m),
// written to the side array.  Runs only when somebody asks for apic_b (download, upload of F/aux, new particles).
// Accuracy: the stress is re-evaluated from the stored (F, aux) by calculate_force(), not by the fused G2P path
// that produced A, so apic_b comes back to ~1e-6 * |stress S| / (4 m) absolute — 1e-5..1e-4 relative in practice.
__global__ __launch_bounds__(256) void k_recover_b(Params P, const RecG *__restrict__ rg, const RecP *__restrict__ rp,
                                                   float *__restrict__ rb, const GroupParams *__restrict__ groups) {
  const float S = -4.0f * P.idx * P.dt;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const RecG r = rg[i];
    if (r.pid < 0) continue;
    const GroupParams g = groups[r.gid];
    mat3 F;
#pragma unroll
    for (int k = 0; k < 9; k++) F.m[k] = r.F[k];
    const mat3 stress = calculate_force(g, F, r.aux);
    const float im4 = 1.0f / (4.0f * g.p[0]);
#pragma unroll
    for (int k = 0; k < 9; k++) rb[(size_t)i * BW + k] = fmaf(-stress.m[k], S, rp[i].A[k]) * im4;
  }
}

// general_action "delete_particles_inside_level_set" (src/mpm.cpp:962-974): every live particle whose level-set
// value at its position is negative is deleted for good (pid = -1, like clear_boundary_particles does in k_g2p)
__global__ __launch_bounds__(256) void k_delete_inside_levelset(Params P, RecG *__restrict__ rg, LevelSetDev LS,
                                                                Counters *cnt) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    if (rg[i].pid < 0) continue;
    const float xw[3] = {rg[i].x[0], rg[i].x[1], rg[i].x[2]};
    float phi, nrm[3];
    if (levelset_eval(LS, P.t, xw, P.idx, phi, nrm) && phi < 0.0f) {
      rg[i].pid = -1;
      atomicAdd(&cnt->n_dead, 1u);
    }
  }
}

// AsyncMPM, first half of update_dt_limits (src/async/async_mpm.cpp:91-111): per SCHEDULER block — the reference's SPGrid
// block of 4 x 4 x 8 nodes holding the particle's base node — the smallest get_allowed_dt(dx) and the largest |v|^2 of
// its particles, and their number.  tab[3 b + {0, 1, 2}] = (bits of min allowed dt, bits of max |v|^2, count); positive
// floats order like their bit patterns, so integer atomics do.  Also writes the block id of every particle.
__global__ __launch_bounds__(256) void k_async_block_reduce(Params P, const RecG *__restrict__ rg, const RecP *__restrict__ rp,
                                                            const GroupParams *__restrict__ groups, int nbx, int nby, int nbz,
                                                            uint32_t *__restrict__ tab, uint32_t *__restrict__ blk_of) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const RecG r = rg[i];
    if (r.pid < 0) { blk_of[i] = INVALID; continue; }
    const RecP q = rp[i];
    int b[3];
    bool in = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      b[k] = (int)(r.x[k] * P.idx - 0.5f);  // get_grid_base_pos, src/mpm.h:252-255
      in = in && r.x[k] * P.idx >= 0.5f;
    }
    const int bx = b[0] >> 2, by = b[1] >> 2, bz = b[2] >> 3;
    if (!in || bx >= nbx || by >= nby || bz >= nbz) { blk_of[i] = INVALID; continue; }
    const uint32_t blk = ((uint32_t)bx * nby + by) * nbz + bz;
    blk_of[i] = blk;
    mat3 F;
#pragma unroll
    for (int k = 0; k < 9; k++) F.m[k] = r.F[k];
    const float adt = allowed_dt(groups[r.gid], F, r.aux, q.v, P.dx);
    const float v2 = q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2];
    atomicMin(&tab[3 * blk + 0], __float_as_uint(fmaxf(adt, 0.0f)));
    atomicMax(&tab[3 * blk + 1], __float_as_uint(v2));
    atomicAdd(&tab[3 * blk + 2], 1u);
  }
}
// per-particle (dt_limit, stiffness_limit, cfl_limit) = its block's (AsyncMPM::visualize, src/async/async_visualize.cpp:17-26)
__global__ __launch_bounds__(256) void k_async_particle_limits(Params P, const uint32_t *__restrict__ blk_of,
                                                               const int32_t *__restrict__ blk_limits, int32_t *__restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const uint32_t b = blk_of[i];
#pragma unroll
    for (int k = 0; k < 3; k++) out[3 * (size_t)i + k] = b == INVALID ? 1 : blk_limits[3 * (size_t)b + k];
  }
}
__global__ void k_debug_allowed_dt(GroupParams g, int64_t n, const float *F, const float *aux, const float *v, float dx, float *out) {
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mat3 f;
    for (int k = 0; k < 9; k++) f.m[k] = F[9 * i + k];
    const float vv[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    out[i] = allowed_dt(g, f, aux[i], vv, dx);
  }
}

// sum of MPMParticle::potential_energy() (src/particles.cpp:323-327 linear, :400-407 jelly, :785-796 elastic;
// the other types do not define it in the reference: TC_NOT_IMPLEMENTED) -> out[0]; out[1] counts particles of
// types without a potential energy
__global__ __launch_bounds__(256) void k_potential_energy(Params P, const RecG *__restrict__ rg,
                                                          const GroupParams *__restrict__ groups, double *out) {
  double e = 0.0, bad = 0.0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const RecG r = rg[i];
    if (r.pid < 0) continue;
    const GroupParams g = groups[r.gid];
    mat3 F;
#pragma unroll
    for (int k = 0; k < 9; k++) F.m[k] = r.F[k];
    const float mu = g.p[2], la = g.p[3], vol = g.p[1];
    if (g.type == MPMHIP_LINEAR) {
      float n2 = 0.0f, tr = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const float eab = 0.5f * (F(a, b) + F(b, a)) - (a == b ? 1.0f : 0.0f);
          n2 = fmaf(eab, eab, n2);
          if (a == b) tr += eab;
        }
      e += vol * (mu * n2 + 0.5f * la * tr * tr);
    } else if (g.type == MPMHIP_JELLY || g.type == MPMHIP_ELASTIC) {
      mat3 U; float lam[3], s[3];
      sym_eig3_FFt(F, U, lam);
      const float J = mat_det(F);
      signed_sigma(lam, J, s);
      if (g.type == MPMHIP_JELLY) {  // |F - R|_F^2 = sum (sigma - 1)^2
        const float n2 = (s[0] - 1) * (s[0] - 1) + (s[1] - 1) * (s[1] - 1) + (s[2] - 1) * (s[2] - 1);
        e += vol * (mu * n2 + 0.5f * la * (J - 1.0f) * (J - 1.0f));
      } else {
        const float l0 = logf(fabsf(s[0])), l1 = logf(fabsf(s[1])), l2 = logf(fabsf(s[2]));
        const float sum = l0 + l1 + l2;
        e += vol * (mu * (l0 * l0 + l1 * l1 + l2 * l2) + 0.5f * la * sum * sum);
      }
    } else {
      bad += 1.0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { e += __shfl_xor(e, off); bad += __shfl_xor(bad, off); }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], e);
    if (bad != 0.0) atomicAdd(&out[1], bad);
  }
}


}  // namespace mpm
