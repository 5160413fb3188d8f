// taichi_mpm_amd/csrc/k_debug.h — device math exposed for parity tests (mpmhip_debug_*)
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

// ------------------------------------------------------------------------------------------------ bandwidth probe
// plain streaming copy (16 bytes per lane, grid-stride): what this box's HBM delivers to the simplest possible
// kernel — the yardstick bench.py reports next to the nominal 8 TB/s
// UNROLL independent 16-byte loads per lane are in flight before the first store; NT = non-temporal stores
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_stream_copy(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) st_rec(dst + i + u * stride, v[u], NT);
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// 64-byte record gather with a KNOWN byte count: the access pattern of the transfer kernels (one lane fetches one
// whole record with four 16-byte loads through an index), used to calibrate rocprofv3's FETCH_SIZE for this width
// (profiles/calibrate_fetch.py): n records of 64 B + n indices of 4 B read, 16 B per workgroup written.
__global__ __launch_bounds__(256) void k_gather_records_probe(const float4 *__restrict__ rec, const uint32_t *__restrict__ idx,
                                                              uint32_t n, float4 *__restrict__ out) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const size_t i = idx[p];
    const float4 a = rec[i * 4 + 0], b = rec[i * 4 + 1], c = rec[i * 4 + 2], d = rec[i * 4 + 3];
    acc.x += a.x + b.x; acc.y += a.y + c.y; acc.z += b.z + d.z; acc.w += c.w + d.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = acc;
}
// index patterns for the probe: 0 identity, 1 blocks of 8 consecutive records in shuffled order (the decayed order of an
// evolved scene: cell-sized runs), 2 fully shuffled (multiplicative hash bijection on a power-of-two range)
__global__ __launch_bounds__(256) void k_probe_indices(uint32_t *__restrict__ idx, uint32_t n, int mode) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    uint32_t i = p;
    if (mode == 1) i = ((((p >> 3) * 2654435761u) & ((n >> 3) - 1u)) << 3) | (p & 7u);   // n: power of two
    else if (mode == 2) i = (p * 2654435761u) & (n - 1u);
    idx[p] = i;
  }
}

// ------------------------------------------------------------------------------------------------ debug math
__global__ void k_debug_svd(int64_t n, const float *F, float *U, float *S, float *V) {
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mat3 f, u;
    for (int k = 0; k < 9; k++) f.m[k] = F[9 * i + k];
    float lam[3], s[3];
    sym_eig3_FFt(f, u, lam);
    signed_sigma(lam, mat_det(f), s);
    for (int k = 0; k < 9; k++) U[9 * i + k] = u.m[k];
    for (int k = 0; k < 3; k++) S[3 * i + k] = s[k];
    // V = F^T U S^-1 (never needed by the product path; provided for the parity test)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) V[9 * i + 3 * r + c] = (f(0, r) * u(0, c) + f(1, r) * u(1, c) + f(2, r) * u(2, c)) / s[c];
  }
}
__global__ void k_debug_force(GroupParams g, int64_t n, const float *F, const float *aux, float *out) {
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mat3 f;
    for (int k = 0; k < 9; k++) f.m[k] = F[9 * i + k];
    mat3 r = calculate_force(g, f, aux[i]);
    for (int k = 0; k < 9; k++) out[9 * i + k] = r.m[k];
  }
}
// plasticity alone, or (force_out != nullptr) the fused plasticity + next-step force of k_g2p
// (the fused form runs as k_g2p runs it: the refinement of an ill-conditioned F works in 20 floats of LDS private to the lane)
__global__ __launch_bounds__(256) void k_debug_plasticity(GroupParams g, int64_t n, const float *cdg, float *F, float *aux,
                                                          float *force_out) {
  __shared__ float4 slab[256 * 5];
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mat3 f, c;
    for (int k = 0; k < 9; k++) { f.m[k] = F[9 * i + k]; c.m[k] = cdg[9 * i + k]; }
    float a = aux[i];
    if (force_out) {
      mat3 st;
      plasticity_and_force<MAT_ALL, true>(g, c, f, a, st, reinterpret_cast<float *>(slab + threadIdx.x * 5));
      for (int k = 0; k < 9; k++) force_out[9 * i + k] = st.m[k];
    } else {
      plasticity(g, c, f, a);
    }
    if (g.type != MPMHIP_WATER)
      for (int k = 0; k < 9; k++) F[9 * i + k] = f.m[k];
    aux[i] = a;
  }
}

// ------------------------------------------------------------------------------------------------ cond(F) census
// How ill-conditioned are the deformation gradients of the state a ctx holds?  Per live particle the eigen-solve of F F^T as the
// transfer kernels run it (sym_eig3_FFt, refinement left out) -> cond(F) = sqrt(lam_max / lam_min); out: [0] live particles,
// [1] particles with lam_min < lam_max / 64 — the ones sym_eig3_FFt hands to sym_eig3_refine (cond > 8) —, [2] waves of 64
// consecutive slots, [3] waves holding at least one such particle, [4] max cond (float bits), [8 + b] histogram over
// b = clamp(floor(8 log2 cond), 0, 255): eighth-octave bins (bench.py: `evolved.cond_F`; DESIGN.md section 2, table of the
// ill-conditioned fixture: the device's tolerances hold to cond 1e2).
constexpr int COND_BINS = 256;
__global__ __launch_bounds__(256) void k_cond_census(Params P, const float4 *__restrict__ rg, unsigned long long *__restrict__ out) {
  const uint32_t n = P.n_slots, stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count (ballots)
    const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false, ill = false;
    if (i < n) {
      const float4 a = rg[(size_t)i * 4], b = rg[(size_t)i * 4 + 1], c = rg[(size_t)i * 4 + 2], d = rg[(size_t)i * 4 + 3];
      live = __float_as_int(d.z) >= 0;
      if (live) {
        const float F[9] = {b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x};
        // eigenvalues of the symmetric F F^T in double (closed form would do; a few Jacobi sweeps are simpler and exact enough here)
        double A[3][3];
        for (int r = 0; r < 3; r++)
          for (int q = 0; q < 3; q++) A[r][q] = (double)F[3 * r] * F[3 * q] + (double)F[3 * r + 1] * F[3 * q + 1] + (double)F[3 * r + 2] * F[3 * q + 2];
        for (int sweep = 0; sweep < 6; sweep++)
          for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2, r = 3 - p - q;
            const double apq = A[p][q];
            if (fabs(apq) < 1e-300) continue;
            const double th = 0.5 * (A[q][q] - A[p][p]) / apq;
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
            const double app = A[p][p] - t * apq, aqq = A[q][q] + t * apq;
            const double arp = cs * A[r][p] - sn * A[r][q], arq = sn * A[r][p] + cs * A[r][q];
            A[p][p] = app; A[q][q] = aqq; A[p][q] = A[q][p] = 0.0;
            A[r][p] = A[p][r] = arp; A[r][q] = A[q][r] = arq;
          }
        const double lmax = fmax(A[0][0], fmax(A[1][1], A[2][2])), lmin = fmin(A[0][0], fmin(A[1][1], A[2][2]));
        ill = lmin < lmax / 64.0;
        const float cond = lmin > 0.0 ? (float)sqrt(lmax / lmin) : 3.0e38f;
        int bin = (int)floorf(8.0f * log2f(fmaxf(cond, 1.0f)));
        bin = bin < 0 ? 0 : (bin >= COND_BINS ? COND_BINS - 1 : bin);
        atomicAdd(&out[8 + bin], 1ull);
        atomicMax(reinterpret_cast<unsigned int *>(&out[4]), __float_as_uint(cond));
      }
    }
    const unsigned long long lv = __ballot(live), il = __ballot(ill);
    if ((threadIdx.x & 63) == 0 && lv) {
      atomicAdd(&out[0], (unsigned long long)__popcll(lv));
      atomicAdd(&out[1], (unsigned long long)__popcll(il));
      atomicAdd(&out[2], 1ull);
      if (il) atomicAdd(&out[3], 1ull);
    }
  }
}

}  // namespace mpm
