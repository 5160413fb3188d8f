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


}  // namespace mpm
