// taichi_mpm_amd/csrc/k_bgeo.h — frame output: the per-particle rows of the .bgeo file the reference writes
// (MPM<dim>::write_partio, src/visualize.cpp:17-100, through Partio's writeBGEO, external/partio/src/io/BGEO.cpp:57-194)
// are assembled ON THE DEVICE — gathered in ascending-id order, interleaved and byte-swapped to big-endian — so the
// host only copies one contiguous block behind the header.  Part of libmpmhip; off the hot path.
#pragma once
#include "mpm_common.h"

namespace mpm {

constexpr int BGEO_W_PLAIN = 12;    // x y z w | type | index | limit[3] | v[3]
constexpr int BGEO_W_VERBOSE = 23;  // ... | m | boundary_normal[3] | debug[3] | states | boundary_distance | near_boundary | apic_frobenius_norm

// per slot: creation id, or -1 for a dead slot (the host derives the ascending-id order from this)
__global__ __launch_bounds__(256) void k_bgeo_ids(Params P, const RecG *__restrict__ rg, int32_t *__restrict__ ids) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) ids[i] = rg[i].pid;
}

__device__ __forceinline__ uint32_t be(float f) { return __builtin_bswap32(__float_as_uint(f)); }
__device__ __forceinline__ uint32_t be(int32_t i) { return __builtin_bswap32((uint32_t)i); }

// row j <- particle in slot order[j].  Without rigid bodies the reference's remaining per-particle fields keep their
// constructor values (src/particles.h:92-99): type = is_rigid = 0, limit = (1, 1, 1) unless async stepping is on,
// boundary_normal = 0, states = 0, boundary_distance = 0, near_boundary = 0.
template <bool VERBOSE>
__global__ __launch_bounds__(256) void k_bgeo_rows(uint32_t n, const uint32_t *__restrict__ order, const RecG *__restrict__ rg,
                                                   const RecP *__restrict__ rp, const float *__restrict__ rb,
                                                   const GroupParams *__restrict__ groups,
                                                   const int32_t *__restrict__ limits /* [slot][3] or nullptr */,
                                                   uint32_t *__restrict__ rows) {
  constexpr int W = VERBOSE ? BGEO_W_VERBOSE : BGEO_W_PLAIN;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t s = order[j];
    const RecG g = rg[s];
    const RecP p = rp[s];
    uint32_t w[W];
    w[0] = be(g.x[0]); w[1] = be(g.x[1]); w[2] = be(g.x[2]); w[3] = be(1.0f);  // homogeneous coordinate, BGEO.cpp:158-160
    w[4] = be(0);                                                                // type = int(is_rigid())
    w[5] = be(g.pid);                                                            // index = id
    w[6] = w[7] = w[8] = be(1);                                                  // dt_limit, stiffness_limit, cfl_limit
    if (limits) {  // async stepping: the limits of the particle's scheduler block (src/async/async_visualize.cpp:17-26)
      w[6] = be(limits[3 * (size_t)s]); w[7] = be(limits[3 * (size_t)s + 1]); w[8] = be(limits[3 * (size_t)s + 2]);
    }
    w[9] = be(p.v[0]); w[10] = be(p.v[1]); w[11] = be(p.v[2]);
    if constexpr (VERBOSE) {
      const GroupParams gp = groups[g.gid];
      w[12] = be(gp.p[0]);  // m = get_mass()
      w[13] = w[14] = w[15] = be(0.0f);
      // get_debug_info(): (0, material number, 0); water (j, 5, sticky = 0); elastic (E, 8, 0) (src/particles.cpp:157..839)
      const float d0 = gp.type == MPMHIP_WATER ? g.aux : gp.type == MPMHIP_ELASTIC ? gp.p[4] : 0.0f;
      w[16] = be(d0); w[17] = be((float)gp.type); w[18] = be(0.0f);
      w[19] = be((int32_t)g.pad); w[20] = be(0.0f); w[21] = be(0);  // states = the CPIC colour word (0 without rigid bodies)
      const float *b = rb + (size_t)s * BW;  // || 0.5 (apic_b - apic_b^T) ||_F   (visualize.cpp:70-71)
      float sum = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float h = 0.5f * (b[a * 3 + c] - b[c * 3 + a]);
          sum = __fmaf_rn(h, h, sum);
        }
      w[22] = be(sqrtf(sum));  // correctly rounded (HIP default); __fsqrt_rn maps to the native approximation here
    }
    uint32_t *dst = rows + (size_t)j * W;
    if constexpr (!VERBOSE) {  // 48-byte rows: three aligned 16-byte stores
      uint4 *d4 = reinterpret_cast<uint4 *>(dst);
      d4[0] = make_uint4(w[0], w[1], w[2], w[3]);
      d4[1] = make_uint4(w[4], w[5], w[6], w[7]);
      d4[2] = make_uint4(w[8], w[9], w[10], w[11]);
    } else {
#pragma unroll
      for (int k = 0; k < W; k++) dst[k] = w[k];
    }
  }
}

}  // namespace mpm
