// taichi_mpm_amd/csrc/mpmhip.hip — MI355X (gfx950) MLS-MPM time-stepping core: HIP kernels + C ABI.
//
// One substep (reference: MPM<3>::substep, src/mpm.cpp:452-575) is
//
//   sort      k_build_keys -> k_scan_*<0> (bitmap prefix) -> k_emit_active -> k_rank -> k_block_totals ->
//             k_scan_*<1> (block offsets) -> k_cell_start -> k_reorder -> k_sort_cleanup
//             (replaces sort_particles_and_populate_grid src/mpm.cpp:770-918, sort_allocator :752-768
//              and clear_boundary_particles :582-633: dead particles simply get no slot)
//   P2G       k_p2g        one wavefront per active 4x4x4-cell block, one lane per cell: register
//                          accumulation over the cell's particles, then DS float atomics into the 6^3-node
//                          LDS tile, tile written out non-atomically                (src/transfer.cpp:467-569)
//   grid      k_grid       sums the <=8 overlapping block tiles of every touched grid block, normalises,
//                          applies gravity + level-set boundary                  (src/mpm.cpp:277-372)
//   G2P       k_g2p        6^3 velocity tile in LDS, 27-tap gather, F update + plasticity, advection
//                                                                                (src/transfer.cpp:837-954)
//
// Data layout (all fp32, resident in HBM for the life of the ctx):
//   particles   SoA, 25 float arrays (x3 v3 B9 F9 aux1) + u8 group id + i32 creation id, ping-pong pair;
//               physically sorted every substep by key = Morton(block) << 6 | cell-in-block, so a
//               workgroup's particles are one contiguous, coalesced range.
//   blocks      an "active" block = 4x4x4 cells holding >=1 particle.  A bitmap over the Morton block
//               space + per-word popcount prefix gives each active block a dense slot (its rank in
//               Morton order) without any pass over the whole grid.
//   tiles       float4[216] per active block: the block's private (4+2)^3-node P2G result.
//   gridv       float4[64] per touched grid block (v.xyz, m) in block-major order; slot = 8*a+o of the
//               owning (active block a, corner offset o); `fat_slot` maps Morton block -> slot.
// No global float atomics anywhere: inter-block write conflicts of P2G are resolved by the tile
// reduction in k_grid, which makes the grid deterministic given the per-tile sums.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mpm_math.h"

namespace mpm {

constexpr int BS = 4;    // cells per block edge
constexpr int BC = 64;   // cells per block
constexpr int TS = 6;    // tile edge in nodes (BS + 2: quadratic stencil reaches base+2)
constexpr int TN = 216;  // nodes per tile
constexpr uint32_t INVALID = 0xFFFFFFFFu;
constexpr int NF = 25;   // float fields per particle
enum { FX = 0, FV = 3, FB = 6, FF = 15, FAUX = 24 };

struct SoA {
  float *f[NF];
  uint8_t *gid;
  int32_t *pid;
};

struct Counters {
  uint32_t n;         // live particles in the current SoA
  uint32_t n_next;    // live particles after the reorder in flight
  uint32_t n_active;  // active blocks
  uint32_t error;     // bit0: active blocks exceeded max_blocks
};

struct Params {
  int res[3];
  float dx, idx, dt;
  float g[3];
  int particle_gravity;
  float apic_damping, rpic_damping;
  int clean_boundary;
  int n_planes;
  float planes[8][4];
  float friction;
  int kbits;         // Morton bits per axis
  uint32_t nbw;      // bitmap words = 8^kbits / 32
  uint32_t max_blocks;
};

// ------------------------------------------------------------------------------------------------ Morton
__host__ __device__ __forceinline__ uint32_t spread3(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__host__ __device__ __forceinline__ uint32_t compact3(uint32_t v) {
  v &= 0x09249249u;
  v = (v | (v >> 2)) & 0x030c30c3u;
  v = (v | (v >> 4)) & 0x0300f00fu;
  v = (v | (v >> 8)) & 0x030000ffu;
  v = (v | (v >> 16)) & 0x3ffu;
  return v;
}
__host__ __device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
}
__host__ __device__ __forceinline__ void demorton3(uint32_t m, int &x, int &y, int &z) {
  x = (int)compact3(m >> 2); y = (int)compact3(m >> 1); z = (int)compact3(m);
}

__device__ __forceinline__ bool block_active(const uint32_t *__restrict__ bits, uint32_t bkey) {
  return (bits[bkey >> 5] >> (bkey & 31)) & 1u;
}
__device__ __forceinline__ uint32_t block_slot(const uint32_t *__restrict__ bits, const uint32_t *__restrict__ wprefix,
                                               uint32_t bkey) {
  const uint32_t w = bits[bkey >> 5];
  return wprefix[bkey >> 5] + __popc(w & ((1u << (bkey & 31)) - 1u));
}

// ------------------------------------------------------------------------------------------------ sort
// key of a particle: Morton(block of its base cell) << 6 | cell in block; INVALID for dead particles
// (non-finite x/v, near the domain wall when clean_boundary — src/mpm.h:269-276, src/mpm.cpp:592-598 —
// or with a stencil that would leave the grid, where the reference has undefined behaviour).
__global__ __launch_bounds__(256) void k_build_keys(Params P, SoA s, const Counters *__restrict__ cnt,
                                                    uint32_t *__restrict__ key, uint8_t *__restrict__ blk_flag) {
  const uint32_t n = cnt->n;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count: all lanes take part in the shuffle
    const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    float X[3];
    bool alive = i < n;
    const uint32_t ii = alive ? i : 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float x = s.f[FX + k][ii], v = s.f[FV + k][ii];
      alive = alive && isfinite(x) && isfinite(v);
      X[k] = x * P.idx;
    }
    if (P.clean_boundary) {
      const float mn = fminf(X[0], fminf(X[1], X[2]));
      const float mx = fmaxf(X[0] - P.res[0], fmaxf(X[1] - P.res[1], X[2] - P.res[2]));
      alive = alive && !(mn < 7.0f || mx > -7.0f);
    }
    int b[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      alive = alive && (X[k] >= 0.5f);
      b[k] = alive ? (int)(X[k] - 0.5f) : 0;  // MPMKernel<dim,2>::get_stencil_start, src/kernel.h:119-121
      alive = alive && (b[k] + 2 <= P.res[k]);
    }
    uint32_t kk = INVALID, bkey = INVALID;
    if (alive) {
      bkey = morton3(b[0] >> 2, b[1] >> 2, b[2] >> 2);
      kk = (bkey << 6) | ((b[0] & 3) << 4) | ((b[1] & 3) << 2) | (b[2] & 3);
    }
    // mark the block active: a plain byte store (all writers store the same value, no atomics, no
    // serialisation), one per run of equal blocks in the wave; k_pack_flags turns the bytes into the bitmap
    const uint32_t prev = __shfl_up(bkey, 1);
    if (alive && ((threadIdx.x & 63) == 0 || prev != bkey)) blk_flag[bkey] = 1;
    if (i < n) key[i] = kk;
  }
}

// byte flags -> active-block bitmap (bit b of word w = block with Morton key 32w+b); clears the flags
__global__ __launch_bounds__(256) void k_pack_flags(Params P, uint8_t *__restrict__ blk_flag,
                                                    uint32_t *__restrict__ bits) {
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < P.nbw; w += gridDim.x * blockDim.x) {
    uint4 *src = reinterpret_cast<uint4 *>(blk_flag + (size_t)w * 32);
    const uint4 lo = src[0], hi = src[1];
    const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      // bytes are 0/1: gather bit 0 of each of the 4 bytes
      const uint32_t v = q[k];
      m |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * k);
    }
    bits[w] = m;
    if (m) { src[0] = make_uint4(0, 0, 0, 0); src[1] = make_uint4(0, 0, 0, 0); }
  }
}

// ---- multi-workgroup exclusive scan of a uint32 sequence, two launches:
//   k_scan_partials: workgroup i reduces chunk i (SCAN_CHUNK elements) -> partials[i]
//   k_scan_apply   : workgroup i adds the partials before it to a local scan of its chunk
// MODE 0: element w = popcount(bits[w]) (active-block bitmap; 8^k/8 bytes, 256 KiB for a 256^3 grid),
//         out = word_prefix (number of active blocks with Morton key < 32 w), total -> cnt->n_active
// MODE 1: element a = particles in active block a, out = act_start[0..n_active], total -> cnt->n_next
constexpr int SCAN_CHUNK = 2048;  // 256 threads x 8

__device__ __forceinline__ uint32_t wg_exclusive_scan_256(uint32_t v, uint32_t *lds /*>=5*/, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(inc, off);
    if ((int)lane >= off) inc += u;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}

template <int MODE>
__device__ __forceinline__ uint32_t scan_count(const Params &P, const Counters *cnt) {
  return MODE == 0 ? P.nbw : min(cnt->n_active, P.max_blocks);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_scan_partials(Params P, const Counters *__restrict__ cnt,
                                                       const uint32_t *__restrict__ in,
                                                       uint32_t *__restrict__ partials) {
  __shared__ uint32_t lds[8];
  const uint32_t n = scan_count<MODE>(P, cnt);
  const uint32_t base = blockIdx.x * SCAN_CHUNK;
  if (base >= n) return;
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = base + k * 256 + threadIdx.x;
    if (i < n) sum += (MODE == 0) ? (uint32_t)__popc(in[i]) : in[i];
  }
  uint32_t total;
  wg_exclusive_scan_256(sum, lds, total);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_scan_apply(Params P, Counters *cnt, const uint32_t *__restrict__ in,
                                                    const uint32_t *__restrict__ partials,
                                                    uint32_t *__restrict__ out) {
  __shared__ uint32_t lds[8];
  const uint32_t n = scan_count<MODE>(P, cnt);
  const uint32_t base = blockIdx.x * SCAN_CHUNK;
  if (base >= n && !(n == 0 && blockIdx.x == 0)) return;
  // sum of the partials of the chunks before this one
  uint32_t pre = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256) pre += partials[j];
  uint32_t chunk_base;
  wg_exclusive_scan_256(pre, lds, chunk_base);
  // thread t owns 8 consecutive elements
  uint32_t v[8], sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = base + threadIdx.x * 8 + k;
    v[k] = (i < n) ? ((MODE == 0) ? (uint32_t)__popc(in[i]) : in[i]) : 0u;
    sum += v[k];
  }
  uint32_t total;
  uint32_t run = chunk_base + wg_exclusive_scan_256(sum, lds, total);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = base + threadIdx.x * 8 + k;
    if (i < n) out[i] = run;
    run += v[k];
  }
  if (base + SCAN_CHUNK >= n && threadIdx.x == 255) {  // last chunk: publish the grand total
    const uint32_t grand = chunk_base + total;
    if (MODE == 0) {
      if (grand > P.max_blocks) cnt->error |= 1u;
      cnt->n_active = grand;
    } else {
      out[n] = grand;
      cnt->n_next = grand;
    }
  }
}

__global__ __launch_bounds__(256) void k_emit_active(Params P, const uint32_t *__restrict__ bits,
                                                     const uint32_t *__restrict__ wprefix,
                                                     uint32_t *__restrict__ act_blk) {
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < P.nbw; w += gridDim.x * blockDim.x) {
    uint32_t m = bits[w];
    uint32_t slot = wprefix[w];
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      if (slot < P.max_blocks) act_blk[slot] = (w << 5) | (uint32_t)b;
      slot++;
    }
  }
}

// rank of each particle inside its cell.  Runs of equal keys in consecutive lanes (the common case:
// particles are already nearly sorted) are aggregated into one returning atomic per run.
__global__ __launch_bounds__(256) void k_rank(Params P, const Counters *__restrict__ cnt, uint32_t *__restrict__ key,
                                              uint32_t *__restrict__ rank, uint32_t *__restrict__ cell_cnt,
                                              const uint32_t *__restrict__ bits,
                                              const uint32_t *__restrict__ wprefix) {
  const uint32_t n = cnt->n;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nloop = (n + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count: every lane takes part in the shuffles
    const uint32_t i = it * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = (i < n) ? key[i] : INVALID;
    uint32_t cidx = INVALID;
    if (k != INVALID) {
      const uint32_t slot = block_slot(bits, wprefix, k >> 6);
      cidx = (slot < P.max_blocks) ? slot * BC + (k & 63u) : INVALID;
    }
    const uint32_t prev = __shfl_up(cidx, 1);
    const bool head = (lane == 0) || (cidx != prev);
    const unsigned long long H = __ballot(head);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int start = 63 - __clzll(H & le);
    const unsigned long long above = H & ~le;
    const int end = above ? (__ffsll((long long)above) - 1) : 64;
    uint32_t base = 0;
    if ((int)lane == start && cidx != INVALID) base = atomicAdd(&cell_cnt[cidx], (uint32_t)(end - start));
    base = __shfl(base, start);
    if (i < n) {
      key[i] = cidx;  // from here on `key` holds slot*64 + cell
      rank[i] = base + (lane - start);
    }
  }
}

// one wave per active block: particles per block
__global__ __launch_bounds__(256) void k_block_totals(const Counters *__restrict__ cnt,
                                                      const uint32_t *__restrict__ cell_cnt,
                                                      uint32_t *__restrict__ totals) {
  const uint32_t na = min(cnt->n_active, 0x7fffffffu);
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t a = wave; a < na; a += nwaves) {
    uint32_t v = cell_cnt[a * BC + lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) totals[a] = v;
  }
}

// per-cell counts -> global start offset of every cell: cell_start[slot*64 + c], plus a sentinel at
// [n_active*64] so that the particles of cell i are always [cell_start[i], cell_start[i+1])
__global__ __launch_bounds__(256) void k_cell_start(Params P, const Counters *__restrict__ cnt,
                                                    const uint32_t *__restrict__ cell_cnt,
                                                    const uint32_t *__restrict__ act_start,
                                                    uint32_t *__restrict__ cell_start) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t a = wave; a < na; a += nwaves) {
    const uint32_t c = cell_cnt[a * BC + lane];
    uint32_t v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(v, off);
      if ((int)lane >= off) v += u;
    }
    cell_start[a * BC + lane] = act_start[a] + v - c;
    if (a == na - 1 && lane == 63) cell_start[na * BC] = act_start[a] + v;
  }
  if (na == 0 && wave == 0 && lane == 0) cell_start[0] = 0;
}

// Layout of a block's particles in memory: RANK-MAJOR.  First the 0th particle of every non-empty cell (in cell
// order), then the 1st particle of every cell that has one, ...  With the one-lane-per-cell mapping of k_p2g,
// iteration r of the wave then reads 64 (or fewer) CONSECUTIVE particles: fully coalesced SoA loads.
// dest[cell_start[c] + r] = final slot of the particle with rank r in cell c.  One wave per block.
__global__ __launch_bounds__(256) void k_layout(Params P, const Counters *__restrict__ cnt,
                                                const uint32_t *__restrict__ cell_start,
                                                uint32_t *__restrict__ dest) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t a = wave; a < na; a += nwaves) {
    const uint32_t s0 = cell_start[a * BC + lane], s1 = cell_start[a * BC + lane + 1];
    const uint32_t c = s1 - s0;
    uint32_t run = __shfl(s0, 0);  // block start
    for (uint32_t r = 0;; r++) {
      const unsigned long long m = __ballot(c > r);
      if (!m) break;
      if (c > r) dest[s0 + r] = run + (uint32_t)__popcll(m & lt);
      run += (uint32_t)__popcll(m);
    }
  }
}

// physical reorder: scatter every live particle to its sorted slot (sort_allocator, src/mpm.cpp:752-768,
// done every substep here; dead particles are dropped = clear_boundary_particles, src/mpm.cpp:582-633)
__global__ __launch_bounds__(256) void k_reorder(const Counters *__restrict__ cnt, SoA src, SoA dst,
                                                 const uint32_t *__restrict__ key, const uint32_t *__restrict__ rank,
                                                 const uint32_t *__restrict__ cell_start,
                                                 const uint32_t *__restrict__ dest) {
  const uint32_t n = cnt->n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t c = key[i];
    if (c == INVALID) continue;
    const uint32_t j = dest[cell_start[c] + rank[i]];
#pragma unroll
    for (int f = 0; f < NF; f++) dst.f[f][j] = src.f[f][i];
    dst.gid[j] = src.gid[i];
    dst.pid[j] = src.pid[i];
  }
}

__global__ __launch_bounds__(256) void k_sort_cleanup(Params P, Counters *cnt, uint32_t *__restrict__ cell_cnt) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t total = na * BC;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) cell_cnt[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt->n = cnt->n_next;
}

// ------------------------------------------------------------------------------------------------ P2G
// rasterize_optimized / block_op_normal (src/transfer.cpp:467-569).
// Mapping: one wavefront per active 4^3-cell block, ONE LANE PER CELL.  Particles are sorted by cell, so lane c
// walks the particles of its cell sequentially and accumulates their 27x4 node contributions in registers
// (the reference walks cells sequentially inside a block and accumulates into its scratch tile the same
// way, :474-483).  Write conflicts between particles of one cell therefore never reach memory; the 27x4
// per-cell sums are then merged into the block's 6^3-node LDS tile by ordered, non-atomic float4
// read-modify-writes (see below).  The tile is written out non-atomically; conflicts between blocks are
// resolved by k_grid.  No atomics of any kind on the P2G path.
__global__ __launch_bounds__(64) void k_p2g(Params P, SoA s, const Counters *__restrict__ cnt,
                                            const uint32_t *__restrict__ act_blk,
                                            const uint32_t *__restrict__ cell_start,
                                            const GroupParams *__restrict__ groups, float4 *__restrict__ tiles) {
  __shared__ float4 tile[TN];  // (m*vx, m*vy, m*vz, m) per node of the block's 6^3 tile
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int lane = threadIdx.x;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const int nbase = (cx * TS + cy) * TS + cz;
  const float S = -4.0f * P.idx * P.dt;  // src/transfer.cpp:465
  for (uint32_t a = blockIdx.x; a < na; a += gridDim.x) {
    for (int t = lane; t < TN; t += 64) tile[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    const float ox = (float)(bx * BS + cx), oy = (float)(by * BS + cy), oz = (float)(bz * BS + cz);
    const uint32_t cs0 = cell_start[a * BC + lane];
    const uint32_t count = cell_start[a * BC + lane + 1] - cs0;
    uint32_t run = __shfl(cs0, 0);  // first particle of the block; rank-major layout (k_layout)
    const unsigned long long lt = (1ull << lane) - 1ull;
    float acc[27][4];
#pragma unroll
    for (int n = 0; n < 27; n++) { acc[n][0] = 0.0f; acc[n][1] = 0.0f; acc[n][2] = 0.0f; acc[n][3] = 0.0f; }
    for (uint32_t r = 0;; r++) {
      const unsigned long long am = __ballot(count > r);
      if (!am) break;
      const uint32_t p = run + (uint32_t)__popcll(am & lt);
      run += (uint32_t)__popcll(am);
      if (count <= r) continue;
      const GroupParams g = groups[s.gid[p]];
      const float mass = g.p[0];
      float v[3] = {s.f[FV][p], s.f[FV + 1][p], s.f[FV + 2][p]};
      if (P.particle_gravity) {  // src/transfer.cpp:485-487
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = fmaf(P.g[k], P.dt, v[k]);
      }
      // position relative to the base cell, in grid units: in [0.5, 1.5)^3  (:490,518)
      const float r0 = s.f[FX][p] * P.idx - ox, r1 = s.f[FX + 1][p] * P.idx - oy, r2 = s.f[FX + 2][p] * P.idx - oz;
      float w0[3], w1[3], w2[3];
      bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
      mat3 F, B;
#pragma unroll
      for (int k = 0; k < 9; k++) { F.m[k] = s.f[FF + k][p]; B.m[k] = s.f[FB + k][p]; }
      const mat3 stress = calculate_force(g, F, s.f[FAUX][p]);  // :509
      mat3 A;
      const float m4 = 4.0f * mass;  // Kernel::inv_D() * mass, :507
#pragma unroll
      for (int k = 0; k < 9; k++) A.m[k] = fmaf(stress.m[k], S, B.m[k] * m4);  // :521-522
      const float mv0 = mass * v[0], mv1 = mass * v[1], mv2 = mass * v[2];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float d0 = r0 - (float)i;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const float d1 = r1 - (float)j;
          const float wij = w0[i] * w1[j];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const float d2 = r2 - (float)k;
            const float w = wij * w2[k];
            const int n = (i * 3 + j) * 3 + k;
            // :535-541  contrib = (affine * dpos + mass*v, mass); g += weight * contrib
            const float q0 = fmaf(A(0, 2), d2, fmaf(A(0, 1), d1, fmaf(A(0, 0), d0, mv0)));
            const float q1 = fmaf(A(1, 2), d2, fmaf(A(1, 1), d1, fmaf(A(1, 0), d0, mv1)));
            const float q2 = fmaf(A(2, 2), d2, fmaf(A(2, 1), d1, fmaf(A(2, 0), d0, mv2)));
            acc[n][0] = fmaf(w, q0, acc[n][0]);
            acc[n][1] = fmaf(w, q1, acc[n][1]);
            acc[n][2] = fmaf(w, q2, acc[n][2]);
            acc[n][3] = fmaf(w, mass, acc[n][3]);
          }
        }
      }
    }
    // Merge the per-cell sums into the tile.  The tile belongs to this wavefront alone, and within one
    // (i,j,k) step all 64 lanes address distinct nodes (same stencil offset, different cells), so a plain
    // float4 read-modify-write is race-free as long as the 27 steps stay in program order: LDS operations of
    // one wave execute in order, the wave_barrier keeps the compiler from interleaving them.  (DS float atomics
    // cost ~2 LDS cycles per LANE on gfx950 even without conflicts: measured 145 cycles per ds_add_f32.)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int n = (i * 3 + j) * 3 + k;
          const int node = nbase + (i * TS + j) * TS + k;
          if (count > 0) {
            float4 t = tile[node];
            t.x += acc[n][0]; t.y += acc[n][1]; t.z += acc[n][2]; t.w += acc[n][3];
            tile[node] = t;
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
        }
    __syncthreads();
    for (int t = lane; t < TN; t += 64)
      tiles[(size_t)a * TN + t] = tile[t];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ grid
// Candidate (a, o): grid block c = block(a) + o, o in {0,1}^3, is one of the 8 grid blocks the tile of
// active block a overlaps.  c is processed by its "owner": the candidate with the smallest o among the
// active blocks c - o'.  The owner sums the overlapping tiles (<= 8), then
//   mode 0: normalize_grid_and_apply_external_force + apply_grid_boundary_conditions (src/mpm.cpp:277-372)
//           -> gridv[slot = 8a+o], fat_slot[morton(c)] = slot
//   mode 1: raw (m v, m) sums written to a dense node-major array (parity / download only)
//   mode 2: dense (v, m) array -> gridv (upload_grid)        mode 3: gridv -> dense (download_grid)
__global__ __launch_bounds__(64) void k_grid(Params P, int mode, const Counters *__restrict__ cnt,
                                             const uint32_t *__restrict__ act_blk,
                                             const uint32_t *__restrict__ bits,
                                             const uint32_t *__restrict__ wprefix,
                                             const float4 *__restrict__ tiles, float4 *__restrict__ gridv,
                                             uint32_t *__restrict__ fat_slot, float4 *__restrict__ dense) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int l = threadIdx.x;
  const int lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
  for (uint32_t cand = blockIdx.x; cand < na * 8u; cand += gridDim.x) {
    const uint32_t a = cand >> 3;
    const int o = cand & 7;
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    const int cx = bx + (o >> 2), cy = by + ((o >> 1) & 1), cz = bz + (o & 1);
    // owner test + gather of contributing tiles (all wave-uniform)
    bool owner = true;
    uint32_t src_slot[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int sx = cx - (q >> 2), sy = cy - ((q >> 1) & 1), sz = cz - (q & 1);
      src_slot[q] = INVALID;
      if (sx >= 0 && sy >= 0 && sz >= 0) {
        const uint32_t bk = morton3(sx, sy, sz);
        if (block_active(bits, bk)) {
          if (q < o) owner = false;
          src_slot[q] = block_slot(bits, wprefix, bk);
        }
      }
    }
    if (!owner) continue;
    const uint32_t slot = a * 8u + (uint32_t)o;
    const int gi = cx * BS + lx, gj = cy * BS + ly, gk = cz * BS + lz;
    const bool in_grid = gi <= P.res[0] && gj <= P.res[1] && gk <= P.res[2];
    const size_t dense_idx = ((size_t)gi * (P.res[1] + 1) + gj) * (P.res[2] + 1) + gk;
    if (mode == 2) {
      gridv[(size_t)slot * BC + l] = in_grid ? dense[dense_idx] : make_float4(0, 0, 0, 0);
      if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
      continue;
    }
    if (mode == 3) {
      if (in_grid) dense[dense_idx] = gridv[(size_t)slot * BC + l];
      continue;
    }
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int tx = lx + 4 * (q >> 2), ty = ly + 4 * ((q >> 1) & 1), tz = lz + 4 * (q & 1);
      if (src_slot[q] != INVALID && tx < TS && ty < TS && tz < TS) {
        const float4 t = tiles[(size_t)src_slot[q] * TN + (tx * TS + ty) * TS + tz];
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
    }
    if (mode == 1) {
      if (in_grid) dense[dense_idx] = acc;
      continue;
    }
    float v[3] = {acc.x, acc.y, acc.z};
    const float m = acc.w;
    if (m > 0.0f) {  // src/mpm.cpp:282-292; increment is gravity*dt only when !particle_gravity (:526-530)
      const float im = 1.0f / m;
#pragma unroll
      for (int k = 0; k < 3; k++) v[k] = fmaf(v[k], im, P.particle_gravity ? 0.0f : P.g[k] * P.dt);
    }
    if (m != 0.0f && P.n_planes > 0) {  // src/mpm.cpp:313-368
      float phi = 1e30f, nrm[3] = {0, 0, 0};
      for (int p = 0; p < P.n_planes; p++) {
        const float ph = (P.planes[p][0] * (gi * P.dx) + P.planes[p][1] * (gj * P.dx) + P.planes[p][2] * (gk * P.dx) +
                          P.planes[p][3]) * P.idx;
        if (ph < phi) { phi = ph; nrm[0] = P.planes[p][0]; nrm[1] = P.planes[p][1]; nrm[2] = P.planes[p][2]; }
      }
      if (!(phi < -3.0f || 0.0f < phi)) {
        const float vb[3] = {0, 0, 0};
        friction_project(v, vb, nrm, P.friction);
      }
    }
    gridv[(size_t)slot * BC + l] = make_float4(v[0], v[1], v[2], m);
    if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
  }
}

// ------------------------------------------------------------------------------------------------ G2P
// resample_optimized / block_op_normal (src/transfer.cpp:837-954)
template <int NT, int MINW>
__global__ __launch_bounds__(NT, MINW) void k_g2p(Params P, SoA s, const Counters *__restrict__ cnt,
                                            const uint32_t *__restrict__ act_blk,
                                            const uint32_t *__restrict__ act_start,
                                            const GroupParams *__restrict__ groups,
                                            const float4 *__restrict__ gridv,
                                            const uint32_t *__restrict__ fat_slot) {
  __shared__ float4 tile[TN];
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int tid = threadIdx.x;
  const float scale = -4.0f * P.idx * P.dt;  // :938
  for (uint32_t a = blockIdx.x; a < na; a += gridDim.x) {
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    for (int t = tid; t < TN; t += NT) {
      const int tx = t / (TS * TS), ty = (t / TS) % TS, tz = t % TS;
      const int qx = tx >> 2, qy = ty >> 2, qz = tz >> 2;
      const uint32_t fs = fat_slot[morton3(bx + qx, by + qy, bz + qz)];
      tile[t] = gridv[(size_t)fs * BC + (((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3))];
    }
    __syncthreads();
    const float ox = (float)(bx * BS), oy = (float)(by * BS), oz = (float)(bz * BS);
    const uint32_t p0 = act_start[a], p1 = act_start[a + 1];
    for (uint32_t p = p0 + tid; p < p1; p += NT) {
      const GroupParams g = groups[s.gid[p]];
      const float x0 = s.f[FX][p], x1 = s.f[FX + 1][p], x2 = s.f[FX + 2][p];
      const float X0 = x0 * P.idx - ox, X1 = x1 * P.idx - oy, X2 = x2 * P.idx - oz;
      const int c0 = (int)(X0 - 0.5f), c1 = (int)(X1 - 0.5f), c2 = (int)(X2 - 0.5f);
      const float r0 = X0 - (float)c0, r1 = X1 - (float)c1, r2 = X2 - (float)c2;
      float w0[3], w1[3], w2[3];
      bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
      float v0 = 0, v1 = 0, v2 = 0;
      mat3 b;
#pragma unroll
      for (int k = 0; k < 9; k++) b.m[k] = 0.0f;
      const int nbase = (c0 * TS + c1) * TS + c2;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float d0 = r0 - (float)i;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const float d1 = r1 - (float)j;
          const float wij = w0[i] * w1[j];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const float d2 = r2 - (float)k;
            const float w = wij * w2[k];
            const float4 gv = tile[nbase + (i * TS + j) * TS + k];
            // :898-903  v_ = fma(grid_vel, w, v_);  b_[r] = fma(w*grid_vel, dpos[r], b_[r])
            v0 = fmaf(gv.x, w, v0); v1 = fmaf(gv.y, w, v1); v2 = fmaf(gv.z, w, v2);
            const float a0 = w * gv.x, a1 = w * gv.y, a2 = w * gv.z;
            b(0, 0) = fmaf(a0, d0, b(0, 0)); b(0, 1) = fmaf(a0, d1, b(0, 1)); b(0, 2) = fmaf(a0, d2, b(0, 2));
            b(1, 0) = fmaf(a1, d0, b(1, 0)); b(1, 1) = fmaf(a1, d1, b(1, 1)); b(1, 2) = fmaf(a1, d2, b(1, 2));
            b(2, 0) = fmaf(a2, d0, b(2, 0)); b(2, 1) = fmaf(a2, d1, b(2, 1)); b(2, 2) = fmaf(a2, d2, b(2, 2));
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep at most one i-plane (9 float4 LDS reads) in flight: VGPR pressure
      }
      // apic_b = damp_affine_momemtum(b) (src/mpm.h:465-469); the reference's optimised path has a bug
      // here (passes the block index, transfer.cpp:925-926) — we implement the intended damping.
      if (P.rpic_damping != 0.0f || P.apic_damping != 0.0f) {
        const float ks = 1.0f - P.rpic_damping, ka = 1.0f - P.apic_damping;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float sym = 0.5f * (b(r, c) + b(c, r));
            s.f[FB + 3 * r + c][p] = ks * sym + ka * (b(r, c) - sym);
          }
      } else {
#pragma unroll
        for (int k = 0; k < 9; k++) s.f[FB + k][p] = b.m[k];
      }
      s.f[FV][p] = v0; s.f[FV + 1][p] = v1; s.f[FV + 2][p] = v2;
      mat3 cdg;  // :940-942  cdg = I + (-4 inv_dx dt) b
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) cdg(r, c) = fmaf(scale, b(r, c), (r == c) ? 1.0f : 0.0f);
      mat3 F;
#pragma unroll
      for (int k = 0; k < 9; k++) F.m[k] = s.f[FF + k][p];
      float aux = s.f[FAUX][p];
      plasticity(g, cdg, F, aux);  // :950
      if (g.type != MPMHIP_WATER) {
#pragma unroll
        for (int k = 0; k < 9; k++) s.f[FF + k][p] = F.m[k];
      }
      s.f[FAUX][p] = aux;
      s.f[FX][p] = fmaf(v0, P.dt, x0);  // :951
      s.f[FX + 1][p] = fmaf(v1, P.dt, x1);
      s.f[FX + 2][p] = fmaf(v2, P.dt, x2);
    }
    __syncthreads();
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
__global__ void k_debug_plasticity(GroupParams g, int64_t n, const float *cdg, float *F, float *aux) {
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mat3 f, c;
    for (int k = 0; k < 9; k++) { f.m[k] = F[9 * i + k]; c.m[k] = cdg[9 * i + k]; }
    float a = aux[i];
    plasticity(g, c, f, a);
    for (int k = 0; k < 9; k++) F[9 * i + k] = f.m[k];
    aux[i] = a;
  }
}

}  // namespace mpm

// ================================================================================================ host side
using namespace mpm;

static thread_local std::string g_create_error;

enum { PH_SORT = 0, PH_P2G = 1, PH_GRID = 2, PH_G2P = 3, PH_COUNT = 4 };

struct mpmhip_ctx {
  mpmhip_config cfg;
  Params P;
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;
  // particles
  int64_t cap = 0;
  int64_t n_host = 0;     // upper bound of live particles (exact until something is deleted)
  int32_t next_pid = 0;
  SoA soa[2];
  int cur = 0;
  float *pool_f[2] = {nullptr, nullptr};
  uint8_t *pool_g[2] = {nullptr, nullptr};
  int32_t *pool_i[2] = {nullptr, nullptr};
  uint32_t *key = nullptr, *rank = nullptr;
  // blocks
  uint32_t NB = 0;
  uint8_t *blk_flag = nullptr;
  uint32_t *dest = nullptr;
  uint32_t *bits = nullptr, *wprefix = nullptr, *act_blk = nullptr, *act_start = nullptr, *totals = nullptr;
  uint32_t *cell_cnt = nullptr, *cell_start = nullptr, *partials = nullptr, *fat_slot = nullptr;
  float4 *tiles = nullptr, *gridv = nullptr, *dense = nullptr;
  Counters *cnt = nullptr;
  std::vector<GroupParams> groups;
  GroupParams *d_groups = nullptr;
  int groups_cap = 256;
  bool sorted = false;
  int g2p_minw = 2;  // tuning knob (env MPMHIP_G2P_MINW): __launch_bounds__ waves/SIMD of k_g2p
  float t = 0.0f, request_t = 0.0f;  // `real` accumulators, as in the reference (src/mpm.h:99, mpm.cpp:573)
  int64_t substeps = 0;
  // profiling
  bool profiling = false;
  struct Ev { hipEvent_t e[PH_COUNT + 1]; };
  std::vector<Ev> ev_pool;
  size_t ev_used = 0;
  double phase_ms[PH_COUNT] = {0, 0, 0, 0};
  int64_t prof_substeps = 0;
};

static int fail(mpmhip_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(c, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) return fail((c), MPMHIP_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
static hipError_t dmalloc(T **p, size_t count) { return hipMalloc((void **)p, count * sizeof(T)); }

static void bind_soa(mpmhip_ctx *c) {
  for (int s = 0; s < 2; s++) {
    for (int f = 0; f < NF; f++) c->soa[s].f[f] = c->pool_f[s] + (size_t)f * c->cap;
    c->soa[s].gid = c->pool_g[s];
    c->soa[s].pid = c->pool_i[s];
  }
}

static int particle_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (int)b;
}

static int launch_check(mpmhip_ctx *c, const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, MPMHIP_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return MPMHIP_OK;
}

template <typename K, typename... Args>
static int run_debug(mpmhip_ctx *c, K kernel, Args... args) {
  hipLaunchKernelGGL(kernel, dim3(256), dim3(256), 0, c->stream, args...);
  int rc = launch_check(c, "debug kernel");
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPMHIP_OK;
}

extern "C" {

uint32_t mpmhip_abi_version(void) { return MPMHIP_ABI_VERSION; }

const char *mpmhip_last_error(const mpmhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mpmhip_create(const mpmhip_config *cfg, mpmhip_ctx **out) {
  if (!cfg || !out) return fail(nullptr, MPMHIP_EINVAL, "null argument");
  *out = nullptr;
  for (int k = 0; k < 3; k++)
    if (cfg->res[k] < 8 || cfg->res[k] > 1000) return fail(nullptr, MPMHIP_EINVAL, "res[%d]=%d outside [8,1000]", k, cfg->res[k]);
  if (!(cfg->dx > 0) || !(cfg->dt >= 0)) return fail(nullptr, MPMHIP_EINVAL, "dx must be > 0 and dt >= 0");
  if (cfg->max_particles <= 0 || cfg->max_particles >= (1ll << 31)) return fail(nullptr, MPMHIP_EINVAL, "max_particles out of range");
  if (cfg->n_planes < 0 || cfg->n_planes > 8) return fail(nullptr, MPMHIP_EINVAL, "n_planes out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MPMHIP_EHIP, "no HIP device available (libmpmhip has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MPMHIP_EINVAL, "device %d of %d", cfg->device, ndev);
  mpmhip_ctx *c = new (std::nothrow) mpmhip_ctx();
  if (!c) return fail(nullptr, MPMHIP_ENOMEM, "host allocation failed");
  c->cfg = *cfg;
  c->device = cfg->device;
  if (const char *e = getenv("MPMHIP_G2P_MINW")) c->g2p_minw = atoi(e);
  auto bail = [&](int code) { g_create_error = c->err; mpmhip_destroy(c); return code; };
  if (hipSetDevice(c->device) != hipSuccess) { fail(c, MPMHIP_EHIP, "hipSetDevice failed"); return bail(MPMHIP_EHIP); }
  Params &P = c->P;
  memset(&P, 0, sizeof P);
  int maxnb = 0;
  for (int k = 0; k < 3; k++) {
    P.res[k] = cfg->res[k];
    P.g[k] = cfg->gravity[k];
    int nb = (cfg->res[k] + 1 + BS - 1) / BS + 1;
    if (nb > maxnb) maxnb = nb;
  }
  P.dx = cfg->dx; P.idx = 1.0f / cfg->dx; P.dt = cfg->dt;
  P.particle_gravity = cfg->particle_gravity; P.apic_damping = cfg->apic_damping; P.rpic_damping = cfg->rpic_damping;
  P.clean_boundary = cfg->clean_boundary; P.n_planes = cfg->n_planes; P.friction = cfg->friction;
  memcpy(P.planes, cfg->planes, sizeof P.planes);
  int kbits = 1;
  while ((1 << kbits) < maxnb) kbits++;
  if (kbits > 8) { fail(c, MPMHIP_EINVAL, "grid too large for 32-bit keys"); return bail(MPMHIP_EINVAL); }
  P.kbits = kbits;
  c->NB = 1u << (3 * kbits);
  P.nbw = c->NB / 32u;
  if (P.nbw == 0) P.nbw = 1;
  c->cap = cfg->max_particles;
  int64_t mb = cfg->max_blocks;
  if (mb <= 0) {
    mb = c->cap / 48 + 4096;  // a block of 64 cells at >= ~1 particle/cell on average, plus slack
  }
  if (mb > (int64_t)c->NB) mb = c->NB;
  P.max_blocks = (uint32_t)mb;

  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  for (int s = 0; s < 2; s++) {
    A(dmalloc(&c->pool_f[s], (size_t)NF * c->cap));
    A(dmalloc(&c->pool_g[s], (size_t)c->cap));
    A(dmalloc(&c->pool_i[s], (size_t)c->cap));
  }
  A(dmalloc(&c->key, (size_t)c->cap));
  A(dmalloc(&c->rank, (size_t)c->cap));
  A(dmalloc(&c->bits, (size_t)P.nbw));
  A(dmalloc(&c->blk_flag, (size_t)P.nbw * 32));
  A(dmalloc(&c->dest, (size_t)c->cap));
  A(dmalloc(&c->wprefix, (size_t)P.nbw));
  A(dmalloc(&c->fat_slot, (size_t)c->NB));
  A(dmalloc(&c->act_blk, (size_t)mb + 1));
  A(dmalloc(&c->act_start, (size_t)mb + 2));
  A(dmalloc(&c->totals, (size_t)mb + 1));
  A(dmalloc(&c->cell_cnt, (size_t)mb * BC));
  A(dmalloc(&c->cell_start, (size_t)mb * BC + 1));
  A(dmalloc(&c->partials, (size_t)((P.nbw > (uint32_t)mb ? P.nbw : (uint32_t)mb) / SCAN_CHUNK + 2)));
  A(dmalloc(&c->tiles, (size_t)mb * TN));
  A(dmalloc(&c->gridv, (size_t)mb * 8 * BC));
  A(dmalloc(&c->cnt, 1));
  A(dmalloc(&c->d_groups, (size_t)c->groups_cap));
  if (e != hipSuccess) {
    fail(c, MPMHIP_ENOMEM, "device allocation failed: %s (max_particles=%lld, max_blocks=%lld)", hipGetErrorString(e),
         (long long)c->cap, (long long)mb);
    return bail(MPMHIP_ENOMEM);
  }
  bind_soa(c);
  A(hipMemset(c->bits, 0, sizeof(uint32_t) * P.nbw));
  A(hipMemset(c->blk_flag, 0, (size_t)P.nbw * 32));
  A(hipMemset(c->cell_cnt, 0, sizeof(uint32_t) * (size_t)mb * BC));
  A(hipMemset(c->cnt, 0, sizeof(Counters)));
  A(hipMemset(c->cell_start, 0, sizeof(uint32_t) * ((size_t)mb * BC + 1)));
  A(hipMemset(c->act_start, 0, sizeof(uint32_t) * ((size_t)mb + 2)));
  A(hipMemset(c->fat_slot, 0, sizeof(uint32_t) * (size_t)c->NB));
  A(hipDeviceSynchronize());
  if (e != hipSuccess) { fail(c, MPMHIP_EHIP, "device init failed: %s", hipGetErrorString(e)); return bail(MPMHIP_EHIP); }
  *out = c;
  return MPMHIP_OK;
}

void mpmhip_destroy(mpmhip_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->own_stream) hipStreamSynchronize(c->own_stream);
  for (auto &ev : c->ev_pool)
    for (int k = 0; k <= PH_COUNT; k++) hipEventDestroy(ev.e[k]);
  for (int s = 0; s < 2; s++) { hipFree(c->pool_f[s]); hipFree(c->pool_g[s]); hipFree(c->pool_i[s]); }
  hipFree(c->key); hipFree(c->rank); hipFree(c->blk_flag); hipFree(c->dest); hipFree(c->bits); hipFree(c->wprefix); hipFree(c->fat_slot);
  hipFree(c->act_blk); hipFree(c->act_start); hipFree(c->totals); hipFree(c->cell_cnt); hipFree(c->cell_start); hipFree(c->partials);
  hipFree(c->tiles); hipFree(c->gridv); hipFree(c->dense); hipFree(c->cnt); hipFree(c->d_groups);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}

int mpmhip_set_stream(mpmhip_ctx *c, void *s) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return MPMHIP_OK;
}

int mpmhip_set_levelset(mpmhip_ctx *c, int32_t n_planes, const float *planes, float friction) {
  if (!c || n_planes < 0 || n_planes > 8 || (n_planes > 0 && !planes)) return MPMHIP_EINVAL;
  c->P.n_planes = n_planes;
  c->cfg.n_planes = n_planes;
  for (int i = 0; i < n_planes; i++)
    for (int k = 0; k < 4; k++) c->P.planes[i][k] = c->cfg.planes[i][k] = planes[4 * i + k];
  c->P.friction = c->cfg.friction = friction;
  return MPMHIP_OK;
}

int mpmhip_add_group(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM]) {
  if (!c || !params) return MPMHIP_EINVAL;
  if (material == MPMHIP_VISCO) return fail(c, MPMHIP_ENOTIMPL, "material 'visco' is not implemented on the device path yet");
  if (material < MPMHIP_SNOW || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
  if ((int)c->groups.size() >= c->groups_cap) return fail(c, MPMHIP_ECAPACITY, "too many particle groups (max %d)", c->groups_cap);
  if (!(params[0] > 0) || !(params[1] > 0)) return fail(c, MPMHIP_EINVAL, "group mass and vol must be > 0");
  GroupParams g;
  memset(&g, 0, sizeof g);
  memcpy(g.p, params, sizeof g.p);
  g.type = material;
  c->groups.push_back(g);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->d_groups, c->groups.data(), sizeof(GroupParams) * c->groups.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return (int)c->groups.size() - 1;
}

static int refresh_count(mpmhip_ctx *c) {
  Counters h;
  HIPCHK(c, hipMemcpyAsync(&h, c->cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h.error & 1u)
    return fail(c, MPMHIP_ECAPACITY, "active blocks (%u) exceed max_blocks (%u): recreate the ctx with a larger max_blocks",
                h.n_active, c->P.max_blocks);
  c->n_host = h.n;
  return MPMHIP_OK;
}

int mpmhip_add_particles(mpmhip_ctx *c, int32_t group, int64_t n, const float *x, const float *v, const float *F,
                         const float *B, const float *aux) {
  if (!c || n < 0 || (n > 0 && !x)) return MPMHIP_EINVAL;
  if (group < 0 || group >= (int)c->groups.size()) return fail(c, MPMHIP_EINVAL, "unknown group %d", group);
  if (n == 0) return MPMHIP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = refresh_count(c);
  if (rc) return rc;
  if (c->n_host + n > c->cap)
    return fail(c, MPMHIP_ECAPACITY, "particle capacity exceeded: %lld + %lld > %lld", (long long)c->n_host, (long long)n, (long long)c->cap);
  const int mat = c->groups[group].type;
  const float aux0 = (mat == MPMHIP_SNOW || mat == MPMHIP_WATER) ? 1.0f : 0.0f;  // Jp = 1 (:204), j = 1 (:460), logJp = 0 (:595)
  std::vector<float> stage((size_t)n);
  SoA &s = c->soa[c->cur];
  for (int f = 0; f < NF; f++) {
    for (int64_t i = 0; i < n; i++) {
      float val;
      if (f < FV) val = x[3 * i + f];
      else if (f < FB) val = v ? v[3 * i + (f - FV)] : 0.0f;
      else if (f < FF) val = B ? B[9 * i + (f - FB)] : 0.0f;
      else if (f < FAUX) val = F ? F[9 * i + (f - FF)] : (((f - FF) % 4 == 0) ? 1.0f : 0.0f);
      else val = aux ? aux[i] : aux0;
      stage[i] = val;
    }
    HIPCHK(c, hipMemcpy(s.f[f] + c->n_host, stage.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  }
  std::vector<uint8_t> gs((size_t)n, (uint8_t)group);
  std::vector<int32_t> ids((size_t)n);
  for (int64_t i = 0; i < n; i++) ids[i] = c->next_pid + (int32_t)i;
  c->next_pid += (int32_t)n;
  HIPCHK(c, hipMemcpy(s.gid + c->n_host, gs.data(), n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(s.pid + c->n_host, ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
  c->n_host += n;
  uint32_t nn = (uint32_t)c->n_host;
  HIPCHK(c, hipMemcpy(&c->cnt->n, &nn, sizeof nn, hipMemcpyHostToDevice));
  c->sorted = false;
  return MPMHIP_OK;
}

int64_t mpmhip_num_particles(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  if (hipSetDevice(c->device) != hipSuccess) return MPMHIP_EHIP;
  int rc = refresh_count(c);
  return rc ? rc : c->n_host;
}

static int field_info(int32_t field, int &f0, int &width) {
  switch (field) {
    case MPMHIP_F_X: f0 = FX; width = 3; return 0;
    case MPMHIP_F_V: f0 = FV; width = 3; return 0;
    case MPMHIP_F_B: f0 = FB; width = 9; return 0;
    case MPMHIP_F_F: f0 = FF; width = 9; return 0;
    case MPMHIP_F_AUX: f0 = FAUX; width = 1; return 0;
  }
  return -1;
}

int mpmhip_download(mpmhip_ctx *c, int32_t field, void *dst, int64_t n_capacity) {
  if (!c || !dst) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = refresh_count(c);
  if (rc) return rc;
  const int64_t n = c->n_host;
  if (n > n_capacity) return fail(c, MPMHIP_ECAPACITY, "download buffer holds %lld particles, need %lld", (long long)n_capacity, (long long)n);
  SoA &s = c->soa[c->cur];
  if (field == MPMHIP_F_GID) {
    std::vector<uint8_t> g((size_t)n);
    HIPCHK(c, hipMemcpy(g.data(), s.gid, n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) ((int32_t *)dst)[i] = g[i];
    return (int)n;
  }
  if (field == MPMHIP_F_ID) {
    HIPCHK(c, hipMemcpy(dst, s.pid, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return (int)n;
  }
  int f0, width;
  if (field_info(field, f0, width)) return fail(c, MPMHIP_EINVAL, "unknown field %d", field);
  std::vector<float> stage((size_t)n);
  for (int k = 0; k < width; k++) {
    HIPCHK(c, hipMemcpy(stage.data(), s.f[f0 + k], sizeof(float) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) ((float *)dst)[i * width + k] = stage[i];
  }
  return (int)n;
}

int mpmhip_upload(mpmhip_ctx *c, int32_t field, const void *src, int64_t n) {
  if (!c || !src) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = refresh_count(c);
  if (rc) return rc;
  if (n != c->n_host) return fail(c, MPMHIP_EINVAL, "upload of %lld records but the ctx holds %lld particles", (long long)n, (long long)c->n_host);
  SoA &s = c->soa[c->cur];
  if (field == MPMHIP_F_ID) {  // restores creation ids after a caller-side re-allocation
    const int32_t *ids = (const int32_t *)src;
    int32_t mx = -1;
    for (int64_t i = 0; i < n; i++) mx = ids[i] > mx ? ids[i] : mx;
    HIPCHK(c, hipMemcpy(s.pid, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    if (mx + 1 > c->next_pid) c->next_pid = mx + 1;
    return MPMHIP_OK;
  }
  int f0, width;
  if (field_info(field, f0, width)) return fail(c, MPMHIP_EINVAL, "field %d cannot be uploaded", field);
  std::vector<float> stage((size_t)n);
  for (int k = 0; k < width; k++) {
    for (int64_t i = 0; i < n; i++) stage[i] = ((const float *)src)[i * width + k];
    HIPCHK(c, hipMemcpy(s.f[f0 + k], stage.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  }
  if (field == MPMHIP_F_X) c->sorted = false;
  return MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ phases
static int do_sort(mpmhip_ctx *c) {
  const Params &P = c->P;
  hipStream_t st = c->stream;
  const int pg = particle_grid(c->n_host);
  SoA &src = c->soa[c->cur], &dst = c->soa[c->cur ^ 1];
  hipLaunchKernelGGL(k_build_keys, dim3(pg), dim3(256), 0, st, P, src, c->cnt, c->key, c->blk_flag);
  hipLaunchKernelGGL(k_pack_flags, dim3((P.nbw + 255) / 256), dim3(256), 0, st, P, c->blk_flag, c->bits);
  const int nb_chunks = (int)((P.nbw + SCAN_CHUNK - 1) / SCAN_CHUNK);
  const int na_chunks = (int)((P.max_blocks + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipLaunchKernelGGL((k_scan_partials<0>), dim3(nb_chunks), dim3(256), 0, st, P, c->cnt, c->bits, c->partials);
  hipLaunchKernelGGL((k_scan_apply<0>), dim3(nb_chunks), dim3(256), 0, st, P, c->cnt, c->bits, c->partials, c->wprefix);
  hipLaunchKernelGGL(k_emit_active, dim3((P.nbw + 255) / 256), dim3(256), 0, st, P, c->bits, c->wprefix, c->act_blk);
  hipLaunchKernelGGL(k_rank, dim3(pg), dim3(256), 0, st, P, c->cnt, c->key, c->rank, c->cell_cnt, c->bits, c->wprefix);
  hipLaunchKernelGGL(k_block_totals, dim3(1024), dim3(256), 0, st, c->cnt, c->cell_cnt, c->totals);
  hipLaunchKernelGGL((k_scan_partials<1>), dim3(na_chunks), dim3(256), 0, st, P, c->cnt, c->totals, c->partials);
  hipLaunchKernelGGL((k_scan_apply<1>), dim3(na_chunks), dim3(256), 0, st, P, c->cnt, c->totals, c->partials, c->act_start);
  hipLaunchKernelGGL(k_cell_start, dim3(1024), dim3(256), 0, st, P, c->cnt, c->cell_cnt, c->act_start, c->cell_start);
  hipLaunchKernelGGL(k_layout, dim3(1024), dim3(256), 0, st, P, c->cnt, c->cell_start, c->dest);
  hipLaunchKernelGGL(k_reorder, dim3(pg), dim3(256), 0, st, c->cnt, src, dst, c->key, c->rank, c->cell_start, c->dest);
  hipLaunchKernelGGL(k_sort_cleanup, dim3(1024), dim3(256), 0, st, P, c->cnt, c->cell_cnt);
  c->cur ^= 1;
  c->sorted = true;
  return launch_check(c, "sort");
}

static int do_p2g(mpmhip_ctx *c) {
  hipLaunchKernelGGL(k_p2g, dim3(8192), dim3(64), 0, c->stream, c->P, c->soa[c->cur], c->cnt, c->act_blk,
                     c->cell_start, c->d_groups, c->tiles);
  return launch_check(c, "p2g");
}
static int do_grid(mpmhip_ctx *c, int mode) {
  hipLaunchKernelGGL(k_grid, dim3(16384), dim3(64), 0, c->stream, c->P, mode, c->cnt, c->act_blk, c->bits, c->wprefix,
                     c->tiles, c->gridv, c->fat_slot, c->dense);
  return launch_check(c, "grid");
}
static int do_g2p(mpmhip_ctx *c) {
  auto kern = c->g2p_minw == 4 ? k_g2p<256, 4> : (c->g2p_minw == 3 ? k_g2p<256, 3> : k_g2p<256, 2>);
  hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, c->stream, c->P, c->soa[c->cur], c->cnt, c->act_blk,
                     c->act_start, c->d_groups, c->gridv, c->fat_slot);
  return launch_check(c, "g2p");
}

static int need_sorted(mpmhip_ctx *c, const char *who) {
  if (!c->sorted) return fail(c, MPMHIP_EINVAL, "%s needs sorted particles: call mpmhip_sort first", who);
  return MPMHIP_OK;
}

int mpmhip_sort(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  return do_sort(c);
}
int mpmhip_p2g(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "p2g");
  return rc ? rc : do_p2g(c);
}
int mpmhip_grid_update(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "grid_update");
  return rc ? rc : do_grid(c, 0);
}
int mpmhip_g2p(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "g2p");
  if (rc) return rc;
  rc = do_g2p(c);
  c->sorted = false;  // positions moved: the next phase-level p2g needs a new sort
  return rc;
}

static int get_events(mpmhip_ctx *c, mpmhip_ctx::Ev **out) {
  if (c->ev_used == c->ev_pool.size()) {
    mpmhip_ctx::Ev ev;
    for (int k = 0; k <= PH_COUNT; k++) HIPCHK(c, hipEventCreate(&ev.e[k]));
    c->ev_pool.push_back(ev);
  }
  *out = &c->ev_pool[c->ev_used++];
  return MPMHIP_OK;
}

static int collect_events(mpmhip_ctx *c) {
  if (c->ev_used == 0) return MPMHIP_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->ev_used; i++) {
    for (int k = 0; k < PH_COUNT; k++) {
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[k], c->ev_pool[i].e[k + 1]));
      c->phase_ms[k] += ms;
    }
    c->prof_substeps++;
  }
  c->ev_used = 0;
  return MPMHIP_OK;
}

int mpmhip_substep(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  mpmhip_ctx::Ev *ev = nullptr;
  if (c->profiling) {
    if (c->ev_used >= 4096 && (rc = collect_events(c))) return rc;
    if ((rc = get_events(c, &ev))) return rc;
    HIPCHK(c, hipEventRecord(ev->e[0], c->stream));
  }
  if ((rc = do_sort(c))) return rc;
  if (ev) HIPCHK(c, hipEventRecord(ev->e[1], c->stream));
  if ((rc = do_p2g(c))) return rc;
  if (ev) HIPCHK(c, hipEventRecord(ev->e[2], c->stream));
  if ((rc = do_grid(c, 0))) return rc;
  if (ev) HIPCHK(c, hipEventRecord(ev->e[3], c->stream));
  if ((rc = do_g2p(c))) return rc;
  if (ev) HIPCHK(c, hipEventRecord(ev->e[4], c->stream));
  c->sorted = false;
  c->t += c->P.dt;  // src/mpm.cpp:573
  c->substeps++;
  return MPMHIP_OK;
}

int mpmhip_run_substeps(mpmhip_ctx *c, int32_t n) {
  for (int32_t i = 0; i < n; i++) {
    int rc = mpmhip_substep(c);
    if (rc) return rc;
  }
  return MPMHIP_OK;
}

int mpmhip_step(mpmhip_ctx *c, float dt) {  // MPM<dim>::step, src/mpm.cpp:428-439
  if (!c) return MPMHIP_EINVAL;
  if (dt < 0) {
    int rc = mpmhip_substep(c);
    c->request_t = c->t;
    return rc;
  }
  c->request_t += dt;
  while (c->t + c->P.dt < c->request_t) {
    int rc = mpmhip_substep(c);
    if (rc) return rc;
  }
  return MPMHIP_OK;
}

double mpmhip_current_time(const mpmhip_ctx *c) { return c ? (double)c->t : 0.0; }

int mpmhip_synchronize(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  Counters h;
  HIPCHK(c, hipMemcpy(&h, c->cnt, sizeof h, hipMemcpyDeviceToHost));
  if (h.error & 1u)
    return fail(c, MPMHIP_ECAPACITY, "active blocks (%u) exceed max_blocks (%u)", h.n_active, c->P.max_blocks);
  return MPMHIP_OK;
}

static int ensure_dense(mpmhip_ctx *c, size_t &nodes) {
  nodes = (size_t)(c->P.res[0] + 1) * (c->P.res[1] + 1) * (c->P.res[2] + 1);
  if (!c->dense) {
    if (dmalloc(&c->dense, nodes) != hipSuccess) return fail(c, MPMHIP_ENOMEM, "dense grid staging allocation failed");
  }
  return MPMHIP_OK;
}

int mpmhip_download_grid(mpmhip_ctx *c, int32_t which, float *dst) {
  if (!c || !dst || (which != 0 && which != 1)) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  size_t nodes;
  int rc = ensure_dense(c, nodes);
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->dense, 0, nodes * sizeof(float4), c->stream));
  if ((rc = do_grid(c, which == 0 ? 1 : 3))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(dst, c->dense, nodes * sizeof(float4), hipMemcpyDeviceToHost));
  return MPMHIP_OK;
}

int mpmhip_upload_grid(mpmhip_ctx *c, const float *src) {
  if (!c || !src) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "upload_grid");
  if (rc) return rc;
  size_t nodes;
  if ((rc = ensure_dense(c, nodes))) return rc;
  HIPCHK(c, hipMemcpy(c->dense, src, nodes * sizeof(float4), hipMemcpyHostToDevice));
  return do_grid(c, 2);
}

int mpmhip_set_profiling(mpmhip_ctx *c, int32_t enabled) {
  if (!c) return MPMHIP_EINVAL;
  c->profiling = enabled != 0;
  return MPMHIP_OK;
}
int mpmhip_profile_reset(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = collect_events(c);
  for (int k = 0; k < PH_COUNT; k++) c->phase_ms[k] = 0;
  c->prof_substeps = 0;
  return rc;
}
int mpmhip_profile(mpmhip_ctx *c, char *json, size_t cap) {
  if (!c || !json || cap == 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = collect_events(c);
  if (rc) return rc;
  Counters h;
  HIPCHK(c, hipMemcpy(&h, c->cnt, sizeof h, hipMemcpyDeviceToHost));
  int w = snprintf(json, cap,
                   "{\"substeps\":%lld,\"particles\":%u,\"active_blocks\":%u,\"phases\":{\"sort\":%.6f,\"p2g\":%.6f,"
                   "\"grid\":%.6f,\"g2p\":%.6f}}",
                   (long long)c->prof_substeps, h.n, h.n_active, c->phase_ms[0], c->phase_ms[1], c->phase_ms[2], c->phase_ms[3]);
  return (w < 0 || (size_t)w >= cap) ? fail(c, MPMHIP_EINVAL, "profile buffer too small") : MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ debug math
int mpmhip_debug_svd3(mpmhip_ctx *c, int64_t n, const float *F, float *U, float *S, float *V) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  float *dF, *dU, *dS, *dV;
  HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dU, 9 * n)); HIPCHK(c, dmalloc(&dS, 3 * n)); HIPCHK(c, dmalloc(&dV, 9 * n));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  int rc = run_debug(c, k_debug_svd, n, (const float *)dF, dU, dS, dV);
  if (!rc) {
    HIPCHK(c, hipMemcpy(U, dU, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(S, dS, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(V, dV, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
  }
  hipFree(dF); hipFree(dU); hipFree(dS); hipFree(dV);
  return rc;
}

static int make_group(mpmhip_ctx *c, int32_t material, const float *params, GroupParams &g) {
  if (material < MPMHIP_SNOW || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
  memset(&g, 0, sizeof g);
  memcpy(g.p, params, sizeof g.p);
  g.type = material;
  return MPMHIP_OK;
}

int mpmhip_debug_force(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM], int64_t n, const float *F,
                       const float *aux, float *out) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  GroupParams g;
  int rc = make_group(c, material, params, g);
  if (rc) return rc;
  float *dF, *dA, *dO;
  HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dA, n)); HIPCHK(c, dmalloc(&dO, 9 * n));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dA, aux, sizeof(float) * n, hipMemcpyHostToDevice));
  rc = run_debug(c, k_debug_force, g, n, (const float *)dF, (const float *)dA, dO);
  if (!rc) HIPCHK(c, hipMemcpy(out, dO, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
  hipFree(dF); hipFree(dA); hipFree(dO);
  return rc;
}

int mpmhip_debug_plasticity(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM], int64_t n,
                            const float *cdg, float *F, float *aux) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  GroupParams g;
  int rc = make_group(c, material, params, g);
  if (rc) return rc;
  float *dC, *dF, *dA;
  HIPCHK(c, dmalloc(&dC, 9 * n)); HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dA, n));
  HIPCHK(c, hipMemcpy(dC, cdg, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dA, aux, sizeof(float) * n, hipMemcpyHostToDevice));
  rc = run_debug(c, k_debug_plasticity, g, n, (const float *)dC, dF, dA);
  if (!rc) {
    HIPCHK(c, hipMemcpy(F, dF, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(aux, dA, sizeof(float) * n, hipMemcpyDeviceToHost));
  }
  hipFree(dC); hipFree(dF); hipFree(dA);
  return rc;
}

}  // extern "C"
