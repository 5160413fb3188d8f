// taichi_mpm_amd/csrc/mpmhip.hip — MI355X (gfx950) MLS-MPM time-stepping core: HIP kernels + C ABI.
//
// One substep (reference: MPM<3>::substep, src/mpm.cpp:452-575):
//
//   sort   (index sort; replaces sort_particles_and_populate_grid src/mpm.cpp:770-918 and
//           clear_boundary_particles :582-633 — dead particles simply drop out of the index)
//          [k_build_keys]  key = Morton(block) << 6 | cell-in-block per particle (normally produced by the
//                          previous substep's k_g2p, which knows the new position)
//          k_block_table  byte flags -> active-block bitmap + popcount prefix (dense slot of every active block)
//                         + active-block list, one single-pass chained-scan launch
//          k_rank         rank of each particle in its cell (run-aggregated atomics)
//          k_cell_table   per-cell counts -> start of every cell / block in the sorted index (single pass)
//          k_perm         sorted position -> particle slot
//   P2G    k_p2g   one wavefront per active 4x4x4-cell block, ONE LANE PER CELL: register accumulation of the
//                  27x4 node contributions over the cell's particles (records prefetched two particles ahead),
//                  ordered non-atomic float4 merge into the block's 6^3-node LDS tile, tile written out whole
//                                                                                (src/transfer.cpp:467-569)
//   grid   k_grid  sums the <=8 overlapping block tiles of every touched grid block, normalises,
//                  gravity + level-set boundary                                  (src/mpm.cpp:277-372)
//   G2P    k_g2p   6^3 velocity tile in LDS, 27-tap gather, F update, plasticity AND the next substep's stress
//                  from one eigen-solve, advection, next key                     (src/transfer.cpp:837-954)
//
// Data layout (fp32, resident in HBM for the life of the ctx):
//   particles  two arrays of 64-byte records indexed by a stable particle slot (the reference's
//              ParticleAllocator pool index, src/particle_allocator.h:36):
//                RecG {x3, aux, F9, gid, pid, -}   what G2P reads and rewrites
//                RecP {x3, v3, A9, mass}           what P2G reads;  A = stress*(-4 inv_dx dt) + apic_b*(4 m)
//              (src/transfer.cpp:521-522) is produced by G2P, so P2G carries no constitutive work, and
//              apic_b itself goes to a side array (it is only ever consumed through A).
//              Particles never move in memory between substeps: `perm` (sorted position -> slot) is rebuilt
//              every substep and both transfer kernels gather whole 64-byte records through it, one record per
//              lane.  This mirrors the reference's own structure — sorted index array every substep
//              (mpm.cpp:785-807) + physical reorder only every `reorder_interval` substeps (:811-813).
//   blocks     an "active" block = 4x4x4 cells holding >=1 particle; bitmap over the Morton block space +
//              per-word popcount prefix -> dense slot = rank in Morton order, no pass over the whole grid.
//   tiles      float4[216] per active block: its private (4+2)^3-node P2G result.
//   gridv      float4[64] per touched grid block (v.xyz, m); slot = 8a+o of the owning (active block a,
//              corner offset o); `fat_slot` maps Morton block -> slot.
// No float atomics anywhere (LDS float atomics cost ~2 cycles per lane on gfx950; global ones leave the L2).

#include <hip/hip_runtime.h>

#include <algorithm>
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

// 64-byte particle records (4 x float4)
struct alignas(16) RecG {  // G2P side
  float x[3];
  float aux;
  float F[9];
  uint32_t gid;
  int32_t pid;  // creation id; < 0 marks a deleted slot
  uint32_t pad;
};
struct alignas(16) RecP {  // P2G side
  float x[3];
  float v[3];
  float A[9];
  float mass;  // group mass (get_mass()), so P2G needs no group-table lookup
};
static_assert(sizeof(RecG) == 64 && sizeof(RecP) == 64, "records must be 64 bytes");
constexpr int BW = 12;  // floats per apic_b record (9 used): three float4

struct Counters {
  uint32_t n_sorted;  // live particles in the current sorted index
  uint32_t n_active;  // active blocks
  uint32_t n_dead;    // slots marked deleted so far
  uint32_t error;     // bit0: active blocks exceeded max_blocks
};

struct Params {
  int res[3];
  float dx, idx, dt;
  float g[3];
  int particle_gravity;
  float apic_damping, rpic_damping;
  int clean_boundary;
  int kbits;         // Morton bits per axis
  uint32_t nbw;      // bitmap words = 8^kbits / 32
  uint32_t max_blocks;
  uint32_t n_slots;  // particle slots in use (host-known)
  int store_b;       // keep apic_b in the side array
  int ablate;        // PROFILING ONLY (env MPMHIP_ABLATE, results invalid): 1 no G2P stores, 2 no constitutive
                     // update, 4 no 27-tap gather
};

// multi-GPU tiling (include/mpmhip.h, "Multi-GPU tiling"): partition of the cell space into bricks + halo boxes
struct Tiling {
  int enabled, rank;
  int dims[3];
  int cuts[3][MPMHIP_MAX_PARTS + 1];
  int lo[3], hi[3];          // this rank's brick, cells
  int margin;
  int n_boxes;
  uint32_t box_nodes;        // total nodes over all halo boxes
  int int_lo[3], int_hi[3];  // node box that no halo box intersects (fast path of k_grid)
};
struct DevBox {
  int lo[3], dim[3];
  int peer;
  uint32_t off;  // first node of this box in the concatenated (all boxes) node numbering
  float4 *send;
  const float4 *recv;
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

// key of a particle at position x with velocity v: Morton(block of its base cell) << 6 | cell in block;
// INVALID if it must be deleted: non-finite x/v or near the domain wall when clean_boundary
// (src/mpm.h:269-276, src/mpm.cpp:592-598), or a stencil that would leave the grid (reference: UB).
__device__ __forceinline__ uint32_t particle_key(const Params &P, const float x[3], const float v[3], uint32_t &bkey) {
  bool alive = true;
  float X[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    alive = alive && isfinite(x[k]) && isfinite(v[k]);
    X[k] = x[k] * P.idx;
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
  bkey = INVALID;
  if (!alive) return INVALID;
  bkey = morton3(b[0] >> 2, b[1] >> 2, b[2] >> 2);
  return (bkey << 6) | ((b[0] & 3) << 4) | ((b[1] & 3) << 2) | (b[2] & 3);
}

// mark the block active: a plain byte store (all writers store the same value: no atomics, no serialisation),
// one per run of equal blocks among consecutive lanes; k_pack_flags turns the bytes into the bitmap.
// Must be called by all lanes of the wave.
__device__ __forceinline__ void flag_block(uint8_t *__restrict__ blk_flag, uint32_t bkey) {
  const uint32_t prev = __shfl_up(bkey, 1);
  if (bkey != INVALID && ((threadIdx.x & 63) == 0 || prev != bkey)) blk_flag[bkey] = 1;
}

// ------------------------------------------------------------------------------------------------ sort
// standalone key builder (first substep, after uploads, phase-level API); afterwards k_g2p produces the keys
__global__ __launch_bounds__(256) void k_build_keys(Params P, RecG *__restrict__ rg, const RecP *__restrict__ rp,
                                                    Counters *cnt, uint32_t *__restrict__ key,
                                                    uint8_t *__restrict__ blk_flag) {
  const uint32_t n = P.n_slots;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count: all lanes take part in the shuffle
    const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t kk = INVALID, bkey = INVALID;
    if (i < n) {
      const float4 g0 = reinterpret_cast<const float4 *>(rg + i)[0];
      const int32_t pid = rg[i].pid;
      if (pid >= 0) {
        const float4 p0 = reinterpret_cast<const float4 *>(rp + i)[0];
        const float4 p1 = reinterpret_cast<const float4 *>(rp + i)[1];
        const float x[3] = {g0.x, g0.y, g0.z}, v[3] = {p0.w, p1.x, p1.y};
        kk = particle_key(P, x, v, bkey);
        if (kk == INVALID) {  // delete for good (clear_boundary_particles)
          rg[i].pid = -1;
          atomicAdd(&cnt->n_dead, 1u);
        }
      }
      key[i] = kk;
    }
    flag_block(blk_flag, bkey);
  }
}

// ---- single-pass chained scans.  Both tables below are prefix sums over data produced by the previous kernel.
// Instead of the classic three launches (partials, scan of partials, apply) a workgroup publishes the sum of its
// chunk as ONE 64-bit word {epoch, value} (agent-scope atomic: the 8 XCDs' L2s are not coherent for plain
// accesses; the word is self-contained, so relaxed ordering suffices), sums the words of the chunks before it
// (spinning until their epoch matches) and finishes its chunk.  Chunks are handed out by a ticket counter, so a
// chunk's predecessors have always started and never wait on it: no deadlock, no co-residency assumption.  The
// epoch changes with every sort and each kernel zeroes the OTHER kernel's ticket: nothing is cleared by memsets.
__device__ __forceinline__ uint32_t wg_exclusive_scan_256(uint32_t v, uint32_t *lds /*>=4*/, uint32_t &total) {
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

__device__ __forceinline__ void publish(unsigned long long *slot, uint32_t epoch, uint32_t value) {
  __hip_atomic_store(slot, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sum of the published values of chunks [0, chunk): every thread of the 256-thread workgroup gets the result
__device__ __forceinline__ uint32_t sum_predecessors(const unsigned long long *slots, uint32_t chunk, uint32_t epoch,
                                                     uint32_t *lds) {
  uint32_t pre = 0;
  for (uint32_t j = threadIdx.x; j < chunk; j += 256) {
    unsigned long long w;
    while ((uint32_t)((w = __hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch)
      __builtin_amdgcn_s_sleep(1);
    pre += (uint32_t)w;
  }
  uint32_t total;
  wg_exclusive_scan_256(pre, lds, total);
  return total;
}
__device__ __forceinline__ uint32_t take_ticket(uint32_t *ticket, uint32_t *s_chunk) {
  __syncthreads();  // the previous chunk's readers of *s_chunk are done
  if (threadIdx.x == 0) *s_chunk = atomicAdd(ticket, 1u);
  __syncthreads();
  return *s_chunk;
}

// Active-block table, one launch: byte flags -> bitmap `bits` (bit b of word w = block with Morton key 32w+b;
// the flags are cleared behind), per-word prefix `wprefix` (active blocks with key < 32w) = dense slot of every
// active block, the list act_blk[slot] = key, and cnt->n_active.  Chunk = 256 bitmap words, one per thread.
__global__ __launch_bounds__(256) void k_block_table(Params P, uint8_t *__restrict__ blk_flag,
                                                     uint32_t *__restrict__ bits, uint32_t *__restrict__ wprefix,
                                                     uint32_t *__restrict__ act_blk, Counters *cnt,
                                                     unsigned long long *__restrict__ slots, uint32_t *ticket,
                                                     uint32_t epoch) {
  __shared__ uint32_t lds[8];
  __shared__ uint32_t s_chunk;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticket[1] = 0;  // k_cell_table's counter (it is not running now)
  const uint32_t nchunks = (P.nbw + 255) / 256;
  while (true) {
    const uint32_t chunk = take_ticket(ticket, &s_chunk);
    if (chunk >= nchunks) return;
    const uint32_t w = chunk * 256 + threadIdx.x;
    uint32_t m = 0;
    if (w < P.nbw) {
      uint4 *src = reinterpret_cast<uint4 *>(blk_flag + (size_t)w * 32);
      const uint4 lo = src[0], hi = src[1];
      const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t v = q[i];  // four 0/1 bytes -> four bits
        m |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * i);
      }
      if (m) { src[0] = make_uint4(0, 0, 0, 0); src[1] = make_uint4(0, 0, 0, 0); }
    }
    uint32_t total;
    const uint32_t excl = wg_exclusive_scan_256(__popc(m), lds, total);
    if (threadIdx.x == 0) publish(slots + chunk, epoch, total);
    const uint32_t chunk_base = sum_predecessors(slots, chunk, epoch, lds);
    uint32_t run = chunk_base + excl;
    if (w < P.nbw) {
      bits[w] = m;
      wprefix[w] = run;
      uint32_t mm = m;
      while (mm) {
        const int b = __ffs(mm) - 1;
        mm &= mm - 1;
        if (run < P.max_blocks) act_blk[run] = (w << 5) | (uint32_t)b;
        run++;
      }
    }
    if (chunk == nchunks - 1 && threadIdx.x == 255) {
      const uint32_t grand = chunk_base + total;
      if (grand > P.max_blocks) cnt->error |= 1u;
      cnt->n_active = grand;
    }
  }
}

// rank of each particle inside its cell.  Runs of equal keys in consecutive lanes are aggregated into one
// returning atomic per run.  Overwrites key[i] with cidx = slot(block)*64 + cell.
__global__ __launch_bounds__(256) void k_rank(Params P, uint32_t *__restrict__ key, uint32_t *__restrict__ rank,
                                              uint32_t *__restrict__ cell_cnt, const uint32_t *__restrict__ bits,
                                              const uint32_t *__restrict__ wprefix) {
  const uint32_t n = P.n_slots;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count: every lane takes part in the shuffles
    const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = (i < n) ? key[i] : INVALID;
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
      key[i] = cidx;
      rank[i] = base + (lane - start);
    }
  }
}

// Cell table, one launch: per-cell counts (k_rank) -> act_start[a] (first sorted position of active block a,
// sentinel at [n_active]) and cell_start[a*64 + c] (sentinel at [n_active*64]): the particles of cell i are
// perm[cell_start[i] .. cell_start[i+1]).  Zeroes the counters behind itself.  Chunk = the 64 blocks
// [64 t, 64 t + 64): wave w takes the 16 blocks [64 t + 16 w, +16), one lane per cell.
constexpr int CT_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_cell_table(Params P, Counters *cnt, uint32_t *__restrict__ cell_cnt,
                                                    uint32_t *__restrict__ act_start,
                                                    uint32_t *__restrict__ cell_start,
                                                    unsigned long long *__restrict__ slots, uint32_t *ticket,
                                                    uint32_t epoch) {
  __shared__ uint32_t lds[8];
  __shared__ uint32_t s_chunk;
  __shared__ uint32_t blk_tot[CT_BLOCKS];
  if (blockIdx.x == 0 && threadIdx.x == 0) ticket[0] = 0;  // k_block_table's counter (it is not running now)
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  while (true) {
    const uint32_t chunk = take_ticket(ticket + 1, &s_chunk);
    const uint32_t a0 = chunk * CT_BLOCKS;
    if (a0 >= na && !(na == 0 && chunk == 0)) return;
    uint32_t excl[16];  // exclusive in-block prefix of this lane's cell, for the wave's 16 blocks
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t a = a0 + wave * 16 + i;
      uint32_t c = 0;
      if (a < na) {
        c = cell_cnt[(size_t)a * BC + lane];
        cell_cnt[(size_t)a * BC + lane] = 0;
      }
      uint32_t v = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if ((int)lane >= off) v += u;
      }
      excl[i] = v - c;
      if (lane == 63) blk_tot[wave * 16 + i] = v;
    }
    __syncthreads();
    // exclusive scan of the 64 block totals (threads 0..63 hold one block each; other threads contribute 0)
    const uint32_t mine = threadIdx.x < CT_BLOCKS ? blk_tot[threadIdx.x] : 0u;
    uint32_t total;
    const uint32_t boff = wg_exclusive_scan_256(mine, lds, total);
    if (threadIdx.x == 0) publish(slots + chunk, epoch, total);
    if (threadIdx.x < CT_BLOCKS) blk_tot[threadIdx.x] = boff;
    const uint32_t chunk_base = sum_predecessors(slots, chunk, epoch, lds);  // its barriers also cover blk_tot
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t a = a0 + wave * 16 + i;
      if (a < na) {
        const uint32_t start = chunk_base + blk_tot[wave * 16 + i];
        cell_start[(size_t)a * BC + lane] = start + excl[i];
        if (lane == 0) act_start[a] = start;
      }
    }
    if (a0 + CT_BLOCKS >= na && threadIdx.x == 0) {  // last chunk: sentinels + live count
      const uint32_t grand = chunk_base + total;
      act_start[na] = grand;
      cell_start[(size_t)na * BC] = grand;
      cnt->n_sorted = grand;
    }
    if (na == 0) return;
  }
}

// sorted position -> particle slot (the reference's sorted `particles` index vector, src/mpm.cpp:800-807)
__global__ __launch_bounds__(256) void k_perm(Params P, const uint32_t *__restrict__ key,
                                              const uint32_t *__restrict__ rank,
                                              const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ perm) {
  const uint32_t n = P.n_slots;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t c = key[i];
    if (c != INVALID) perm[cell_start[c] + rank[i]] = i;
  }
}

// physical reorder + compaction (sort_allocator, src/mpm.cpp:752-768): records gathered into sorted order
__global__ __launch_bounds__(256) void k_gather_records(const Counters *__restrict__ cnt,
                                                        const uint32_t *__restrict__ perm, const float4 *__restrict__ rg,
                                                        const float4 *__restrict__ rp, const float4 *__restrict__ rb,
                                                        float4 *__restrict__ rg2, float4 *__restrict__ rp2,
                                                        float4 *__restrict__ rb2) {
  const uint32_t n = cnt->n_sorted;
  // one float4 per thread: 4 threads per 64-byte record
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 4u; t += gridDim.x * blockDim.x) {
    const uint32_t j = t >> 2, q = t & 3;
    const uint32_t i = perm[j];
    rg2[(size_t)j * 4 + q] = rg[(size_t)i * 4 + q];
    rp2[(size_t)j * 4 + q] = rp[(size_t)i * 4 + q];
    if (q < 3) rb2[(size_t)j * 3 + q] = rb[(size_t)i * 3 + q];
  }
}
__global__ __launch_bounds__(256) void k_identity_perm(const Counters *__restrict__ cnt, uint32_t *__restrict__ perm) {
  const uint32_t n = cnt->n_sorted;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) perm[i] = i;
}

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

// ------------------------------------------------------------------------------------------------ P2G
// rasterize_optimized / block_op_normal (src/transfer.cpp:467-569).
// Mapping: ONE LANE PER CELL of an active 4^3-cell block.  The sorted index lists the particles of each cell
// contiguously, so lane c walks its cell's particles and accumulates their node contributions in registers (the
// reference walks cells sequentially inside a block and accumulates into its scratch tile the same way,
// :474-483).  Write conflicts between particles of one cell therefore never reach memory; a wave merges its
// per-cell sums into its own 6^3-node LDS tile by ordered, non-atomic float4 read-modify-writes and the tile is
// written out whole; conflicts between blocks are resolved by k_grid.  p2g_cell<N0,N1> handles stencil nodes
// N0..N1-1 of the particles [p0,p1) of the lane's cell, so a block can be one wave (default) or several waves
// splitting the nodes and/or the particles (k_p2g<NS,PS>).
template <int N0, int N1>
__device__ __forceinline__ void p2g_cell(const Params &P, const float4 *__restrict__ rp,
                                         const uint32_t *__restrict__ perm,
                                         const GroupParams *__restrict__ groups, uint32_t p0, uint32_t p1, float ox,
                                         float oy, float oz, int nbase, float4 *tile) {
  constexpr int NN = N1 - N0;
  float acc[NN][4];
#pragma unroll
  for (int n = 0; n < NN; n++) { acc[n][0] = 0.0f; acc[n][1] = 0.0f; acc[n][2] = 0.0f; acc[n][3] = 0.0f; }
  // software pipeline: the records of the next TWO particles and the index of the third are in flight while
  // one particle is computed (one particle's arithmetic is shorter than the loaded HBM latency)
  float4 n0, n1, n2, n3, m0, m1, m2, m3;
  uint32_t inext = 0;
  if (p0 < p1) {
    const size_t i = perm[p0];
    n0 = rp[i * 4 + 0]; n1 = rp[i * 4 + 1]; n2 = rp[i * 4 + 2]; n3 = rp[i * 4 + 3];
    if (p0 + 1 < p1) {
      const size_t j = perm[p0 + 1];
      m0 = rp[j * 4 + 0]; m1 = rp[j * 4 + 1]; m2 = rp[j * 4 + 2]; m3 = rp[j * 4 + 3];
      if (p0 + 2 < p1) inext = perm[p0 + 2];
    }
  }
  for (uint32_t p = p0; p < p1; p++) {
    const float4 q0 = n0, q1 = n1, q2 = n2, q3 = n3;
    n0 = m0; n1 = m1; n2 = m2; n3 = m3;
    if (p + 2 < p1) {
      const size_t i = inext;
      m0 = rp[i * 4 + 0]; m1 = rp[i * 4 + 1]; m2 = rp[i * 4 + 2]; m3 = rp[i * 4 + 3];
      if (p + 3 < p1) inext = perm[p + 3];
    }
    const float mass = q3.w;  // the particle mass travels in the record: no dependent table lookup
    float v0 = q0.w, v1 = q1.x, v2 = q1.y;
    if (P.particle_gravity) {  // src/transfer.cpp:485-487
      v0 = fmaf(P.g[0], P.dt, v0); v1 = fmaf(P.g[1], P.dt, v1); v2 = fmaf(P.g[2], P.dt, v2);
    }
    // position relative to the base cell, in grid units: in [0.5, 1.5)^3  (:490,518)
    const float r0 = q0.x * P.idx - ox, r1 = q0.y * P.idx - oy, r2 = q0.z * P.idx - oz;
    float w0[3], w1[3], w2[3];
    bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
    const float A00 = q1.z, A01 = q1.w, A02 = q2.x, A10 = q2.y, A11 = q2.z, A12 = q2.w, A20 = q3.x, A21 = q3.y,
                A22 = q3.z;
    const float mv0 = mass * v0, mv1 = mass * v1, mv2 = mass * v2;
#pragma unroll
    for (int n = N0; n < N1; n++) {
      const int i3 = n / 9, j = (n / 3) % 3, k = n % 3;
      const float d0 = r0 - (float)i3, d1 = r1 - (float)j, d2 = r2 - (float)k;
      const float w = (w0[i3] * w1[j]) * w2[k];
      // :535-541  contrib = (affine * dpos + mass*v, mass); g += weight * contrib
      const float c0 = fmaf(A02, d2, fmaf(A01, d1, fmaf(A00, d0, mv0)));
      const float c1 = fmaf(A12, d2, fmaf(A11, d1, fmaf(A10, d0, mv1)));
      const float c2 = fmaf(A22, d2, fmaf(A21, d1, fmaf(A20, d0, mv2)));
      acc[n - N0][0] = fmaf(w, c0, acc[n - N0][0]);
      acc[n - N0][1] = fmaf(w, c1, acc[n - N0][1]);
      acc[n - N0][2] = fmaf(w, c2, acc[n - N0][2]);
      acc[n - N0][3] = fmaf(w, mass, acc[n - N0][3]);
    }
  }
  // Merge the per-cell sums into this wave's tile.  The tile belongs to this wavefront alone, and within one
  // stencil-offset step all 64 lanes address distinct nodes (same offset, different cells), so a plain float4
  // read-modify-write is race-free as long as the steps stay in program order: LDS operations of one wave
  // execute in order, the wave_barrier keeps the compiler from interleaving them.  (DS float atomics cost
  // ~2 LDS cycles per LANE on gfx950 even without conflicts: measured 145 cycles per ds_add_f32.)
#pragma unroll
  for (int n = N0; n < N1; n++) {
    const int node = nbase + ((n / 9) * TS + (n / 3) % 3) * TS + n % 3;
    if ((P.ablate & 8) && acc[n - N0][3] != 1.2345e-30f) continue;
    if (p1 > p0) {
      float4 t = tile[node];
      t.x += acc[n - N0][0]; t.y += acc[n - N0][1]; t.z += acc[n - N0][2]; t.w += acc[n - N0][3];
      tile[node] = t;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

// NS = waves splitting the 27 stencil nodes (1 or 2), PS = waves splitting every cell's particles (1, 2 or 4):
// NS*PS wavefronts per block, each with its own LDS tile.  Node splitting halves the accumulator registers
// (occupancy) but both halves load the same records; particle splitting keeps every record load unique.
template <int NS, int PS, int MINW>
__global__ __launch_bounds__(64 * NS * PS, MINW) void k_p2g(Params P, const float4 *__restrict__ rp,
                                                            const Counters *__restrict__ cnt,
                                                            const uint32_t *__restrict__ act_blk,
                                                            const uint32_t *__restrict__ cell_start,
                                                            const uint32_t *__restrict__ perm,
                                                            const GroupParams *__restrict__ groups,
                                                            float4 *__restrict__ tiles) {
  constexpr int NW = NS * PS, NT = 64 * NW;
  __shared__ float4 tile[NW][TN];  // per wave: (m*vx, m*vy, m*vz, m) per node of the block's 6^3 tile
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npart = wave % NS, ppart = wave / NS;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const int nbase = (cx * TS + cy) * TS + cz;
  for (uint32_t a = blockIdx.x; a < na; a += gridDim.x) {
    for (int t = threadIdx.x; t < NW * TN; t += NT) (&tile[0][0])[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    const float ox = (float)(bx * BS + cx), oy = (float)(by * BS + cy), oz = (float)(bz * BS + cz);
    const uint32_t c0 = cell_start[a * BC + lane], c1 = cell_start[a * BC + lane + 1];
    const uint32_t n = c1 - c0;
    const uint32_t p0 = c0 + (n * ppart + PS - 1) / PS, p1 = c0 + (n * (ppart + 1) + PS - 1) / PS;
    if constexpr (NS == 1) {
      p2g_cell<0, 27>(P, rp, perm, groups, p0, p1, ox, oy, oz, nbase, tile[wave]);
    } else {
      if (npart == 0) p2g_cell<0, 14>(P, rp, perm, groups, p0, p1, ox, oy, oz, nbase, tile[wave]);
      else p2g_cell<14, 27>(P, rp, perm, groups, p0, p1, ox, oy, oz, nbase, tile[wave]);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TN; t += NT) {
      float4 u = tile[0][t];
#pragma unroll
      for (int w = 1; w < NW; w++) {
        const float4 q = tile[w][t];
        u.x += q.x; u.y += q.y; u.z += q.z; u.w += q.w;
      }
      tiles[(size_t)a * TN + t] = u;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ grid
// One wavefront per (active block a, o).  The tile of a overlaps the 8 grid blocks c = block(a) + o, o in {0,1}^3
// ("candidates").  A grid block c is processed by its "owner": the candidate with the smallest o among the
// active blocks c - o'.  The owner sums the overlapping tiles (<= 8), then
//   mode 0: normalize_grid_and_apply_external_force + apply_grid_boundary_conditions (src/mpm.cpp:277-372)
//           -> gridv[slot = 8a+o], fat_slot[morton(c)] = slot
//   mode 1: raw (m v, m) sums written to a dense node-major array (parity / download only)
//   mode 2: dense (v, m) array -> gridv (upload_grid)        mode 3: gridv -> dense (download_grid)
// All candidate/owner/source lookups of a block involve only its 27 neighbours b + {-1,0,1}^3: lanes 0..26
// look one neighbour up each (one round trip), the rest is ballots and shuffles.
__device__ __forceinline__ constexpr int nb27(int dx, int dy, int dz) { return ((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1); }

template <int MODE, bool PER_CAND>
__global__ __launch_bounds__(256) void k_grid(Params P, const Counters *__restrict__ cnt,
                                              const uint32_t *__restrict__ act_blk,
                                              const uint32_t *__restrict__ bits,
                                              const uint32_t *__restrict__ wprefix,
                                              const float4 *__restrict__ tiles, float4 *__restrict__ gridv,
                                              uint32_t *__restrict__ fat_slot, float4 *__restrict__ dense, Tiling T,
                                              const DevBox *__restrict__ boxes, LevelSetDev LS) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int l = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
  // The kernel is a chain of dependent lookups (block list -> bitmap/prefix -> tiles -> halo -> store).
  // PER_CAND = false: one wavefront per active block walks its 8 candidates (lookups shared; best when there are
  // more blocks than resident waves).  PER_CAND = true: one wavefront per (block, candidate) — 8x the lookups but
  // an 8x shorter chain for the boundary blocks that own many candidates (best for small per-GPU problems, i.e.
  // the tiled multi-GPU runs: 32 -> 19 us at 1 M particles; 33 -> 88 us at 8 M, hence the switch in do_grid).
  const uint32_t nwork = PER_CAND ? na * 8u : na;
  for (uint32_t work = wave; work < nwork; work += nwaves) {
    const uint32_t a = PER_CAND ? work >> 3 : work;
    const int o_mine = PER_CAND ? (int)(work & 7u) : -1;
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    // neighbour table: lane n < 27 holds (active?, slot) of block b + (n/9-1, n/3%3-1, n%3-1)
    uint32_t nslot = INVALID;
    if (l < 27) {
      const int sx = bx + l / 9 - 1, sy = by + (l / 3) % 3 - 1, sz = bz + l % 3 - 1;
      if (sx >= 0 && sy >= 0 && sz >= 0) {
        const uint32_t bk = morton3(sx, sy, sz);
        if (block_active(bits, bk)) nslot = block_slot(bits, wprefix, bk);
      }
    }
    const uint32_t amask = (uint32_t)__ballot(nslot != INVALID);
#pragma unroll
    for (int o = 0; o < 8; o++) {
      if (PER_CAND && o != o_mine) continue;  // wave-uniform (the loop stays unrolled: compile-time masks)
      const int ox = o >> 2, oy = (o >> 1) & 1, oz = o & 1;
      // sources of c = b + o are c - q = b + (o - q), q in {0,1}^3; owner <=> none of them active for q < o
      uint32_t lower = 0;
#pragma unroll
      for (int q = 0; q < o; q++) lower |= 1u << nb27(ox - (q >> 2), oy - ((q >> 1) & 1), oz - (q & 1));
      if (amask & lower) continue;  // wave-uniform
      const int cx = bx + ox, cy = by + oy, cz = bz + oz;
      const uint32_t slot = a * 8u + (uint32_t)o;
      const int gi = cx * BS + lx, gj = cy * BS + ly, gk = cz * BS + lz;
      const bool in_grid = gi <= P.res[0] && gj <= P.res[1] && gk <= P.res[2];
      const size_t dense_idx = ((size_t)gi * (P.res[1] + 1) + gj) * (P.res[2] + 1) + gk;
      if (MODE == 2) {
        gridv[(size_t)slot * BC + l] = in_grid ? dense[dense_idx] : make_float4(0, 0, 0, 0);
        if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
        continue;
      }
      if (MODE == 3) {
        if (in_grid) dense[dense_idx] = gridv[(size_t)slot * BC + l];
        continue;
      }
      float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int qx = q >> 2, qy = (q >> 1) & 1, qz = q & 1;
        const int nidx = nb27(ox - qx, oy - qy, oz - qz);
        const uint32_t sslot = __shfl(nslot, nidx);
        const int tx = lx + 4 * qx, ty = ly + 4 * qy, tz = lz + 4 * qz;
        if (((amask >> nidx) & 1u) && tx < TS && ty < TS && tz < TS) {
          const float4 t = tiles[(size_t)sslot * TN + (tx * TS + ty) * TS + tz];
          acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
      }
      if (T.n_boxes > 0) {  // tiled: add the other ranks' partial sums, contributors in rank order
        const bool interior = gi >= T.int_lo[0] && gi < T.int_hi[0] && gj >= T.int_lo[1] && gj < T.int_hi[1] &&
                              gk >= T.int_lo[2] && gk < T.int_hi[2];
        if (__any(!interior)) {
          // contributors in rank order; the peers' values are fetched eight boxes at a time (independent loads,
          // one round trip) and then added in order
          float4 tot = make_float4(0, 0, 0, 0);
          bool own = false;
          for (int b0 = 0; b0 < T.n_boxes; b0 += 8) {
            float4 r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              r[u] = make_float4(0, 0, 0, 0);
              if (b0 + u < T.n_boxes) {
                const DevBox &B = boxes[b0 + u];
                const int x = gi - B.lo[0], y = gj - B.lo[1], z = gk - B.lo[2];
                if ((unsigned)x < (unsigned)B.dim[0] && (unsigned)y < (unsigned)B.dim[1] && (unsigned)z < (unsigned)B.dim[2])
                  r[u] = B.recv[((size_t)x * B.dim[1] + y) * B.dim[2] + z];
              }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
              if (b0 + u < T.n_boxes) {
                if (!own && boxes[b0 + u].peer > T.rank) {
                  tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w;
                  own = true;
                }
                tot.x += r[u].x; tot.y += r[u].y; tot.z += r[u].z; tot.w += r[u].w;
              }
            }
          }
          if (!own) { tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w; }
          acc = tot;
        }
      }
      if (MODE == 1) {
        if (in_grid) dense[dense_idx] = acc;
        continue;
      }
      if (MODE == 4) {  // grid kinetic energy sum 1/2 m |v|^2 with v = (m v)/m (calculate_energy, src/mpm.cpp:1078-1096)
        double e = (acc.w != 0.0f) ? 0.5 * ((double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z) / acc.w : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if (l == 0) atomicAdd(reinterpret_cast<double *>(dense), e);
        continue;
      }
      float v[3] = {acc.x, acc.y, acc.z};
      const float m = acc.w;
      if (m > 0.0f) {  // src/mpm.cpp:282-292; increment is gravity*dt only when !particle_gravity (:526-530)
        const float im = 1.0f / m;
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = fmaf(v[k], im, P.particle_gravity ? 0.0f : P.g[k] * P.dt);
      }
      if (m != 0.0f && LS.n > 0) {  // src/mpm.cpp:313-368
        const float xw[3] = {gi * P.dx, gj * P.dx, gk * P.dx};
        float phi, nrm[3] = {0, 0, 0};
        levelset_eval(LS, xw, P.idx, phi, nrm);
        if (!(phi < -3.0f || 0.0f < phi)) {
          const float vb[3] = {0, 0, 0};
          friction_project(v, vb, nrm, LS.friction);
        }
      }
      gridv[(size_t)slot * BC + l] = make_float4(v[0], v[1], v[2], m);
      if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
    }
  }
}

// ------------------------------------------------------------------------------------------------ tiling
// This rank's partial (m v, m) sums on every halo box -> the box's send buffer.  One thread per box node: the
// node's grid block c and the <= 8 active source blocks c - q whose 6^3 tiles overlap it (same sum as k_grid).
__global__ __launch_bounds__(256) void k_halo_pack(Params P, Tiling T, const DevBox *__restrict__ boxes,
                                                   const uint32_t *__restrict__ bits,
                                                   const uint32_t *__restrict__ wprefix,
                                                   const float4 *__restrict__ tiles) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < T.box_nodes; t += gridDim.x * blockDim.x) {
    int b = 0;
    while (b + 1 < T.n_boxes && t >= boxes[b + 1].off) b++;
    const DevBox &B = boxes[b];
    const uint32_t r = t - B.off;
    const int z = r % B.dim[2], y = (r / B.dim[2]) % B.dim[1], x = r / (B.dim[2] * B.dim[1]);
    const int gi = B.lo[0] + x, gj = B.lo[1] + y, gk = B.lo[2] + z;
    const int cx = gi >> 2, cy = gj >> 2, cz = gk >> 2, lx = gi & 3, ly = gj & 3, lz = gk & 3;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int qx = q >> 2, qy = (q >> 1) & 1, qz = q & 1;
      const int sx = cx - qx, sy = cy - qy, sz = cz - qz;
      const int tx = lx + 4 * qx, ty = ly + 4 * qy, tz = lz + 4 * qz;
      if (sx < 0 || sy < 0 || sz < 0 || tx >= TS || ty >= TS || tz >= TS) continue;
      const uint32_t bk = morton3(sx, sy, sz);
      if (bk >= P.nbw * 32u || !block_active(bits, bk)) continue;
      const uint32_t slot = block_slot(bits, wprefix, bk);
      if (slot >= P.max_blocks) continue;
      const float4 v = tiles[(size_t)slot * TN + (tx * TS + ty) * TS + tz];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    B.send[r] = acc;
  }
}

// bounding box (cells) of the active blocks of the last sort: out[0..2] = min, out[3..5] = max (exclusive)
__global__ __launch_bounds__(256) void k_active_bounds(Params P, const Counters *__restrict__ cnt,
                                                       const uint32_t *__restrict__ act_blk, int *__restrict__ out) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < na; a += gridDim.x * blockDim.x) {
    int b[3];
    demorton3(act_blk[a], b[0], b[1], b[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], b[k] * BS); hi[k] = max(hi[k], b[k] * BS + BS); }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = min(lo[k], __shfl_xor(lo[k], off));
      hi[k] = max(hi[k], __shfl_xor(hi[k], off));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[k], lo[k]); atomicMax(&out[3 + k], hi[k]); }
  }
}

__device__ __forceinline__ int part_index(const int *cuts, int n, int c) {
  int p = 0;
  while (p + 1 < n && c >= cuts[p + 1]) p++;
  return p;
}
// destination rank of a live particle at x (brick containing its base cell); -1 if it is not representable
__device__ __forceinline__ int dest_rank(const Params &P, const Tiling &T, float4 g0, bool &beyond_margin) {
  int b[3];
  const float X[3] = {g0.x * P.idx, g0.y * P.idx, g0.z * P.idx};
  beyond_margin = false;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (!isfinite(X[k]) || X[k] < 0.5f) return -1;
    b[k] = (int)(X[k] - 0.5f);
    if (b[k] < T.lo[k] - T.margin || b[k] >= T.hi[k] + T.margin) beyond_margin = true;
  }
  return (part_index(T.cuts[0], T.dims[0], b[0]) * T.dims[1] + part_index(T.cuts[1], T.dims[1], b[1])) * T.dims[2] +
         part_index(T.cuts[2], T.dims[2], b[2]);
}

__global__ void k_scan_init(uint32_t *__restrict__ counts, int world) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < world) counts[t] = 0;
  else if (t < world + 3) counts[t] = (uint32_t)(1 << 30);
  else if (t < world + 6) counts[t] = (uint32_t)-1;
}
// counts[d] = live particles whose base cell belongs to rank d != this rank; bounds[0..2] / [3..5] = min / max+1 of
// the base cells of all live particles (one pass, one wave-reduced atomic set per wave)
__global__ __launch_bounds__(256) void k_leaver_count(Params P, Tiling T, const float4 *__restrict__ rg,
                                                      uint32_t *__restrict__ counts, int *__restrict__ bounds,
                                                      Counters *cnt) {
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    if (__float_as_int(rg[(size_t)i * 4 + 3].z) < 0) continue;
    const float4 g0 = rg[(size_t)i * 4];
    bool beyond;
    const int d = dest_rank(P, T, g0, beyond);
    if (d < 0) continue;
    if (beyond) atomicOr(&cnt->error, 2u);
    if (d != T.rank) atomicAdd(&counts[d], 1u);
    const int b[3] = {(int)(g0.x * P.idx - 0.5f), (int)(g0.y * P.idx - 0.5f), (int)(g0.z * P.idx - 0.5f)};
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], b[k]); hi[k] = max(hi[k], b[k] + 1); }
  }
  // wave reduce -> workgroup reduce -> 6 atomics per WORKGROUP (same-address atomics serialise at ~13 ns each)
  __shared__ int red[4][6];
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = min(lo[k], __shfl_xor(lo[k], off));
      hi[k] = max(hi[k], __shfl_xor(hi[k], off));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    const int l = min(min(red[0][k], red[1][k]), min(red[2][k], red[3][k]));
    const int h = max(max(red[0][3 + k], red[1][3 + k]), max(red[2][3 + k], red[3][3 + k]));
    if (h >= 0) { atomicMin(&bounds[k], l); atomicMax(&bounds[3 + k], h); }
  }
}

// cursor[d] starts at the first record index of destination d; leavers are removed from this rank
__global__ __launch_bounds__(256) void k_leaver_pack(Params P, Tiling T, float4 *__restrict__ rg,
                                                     const float4 *__restrict__ rp, const float4 *__restrict__ rb,
                                                     uint32_t *__restrict__ key, uint32_t *__restrict__ cursor,
                                                     float4 *__restrict__ out, Counters *cnt) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const float4 g3 = rg[(size_t)i * 4 + 3];
    if (__float_as_int(g3.z) < 0) continue;
    bool beyond;
    const int d = dest_rank(P, T, rg[(size_t)i * 4], beyond);
    if (d < 0 || d == T.rank) continue;
    const size_t o = (size_t)atomicAdd(&cursor[d], 1u) * 11;
#pragma unroll
    for (int q = 0; q < 4; q++) out[o + q] = rg[(size_t)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; q++) out[o + 4 + q] = rp[(size_t)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 3; q++) out[o + 8 + q] = rb[(size_t)i * 3 + q];
    rg[(size_t)i * 4 + 3] = make_float4(g3.x, g3.y, __int_as_float(-1), 0.0f);
    key[i] = INVALID;
    atomicAdd(&cnt->n_dead, 1u);
  }
}

// arrivals appended at slots base .. base+n; their keys and block flags join the ones k_g2p produced
__global__ __launch_bounds__(256) void k_import(Params P, uint32_t n, uint32_t base, const float4 *__restrict__ in,
                                                float4 *__restrict__ rg, float4 *__restrict__ rp,
                                                float4 *__restrict__ rb, uint32_t *__restrict__ key,
                                                uint8_t *__restrict__ blk_flag, Counters *cnt) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count (flag_block shuffles)
    const uint32_t j = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bkey = INVALID;
    if (j < n) {
      const size_t i = (size_t)base + j, o = (size_t)j * 11;
      const float4 g0 = in[o], g3 = in[o + 3], p0 = in[o + 4], p1 = in[o + 5];
      const float x[3] = {g0.x, g0.y, g0.z}, v[3] = {p0.w, p1.x, p1.y};
      const uint32_t kk = particle_key(P, x, v, bkey);
      int32_t pid = __float_as_int(g3.z);
      if (kk == INVALID) {
        pid = -1;
        atomicAdd(&cnt->n_dead, 1u);
      }
      key[i] = kk;
      rg[i * 4 + 0] = g0; rg[i * 4 + 1] = in[o + 1]; rg[i * 4 + 2] = in[o + 2];
      rg[i * 4 + 3] = make_float4(g3.x, g3.y, __int_as_float(pid), 0.0f);
      rp[i * 4 + 0] = p0; rp[i * 4 + 1] = p1; rp[i * 4 + 2] = in[o + 6]; rp[i * 4 + 3] = in[o + 7];
      rb[i * 3 + 0] = in[o + 8]; rb[i * 3 + 1] = in[o + 9]; rb[i * 3 + 2] = in[o + 10];
    }
    flag_block(blk_flag, bkey);
  }
}

// ------------------------------------------------------------------------------------------------ G2P
// resample_optimized / block_op_normal (src/transfer.cpp:837-954), one workgroup per active block, one
// particle per lane through the sorted index.  Also produces, for the NEXT substep: the affine matrix A of
// P2G (stress of the updated F from the same eigen-solve as the plasticity) and the sort key of the new position.
template <int NT, int MINW, bool ROLL>
__global__ __launch_bounds__(NT, MINW) void k_g2p(Params P, float4 *__restrict__ rg, float4 *__restrict__ rp,
                                                  float4 *__restrict__ rb, const Counters *__restrict__ cnt,
                                                  const uint32_t *__restrict__ act_blk,
                                                  const uint32_t *__restrict__ act_start,
                                                  const uint32_t *__restrict__ perm,
                                                  const GroupParams *__restrict__ groups,
                                                  const float4 *__restrict__ gridv,
                                                  const uint32_t *__restrict__ fat_slot, Counters *cnt_w,
                                                  uint32_t *__restrict__ key, uint8_t *__restrict__ blk_flag,
                                                  LevelSetDev LS) {
  __shared__ float4 tile[TN];
  // Store staging, one slab per wavefront.  A lane holds its particle's whole record, so a direct store would
  // issue 16-byte pieces at a 64-byte stride: 64 partial-line write requests per instruction (measured: the
  // stores alone cost 0.28 of 0.61 ms).  Records are written row-wise to LDS (80-byte stride: conflict-free
  // b128) and read back transposed, so 4 consecutive lanes store the 4 float4 of one record: full 64-byte
  // segments, 4x fewer write requests.
  __shared__ float4 xpose[NT / 64][64 * 5];
  __shared__ uint32_t xslot[NT / 64][64];
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  float4 *xp = xpose[wave];
  uint32_t *xs = xslot[wave];
  const float scale = -4.0f * P.idx * P.dt;  // :938
  // The workgroup walks "chunks": NT consecutive entries of the sorted index inside one active block.  The
  // record gather of chunk k+1 and the index load of chunk k+2 are issued before the arithmetic of chunk k
  // (also across block boundaries), so every wave keeps 4 KiB of loads in flight while it computes.
  struct Chunk { uint32_t a, p, p1; };
  auto first = [&](uint32_t a) {
    Chunk c;
    c.a = a; c.p = 0; c.p1 = 0;
    while (c.a < na) {
      c.p = act_start[c.a]; c.p1 = act_start[c.a + 1];
      if (c.p < c.p1) break;
      c.a += gridDim.x;  // empty block (all its particles migrated away)
    }
    return c;
  };
  auto next = [&](Chunk c) {
    if (c.a >= na) return c;
    c.p += NT;
    if (c.p >= c.p1) c = first(c.a + gridDim.x);
    return c;
  };
  auto lane_slot = [&](const Chunk &c) -> uint32_t {
    return (c.a < na && c.p + tid < c.p1) ? perm[c.p + tid] : INVALID;
  };
  Chunk cur = first(blockIdx.x);
  Chunk nx = next(cur);
  uint32_t i_cur = lane_slot(cur);
  float4 g0, g1, g2, g3;
  if (i_cur != INVALID) {
    const size_t i = i_cur;
    g0 = rg[i * 4 + 0]; g1 = rg[i * 4 + 1]; g2 = rg[i * 4 + 2]; g3 = rg[i * 4 + 3];
  }
  uint32_t i_nx = lane_slot(nx);
  uint32_t tile_a = INVALID;
  float ox = 0, oy = 0, oz = 0;
  while (cur.a < na) {
    if (cur.a != tile_a) {
      __syncthreads();  // everyone is done with the previous tile
      int bx, by, bz;
      demorton3(act_blk[cur.a], bx, by, bz);
      for (int t = tid; t < TN; t += NT) {
        const int tx = t / (TS * TS), ty = (t / TS) % TS, tz = t % TS;
        const int qx = tx >> 2, qy = ty >> 2, qz = tz >> 2;
        const uint32_t fs = fat_slot[morton3(bx + qx, by + qy, bz + qz)];
        tile[t] = gridv[(size_t)fs * BC + (((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3))];
      }
      __syncthreads();
      ox = (float)(bx * BS); oy = (float)(by * BS); oz = (float)(bz * BS);
      tile_a = cur.a;
    }
    // prefetch: records of the next chunk, index of the one after
    const Chunk nn = next(nx);
    float4 n0, n1, n2, n3;
    if (i_nx != INVALID) {
      const size_t i = i_nx;
      n0 = rg[i * 4 + 0]; n1 = rg[i * 4 + 1]; n2 = rg[i * 4 + 2]; n3 = rg[i * 4 + 3];
    }
    const uint32_t i_nn = lane_slot(nn);
    uint32_t bkey = INVALID, out_slot = INVALID;
    float4 G0, G1, G2, G3, Q0, Q1, Q2, Q3, B0, B1, B2;
    G0 = G1 = G2 = G3 = Q0 = Q1 = Q2 = Q3 = B0 = B1 = B2 = make_float4(0, 0, 0, 0);
    if (i_cur != INVALID) {
      const size_t i = i_cur;
      const uint32_t gid = __float_as_uint(g3.y);
      const GroupParams &g = groups[gid];  // read at use (L1-resident table): keeps 20 VGPRs free
      const float x0 = g0.x, x1 = g0.y, x2 = g0.z;
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
      auto plane = [&](int i3) __attribute__((always_inline)) {
        const float d0 = r0 - (float)i3;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const float d1 = r1 - (float)j;
          const float wij = w0[i3] * w1[j];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const float d2 = r2 - (float)k;
            const float w = wij * w2[k];
            const float4 gv = tile[nbase + (i3 * TS + j) * TS + k];
            // :898-903  v_ = fma(grid_vel, w, v_);  b_[r] = fma(w*grid_vel, dpos[r], b_[r])
            v0 = fmaf(gv.x, w, v0); v1 = fmaf(gv.y, w, v1); v2 = fmaf(gv.z, w, v2);
            const float a0 = w * gv.x, a1 = w * gv.y, a2 = w * gv.z;
            b(0, 0) = fmaf(a0, d0, b(0, 0)); b(0, 1) = fmaf(a0, d1, b(0, 1)); b(0, 2) = fmaf(a0, d2, b(0, 2));
            b(1, 0) = fmaf(a1, d0, b(1, 0)); b(1, 1) = fmaf(a1, d1, b(1, 1)); b(1, 2) = fmaf(a1, d2, b(1, 2));
            b(2, 0) = fmaf(a2, d0, b(2, 0)); b(2, 1) = fmaf(a2, d1, b(2, 1)); b(2, 2) = fmaf(a2, d2, b(2, 2));
          }
        }
      };
      if (!(P.ablate & 4)) {
        if constexpr (ROLL) {  // rolled i-loop: 9 LDS reads in flight instead of 27 (VGPR pressure -> occupancy)
#pragma unroll 1
          for (int i3 = 0; i3 < 3; i3++) plane(i3);
        } else {
          plane(0); plane(1); plane(2);
        }
      }
      mat3 cdg;  // :940-942  cdg = I + (-4 inv_dx dt) b   (undamped b, as in the reference)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) cdg(r, c) = fmaf(scale, b(r, c), (r == c) ? 1.0f : 0.0f);
      // apic_b = damp_affine_momemtum(b) (src/mpm.h:465-469); the reference's optimised path has a bug
      // here (passes the block index, transfer.cpp:925-926) — we implement the intended damping.
      if (P.rpic_damping != 0.0f || P.apic_damping != 0.0f) {
        const float ks = 1.0f - P.rpic_damping, ka = 1.0f - P.apic_damping;
        mat3 bd;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float sym = 0.5f * (b(r, c) + b(c, r));
            bd(r, c) = ks * sym + ka * (b(r, c) - sym);
          }
        b = bd;
      }
      mat3 F;
      F.m[0] = g1.x; F.m[1] = g1.y; F.m[2] = g1.z; F.m[3] = g1.w; F.m[4] = g2.x; F.m[5] = g2.y; F.m[6] = g2.z;
      F.m[7] = g2.w; F.m[8] = g3.x;
      float aux = g0.w;
      mat3 stress;
      if (!(P.ablate & 2)) plasticity_and_force(g, cdg, F, aux, stress);  // :950 + next substep's :509
      else stress = cdg;
      float nx0 = fmaf(v0, P.dt, x0), nx1 = fmaf(v1, P.dt, x1), nx2 = fmaf(v2, P.dt, x2);  // :951
      if (LS.particle_collision) {  // particle_collision_resolution, src/mpm.cpp:414-426 (runs after G2P, :566-569)
        const float xw[3] = {nx0, nx1, nx2};
        float phi, gr[3] = {0, 0, 0};
        if (levelset_eval(LS, xw, P.idx, phi, gr) && phi < 0.0f) {
          const float vn = gr[0] * v0 + gr[1] * v1 + gr[2] * v2;
          nx0 -= gr[0] * phi * P.dx; nx1 -= gr[1] * phi * P.dx; nx2 -= gr[2] * phi * P.dx;
          v0 -= vn * gr[0]; v1 -= vn * gr[1]; v2 -= vn * gr[2];
        }
      }
      const float m4 = 4.0f * g.p[0];
      float A[9];
#pragma unroll
      for (int k = 0; k < 9; k++) A[k] = fmaf(stress.m[k], scale, b.m[k] * m4);  // next P2G's :521-522
      // next substep's key; deleted particles (clear_boundary_particles) are marked for good
      const float nxp[3] = {nx0, nx1, nx2}, nv[3] = {v0, v1, v2};
      const uint32_t kk = particle_key(P, nxp, nv, bkey);
      int32_t pid = __float_as_int(g3.z);
      if (kk == INVALID) {
        pid = -1;
        atomicAdd(&cnt_w->n_dead, 1u);
      }
      key[i] = kk;
      G0 = make_float4(nx0, nx1, nx2, aux);
      G1 = make_float4(F.m[0], F.m[1], F.m[2], F.m[3]);
      G2 = make_float4(F.m[4], F.m[5], F.m[6], F.m[7]);
      G3 = make_float4(F.m[8], g3.y, __int_as_float(pid), 0.0f);
      Q0 = make_float4(nx0, nx1, nx2, v0);
      Q1 = make_float4(v1, v2, A[0], A[1]);
      Q2 = make_float4(A[2], A[3], A[4], A[5]);
      Q3 = make_float4(A[6], A[7], A[8], g.p[0]);
      B0 = make_float4(b.m[0], b.m[1], b.m[2], b.m[3]);
      B1 = make_float4(b.m[4], b.m[5], b.m[6], b.m[7]);
      B2 = make_float4(b.m[8], 0.0f, 0.0f, 0.0f);
      out_slot = (P.ablate & 1) ? INVALID : i_cur;
    }
    // transposed stores through this wave's LDS slab (DS operations of one wave execute in program order)
    xs[lane] = out_slot;
    xp[lane * 5 + 0] = G0; xp[lane * 5 + 1] = G1; xp[lane * 5 + 2] = G2; xp[lane * 5 + 3] = G3;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int src = 16 * k + (lane >> 2), q = lane & 3;
      const uint32_t sl = xs[src];
      const float4 val = xp[src * 5 + q];
      if (sl != INVALID) rg[(size_t)sl * 4 + q] = val;
    }
    __builtin_amdgcn_wave_barrier();
    xp[lane * 5 + 0] = Q0; xp[lane * 5 + 1] = Q1; xp[lane * 5 + 2] = Q2; xp[lane * 5 + 3] = Q3;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int src = 16 * k + (lane >> 2), q = lane & 3;
      const uint32_t sl = xs[src];
      const float4 val = xp[src * 5 + q];
      if (sl != INVALID) rp[(size_t)sl * 4 + q] = val;
    }
    if (P.store_b) {
      __builtin_amdgcn_wave_barrier();
      xp[lane * 5 + 0] = B0; xp[lane * 5 + 1] = B1; xp[lane * 5 + 2] = B2;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int e = 64 * k + lane, src = e / 3, q = e - 3 * src;
        const uint32_t sl = xs[src];
        const float4 val = xp[src * 5 + q];
        if (sl != INVALID) rb[(size_t)sl * 3 + q] = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
    flag_block(blk_flag, bkey);
    cur = nx; nx = nn;
    i_cur = i_nx; i_nx = i_nn;
    g0 = n0; g1 = n1; g2 = n2; g3 = n3;
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
__global__ void k_debug_plasticity(GroupParams g, int64_t n, const float *cdg, float *F, float *aux, float *force_out) {
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mat3 f, c;
    for (int k = 0; k < 9; k++) { f.m[k] = F[9 * i + k]; c.m[k] = cdg[9 * i + k]; }
    float a = aux[i];
    if (force_out) {
      mat3 st;
      plasticity_and_force(g, c, f, a, st);
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

// ================================================================================================ host side
using namespace mpm;

static thread_local std::string g_create_error;

enum { PH_SORT = 0, PH_P2G = 1, PH_EXCH = 2, PH_GRID = 3, PH_G2P = 4, PH_COUNT = 5 };

struct mpmhip_ctx {
  mpmhip_config cfg;
  Params P;
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;
  // particles
  int64_t cap = 0;
  int64_t n_slots = 0;  // slots in use (live + deleted)
  int32_t next_pid = 0;
  RecG *rg = nullptr, *rg2 = nullptr;
  RecP *rp = nullptr, *rp2 = nullptr;
  float *rb = nullptr, *rb2 = nullptr;
  uint32_t *key = nullptr, *rank = nullptr, *perm = nullptr;
  // blocks
  uint32_t NB = 0;
  uint8_t *blk_flag = nullptr;
  uint32_t *bits = nullptr, *wprefix = nullptr, *act_blk = nullptr, *act_start = nullptr;
  uint32_t *cell_cnt = nullptr, *cell_start = nullptr, *fat_slot = nullptr, *ticket = nullptr;
  unsigned long long *scan_slots = nullptr;  // [256] k_block_table + [ct_grid] k_cell_table: {epoch, chunk sum}
  uint32_t sort_epoch = 0, bt_slots = 0;
  float4 *tiles = nullptr, *gridv = nullptr, *dense = nullptr;
  Counters *cnt = nullptr;
  std::vector<GroupParams> groups;
  GroupParams *d_groups = nullptr;
  int groups_cap = 256;
  bool sorted = false;        // perm / cell_start describe the current positions
  bool keys_valid = false;    // key[] + block flags describe the current positions (set by k_g2p)
  bool affine_valid = false;  // RecP.A matches (F, aux, apic_b)
  bool b_stale = false;       // discard_apic_b: the side array is behind RecP.A (k_g2p did not write it)
  int p2g_wgs = 16384;        // workgroups of k_p2g (env MPMHIP_P2G_WGS)
  int p2g_split = 11;         // tuning knob (env MPMHIP_P2G_SPLIT): 10*NS + PS, see do_p2g
  int g2p_minw = 13;          // tuning knob (env MPMHIP_G2P_MINW): __launch_bounds__ waves/SIMD of k_g2p
  int reorder_interval = 0;   // physical reorder every this many substeps (0 = never); env MPMHIP_REORDER_INTERVAL
  float t = 0.0f, request_t = 0.0f;  // `real` accumulators, as in the reference (src/mpm.h:99, mpm.cpp:573)
  int64_t substeps = 0;
  // profiling
  int profiling = 0;  // 0 off, 1 every phase, 2 only G2P, 3 only P2G (two events per substep instead of six)
  struct Ev { hipEvent_t e[PH_COUNT + 1]; };
  std::vector<Ev> ev_pool;
  size_t ev_used = 0;
  double phase_ms[PH_COUNT] = {0, 0, 0, 0, 0};
  int64_t prof_substeps = 0;
  int ev_level = 0;  // level the pooled events were recorded with
  Ev *cur_ev = nullptr;  // events of the substep between substep_begin and substep_end
  // tiling
  LevelSetDev LS;
  Tiling T;
  DevBox *d_boxes = nullptr;
  uint32_t *d_counts = nullptr;
  int *d_bounds = nullptr;
  uint32_t *h_pinned = nullptr;  // 64 KiB of pinned host memory for small readbacks (counters, migration table)
  double *d_energy = nullptr;
  int counts_cap = 0;
  bool compact_requested = false;
  bool in_substep = false;
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
  if (const char *e = getenv("MPMHIP_P2G_SPLIT")) c->p2g_split = atoi(e);
  if (const char *e = getenv("MPMHIP_P2G_WGS")) c->p2g_wgs = atoi(e) > 0 ? atoi(e) : 16384;
  c->reorder_interval = cfg->reorder_interval;
  if (const char *e = getenv("MPMHIP_REORDER_INTERVAL")) c->reorder_interval = atoi(e);
  const int ablate = getenv("MPMHIP_ABLATE") ? atoi(getenv("MPMHIP_ABLATE")) : 0;
  auto bail = [&](int code) { g_create_error = c->err; mpmhip_destroy(c); return code; };
  if (hipSetDevice(c->device) != hipSuccess) { fail(c, MPMHIP_EHIP, "hipSetDevice failed"); return bail(MPMHIP_EHIP); }
  Params &P = c->P;
  memset(&P, 0, sizeof P);
  memset(&c->T, 0, sizeof c->T);
  int maxnb = 0;
  for (int k = 0; k < 3; k++) {
    P.res[k] = cfg->res[k];
    P.g[k] = cfg->gravity[k];
    int nb = (cfg->res[k] + 1 + BS - 1) / BS + 1;
    if (nb > maxnb) maxnb = nb;
  }
  P.dx = cfg->dx; P.idx = 1.0f / cfg->dx; P.dt = cfg->dt;
  P.particle_gravity = cfg->particle_gravity; P.apic_damping = cfg->apic_damping; P.rpic_damping = cfg->rpic_damping;
  P.clean_boundary = cfg->clean_boundary;
  P.store_b = cfg->discard_apic_b ? 0 : 1;
  P.ablate = ablate;
  memset(&c->LS, 0, sizeof c->LS);
  c->LS.particle_collision = cfg->particle_collision;
  if (mpmhip_set_levelset(c, cfg->n_planes, &cfg->planes[0][0], cfg->friction) != MPMHIP_OK) return bail(MPMHIP_EINVAL);
  int kbits = 1;
  while ((1 << kbits) < maxnb) kbits++;
  if (kbits > 8) { fail(c, MPMHIP_EINVAL, "grid too large for 32-bit keys"); return bail(MPMHIP_EINVAL); }
  P.kbits = kbits;
  c->NB = 1u << (3 * kbits);
  P.nbw = c->NB / 32u;
  if (P.nbw == 0) P.nbw = 1;
  c->cap = cfg->max_particles;
  int64_t mb = cfg->max_blocks;
  if (mb <= 0) mb = c->cap / 48 + 4096;  // a block of 64 cells at >= ~1 particle/cell on average, plus slack
  if (mb > (int64_t)c->NB) mb = c->NB;
  P.max_blocks = (uint32_t)mb;

  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  A(dmalloc(&c->rg, (size_t)c->cap));
  A(dmalloc(&c->rp, (size_t)c->cap));
  A(dmalloc(&c->rb, (size_t)c->cap * BW));
  A(dmalloc(&c->key, (size_t)c->cap));
  A(dmalloc(&c->rank, (size_t)c->cap));
  A(dmalloc(&c->perm, (size_t)c->cap));
  A(dmalloc(&c->bits, (size_t)P.nbw));
  A(dmalloc(&c->blk_flag, (size_t)P.nbw * 32));
  A(dmalloc(&c->wprefix, (size_t)P.nbw));
  A(dmalloc(&c->fat_slot, (size_t)c->NB));
  A(dmalloc(&c->act_blk, (size_t)mb + 1));
  A(dmalloc(&c->act_start, (size_t)mb + 2));
  A(dmalloc(&c->cell_cnt, (size_t)mb * BC));
  A(dmalloc(&c->cell_start, (size_t)mb * BC + 1));
  c->bt_slots = (P.nbw + 255) / 256;
  const size_t n_slots64 = c->bt_slots + ((size_t)mb + CT_BLOCKS - 1) / CT_BLOCKS + 1;
  A(dmalloc(&c->scan_slots, n_slots64));
  A(dmalloc(&c->ticket, 2));
  A(dmalloc(&c->tiles, (size_t)mb * TN));
  A(dmalloc(&c->gridv, (size_t)mb * 8 * BC));
  A(dmalloc(&c->cnt, 1));
  A(hipHostMalloc((void **)&c->h_pinned, 65536, hipHostMallocDefault));
  A(dmalloc(&c->d_groups, (size_t)c->groups_cap));
  if (e != hipSuccess) {
    fail(c, MPMHIP_ENOMEM, "device allocation failed: %s (max_particles=%lld, max_blocks=%lld)", hipGetErrorString(e),
         (long long)c->cap, (long long)mb);
    return bail(MPMHIP_ENOMEM);
  }
  A(hipMemset(c->bits, 0, sizeof(uint32_t) * P.nbw));
  A(hipMemset(c->blk_flag, 0, (size_t)P.nbw * 32));
  A(hipMemset(c->cell_cnt, 0, sizeof(uint32_t) * (size_t)mb * BC));
  A(hipMemset(c->cell_start, 0, sizeof(uint32_t) * ((size_t)mb * BC + 1)));
  A(hipMemset(c->act_start, 0, sizeof(uint32_t) * ((size_t)mb + 2)));
  A(hipMemset(c->cnt, 0, sizeof(Counters)));
  A(hipMemset(c->scan_slots, 0, sizeof(unsigned long long) * n_slots64));  // epoch 0 is never used
  A(hipMemset(c->ticket, 0, 2 * sizeof(uint32_t)));
  A(hipMemset(c->fat_slot, 0, sizeof(uint32_t) * (size_t)c->NB));
  A(hipMemset(c->rb, 0, sizeof(float) * (size_t)c->cap * BW));
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
  hipFree(c->rg); hipFree(c->rp); hipFree(c->rb); hipFree(c->rg2); hipFree(c->rp2); hipFree(c->rb2);
  hipFree(c->key); hipFree(c->rank); hipFree(c->perm); hipFree(c->blk_flag); hipFree(c->bits); hipFree(c->wprefix);
  hipFree(c->fat_slot); hipFree(c->act_blk); hipFree(c->act_start); hipFree(c->cell_cnt);
  hipFree(c->cell_start); hipFree(c->scan_slots); hipFree(c->ticket); hipFree(c->tiles); hipFree(c->gridv); hipFree(c->dense);
  hipFree(c->cnt); hipFree(c->d_groups); hipFree(c->d_boxes); hipFree(c->d_counts); hipFree(c->d_bounds); if (c->h_pinned) hipHostFree(c->h_pinned); hipFree(c->d_energy);
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

int mpmhip_set_levelset_shapes(mpmhip_ctx *c, int32_t n, const mpmhip_shape *shapes, float friction) {
  if (!c || n < 0 || n > MPMHIP_MAX_SHAPES || (n > 0 && !shapes)) return MPMHIP_EINVAL;
  for (int i = 0; i < n; i++) {
    if (shapes[i].type < 0 || shapes[i].type > 2) return fail(c, MPMHIP_EINVAL, "shape %d: unknown type %d", i, shapes[i].type);
    if (shapes[i].type == 1 && !(shapes[i].p[3] > 0)) return fail(c, MPMHIP_EINVAL, "shape %d: sphere radius must be > 0", i);
    if (shapes[i].type == 2)
      for (int k = 0; k < 3; k++)
        if (!(shapes[i].p[k] < shapes[i].p[3 + k])) return fail(c, MPMHIP_EINVAL, "shape %d: cuboid needs lo < hi", i);
  }
  c->LS.n = n;
  c->LS.friction = friction;
  for (int i = 0; i < n; i++) {
    c->LS.s[i].type = shapes[i].type;
    c->LS.s[i].inside_out = shapes[i].inside_out;
    for (int k = 0; k < 6; k++) c->LS.s[i].p[k] = shapes[i].p[k];
  }
  return MPMHIP_OK;
}

int mpmhip_set_levelset(mpmhip_ctx *c, int32_t n_planes, const float *planes, float friction) {
  if (!c || n_planes < 0 || n_planes > 8 || (n_planes > 0 && !planes)) return MPMHIP_EINVAL;
  mpmhip_shape sh[8];
  memset(sh, 0, sizeof sh);
  for (int i = 0; i < n_planes; i++)
    for (int k = 0; k < 4; k++) sh[i].p[k] = planes[4 * i + k];
  return mpmhip_set_levelset_shapes(c, n_planes, sh, friction);
}

int mpmhip_add_group(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM]) {
  if (!c || !params) return MPMHIP_EINVAL;
  if (material < MPMHIP_VISCO || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
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

// synchronise and read the device counters; reports the sticky capacity error
static int read_counters(mpmhip_ctx *c, Counters &h) {
  // through the ctx's pinned page: an "async" copy into pageable memory is staged and costs ~50 us more
  Counters *pin = reinterpret_cast<Counters *>(c->h_pinned);
  HIPCHK(c, hipMemcpyAsync(pin, c->cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h = *pin;
  if (h.error & 1u)
    return fail(c, MPMHIP_ECAPACITY, "active blocks (%u) exceed max_blocks (%u): recreate the ctx with a larger max_blocks",
                h.n_active, c->P.max_blocks);
  if (h.error & 2u)
    return fail(c, MPMHIP_ECAPACITY, "a particle moved more than margin=%d cells outside this rank's brick between "
                "two migrations: migrate more often or raise the margin", c->T.margin);
  return MPMHIP_OK;
}

// discard_apic_b: bring the apic_b side array up to date from RecP.A before anybody reads it or invalidates A
static int ensure_b_current(mpmhip_ctx *c) {
  if (!c->b_stale) return MPMHIP_OK;
  if (!c->affine_valid) return fail(c, MPMHIP_EINVAL, "internal: apic_b is stale and the affine matrices are invalid");
  hipLaunchKernelGGL(k_recover_b, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const RecG *)c->rg,
                     (const RecP *)c->rp, c->rb, (const GroupParams *)c->d_groups);
  int rc = launch_check(c, "recover_b");
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->b_stale = false;
  return MPMHIP_OK;
}

int mpmhip_add_particles(mpmhip_ctx *c, int32_t group, int64_t n, const float *x, const float *v, const float *F,
                         const float *B, const float *aux) {
  if (!c || n < 0 || (n > 0 && !x)) return MPMHIP_EINVAL;
  if (group < 0 || group >= (int)c->groups.size()) return fail(c, MPMHIP_EINVAL, "unknown group %d", group);
  if (n == 0) return MPMHIP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->n_slots + n > c->cap)
    return fail(c, MPMHIP_ECAPACITY, "particle capacity exceeded: %lld + %lld > %lld", (long long)c->n_slots, (long long)n, (long long)c->cap);
  if (int rc = ensure_b_current(c)) return rc;  // A of every particle is recomputed from apic_b below
  const int mat = c->groups[group].type;
  // Jp = 1 (:204), j = 1 (:460), logJp = 0 (:595), visco_tau = 1000 (:65)
  const float aux0 = (mat == MPMHIP_SNOW || mat == MPMHIP_WATER) ? 1.0f : (mat == MPMHIP_VISCO ? 1000.0f : 0.0f);
  std::vector<RecG> hg((size_t)n);
  std::vector<RecP> hp((size_t)n);
  std::vector<float> hb((size_t)n * BW, 0.0f);
  for (int64_t i = 0; i < n; i++) {
    RecG &g = hg[i];
    RecP &p = hp[i];
    for (int k = 0; k < 3; k++) {
      g.x[k] = p.x[k] = x[3 * i + k];
      p.v[k] = v ? v[3 * i + k] : 0.0f;
    }
    for (int k = 0; k < 9; k++) {
      g.F[k] = F ? F[9 * i + k] : ((k % 4 == 0) ? 1.0f : 0.0f);
      p.A[k] = 0.0f;
      hb[(size_t)i * BW + k] = B ? B[9 * i + k] : 0.0f;
    }
    g.aux = aux ? aux[i] : aux0;
    g.gid = (uint32_t)group;
    p.mass = c->groups[group].p[0];
    g.pid = c->next_pid + (int32_t)i;
    g.pad = 0;
  }
  c->next_pid += (int32_t)n;
  HIPCHK(c, hipMemcpy(c->rg + c->n_slots, hg.data(), sizeof(RecG) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rp + c->n_slots, hp.data(), sizeof(RecP) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rb + (size_t)c->n_slots * BW, hb.data(), sizeof(float) * n * BW, hipMemcpyHostToDevice));
  c->n_slots += n;
  c->P.n_slots = (uint32_t)c->n_slots;
  c->sorted = c->keys_valid = c->affine_valid = false;
  return MPMHIP_OK;
}

int64_t mpmhip_num_particles(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  if (hipSetDevice(c->device) != hipSuccess) return MPMHIP_EHIP;
  Counters h;
  int rc = read_counters(c, h);
  return rc ? rc : c->n_slots - (int64_t)h.n_dead;
}

// host mirrors of the record arrays (download / upload are not on the hot path)
static int fetch_records(mpmhip_ctx *c, std::vector<RecG> &hg, std::vector<RecP> *hp, std::vector<float> *hb) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)c->n_slots;
  hg.resize(n);
  if (n) HIPCHK(c, hipMemcpy(hg.data(), c->rg, sizeof(RecG) * n, hipMemcpyDeviceToHost));
  if (hp) { hp->resize(n); if (n) HIPCHK(c, hipMemcpy(hp->data(), c->rp, sizeof(RecP) * n, hipMemcpyDeviceToHost)); }
  if (hb) { hb->resize(n * BW); if (n) HIPCHK(c, hipMemcpy(hb->data(), c->rb, sizeof(float) * n * BW, hipMemcpyDeviceToHost)); }
  return MPMHIP_OK;
}

int mpmhip_download(mpmhip_ctx *c, int32_t field, void *dst, int64_t n_capacity) {
  if (!c || !dst) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (field < MPMHIP_F_X || field > MPMHIP_F_ID) return fail(c, MPMHIP_EINVAL, "unknown field %d", field);
  if (field == MPMHIP_F_B)
    if (int rc = ensure_b_current(c)) return rc;
  std::vector<RecG> hg;
  std::vector<RecP> hp;
  std::vector<float> hb;
  int rc = fetch_records(c, hg, field == MPMHIP_F_V ? &hp : nullptr, field == MPMHIP_F_B ? &hb : nullptr);
  if (rc) return rc;
  int64_t m = 0;
  for (size_t i = 0; i < hg.size(); i++) {  // live slots, in slot order
    if (hg[i].pid < 0) continue;
    if (m >= n_capacity) return fail(c, MPMHIP_ECAPACITY, "download buffer holds %lld particles, more are alive", (long long)n_capacity);
    float *f = (float *)dst;
    int32_t *q = (int32_t *)dst;
    switch (field) {
      case MPMHIP_F_X: for (int k = 0; k < 3; k++) f[3 * m + k] = hg[i].x[k]; break;
      case MPMHIP_F_V: for (int k = 0; k < 3; k++) f[3 * m + k] = hp[i].v[k]; break;
      case MPMHIP_F_B: for (int k = 0; k < 9; k++) f[9 * m + k] = hb[i * BW + k]; break;
      case MPMHIP_F_F: for (int k = 0; k < 9; k++) f[9 * m + k] = hg[i].F[k]; break;
      case MPMHIP_F_AUX: f[m] = hg[i].aux; break;
      case MPMHIP_F_GID: q[m] = (int32_t)hg[i].gid; break;
      case MPMHIP_F_ID: q[m] = hg[i].pid; break;
    }
    m++;
  }
  return (int)m;
}

int mpmhip_upload(mpmhip_ctx *c, int32_t field, const void *src, int64_t n) {
  if (!c || !src) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (field < MPMHIP_F_X || field > MPMHIP_F_ID || field == MPMHIP_F_GID)
    return fail(c, MPMHIP_EINVAL, "field %d cannot be uploaded", field);
  std::vector<RecG> hg;
  std::vector<RecP> hp;
  std::vector<float> hb;
  int rc = ensure_b_current(c);
  if (rc) return rc;
  if ((rc = fetch_records(c, hg, &hp, &hb))) return rc;
  int64_t live = 0;
  for (auto &g : hg) live += g.pid >= 0;
  if (n != live) return fail(c, MPMHIP_EINVAL, "upload of %lld records but the ctx holds %lld particles", (long long)n, (long long)live);
  const float *f = (const float *)src;
  const int32_t *q = (const int32_t *)src;
  int64_t m = 0;
  for (size_t i = 0; i < hg.size(); i++) {
    if (hg[i].pid < 0) continue;
    switch (field) {
      case MPMHIP_F_X: for (int k = 0; k < 3; k++) hg[i].x[k] = hp[i].x[k] = f[3 * m + k]; break;
      case MPMHIP_F_V: for (int k = 0; k < 3; k++) hp[i].v[k] = f[3 * m + k]; break;
      case MPMHIP_F_B: for (int k = 0; k < 9; k++) hb[i * BW + k] = f[9 * m + k]; break;
      case MPMHIP_F_F: for (int k = 0; k < 9; k++) hg[i].F[k] = f[9 * m + k]; break;
      case MPMHIP_F_AUX: hg[i].aux = f[m]; break;
      case MPMHIP_F_ID:
        hg[i].pid = q[m] < 0 ? 0 : q[m];
        if (hg[i].pid + 1 > c->next_pid) c->next_pid = hg[i].pid + 1;
        break;
    }
    m++;
  }
  const size_t ns = hg.size();
  HIPCHK(c, hipMemcpy(c->rg, hg.data(), sizeof(RecG) * ns, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rp, hp.data(), sizeof(RecP) * ns, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rb, hb.data(), sizeof(float) * ns * BW, hipMemcpyHostToDevice));
  if (field == MPMHIP_F_X || field == MPMHIP_F_V) c->sorted = c->keys_valid = false;
  if (field == MPMHIP_F_B || field == MPMHIP_F_F || field == MPMHIP_F_AUX) c->affine_valid = false;
  return MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ phases
static int do_reorder(mpmhip_ctx *c);

static int do_sort(mpmhip_ctx *c) {
  Params &P = c->P;
  hipStream_t st = c->stream;
  const int pg = particle_grid(c->n_slots);
  if (!c->keys_valid)
    hipLaunchKernelGGL(k_build_keys, dim3(pg), dim3(256), 0, st, P, c->rg, c->rp, c->cnt, c->key, c->blk_flag);
  const uint32_t bt_chunks = (P.nbw + 255) / 256, ct_chunks = (P.max_blocks + CT_BLOCKS - 1) / CT_BLOCKS;
  const uint32_t epoch = ++c->sort_epoch;
  hipLaunchKernelGGL(k_block_table, dim3(std::min(bt_chunks, 512u)), dim3(256), 0, st, P, c->blk_flag, c->bits,
                     c->wprefix, c->act_blk, c->cnt, c->scan_slots, c->ticket, epoch);
  hipLaunchKernelGGL(k_rank, dim3(pg), dim3(256), 0, st, P, c->key, c->rank, c->cell_cnt, c->bits, c->wprefix);
  hipLaunchKernelGGL(k_cell_table, dim3(std::min(ct_chunks, 512u)), dim3(256), 0, st, P, c->cnt, c->cell_cnt,
                     c->act_start, c->cell_start, c->scan_slots + c->bt_slots, c->ticket, epoch);
  hipLaunchKernelGGL(k_perm, dim3(pg), dim3(256), 0, st, P, c->key, c->rank, c->cell_start, c->perm);
  c->sorted = true;
  c->keys_valid = false;  // key[] now holds cell indices
  int rc = launch_check(c, "sort");
  if (rc) return rc;
  if ((c->reorder_interval > 0 && c->substeps % c->reorder_interval == 0) || c->compact_requested) {  // src/mpm.cpp:811-813
    c->compact_requested = false;
    return do_reorder(c);
  }
  return MPMHIP_OK;
}

// physical reorder into sorted order + compaction of deleted slots (sort_allocator, src/mpm.cpp:752-768).
// Needs the live count on the host, hence one synchronisation: keep reorder_interval large.
static int do_reorder(mpmhip_ctx *c) {
  if (!c->rg2) {
    hipError_t e = dmalloc(&c->rg2, (size_t)c->cap);
    if (e == hipSuccess) e = dmalloc(&c->rp2, (size_t)c->cap);
    if (e == hipSuccess) e = dmalloc(&c->rb2, (size_t)c->cap * BW);
    if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "reorder buffers: %s", hipGetErrorString(e));
  }
  const int pg = particle_grid(c->n_slots * 4);
  hipLaunchKernelGGL(k_gather_records, dim3(pg), dim3(256), 0, c->stream, c->cnt, c->perm, (const float4 *)c->rg,
                     (const float4 *)c->rp, (const float4 *)c->rb, (float4 *)c->rg2, (float4 *)c->rp2, (float4 *)c->rb2);
  hipLaunchKernelGGL(k_identity_perm, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->cnt, c->perm);
  int rc = launch_check(c, "reorder");
  if (rc) return rc;
  Counters h;
  if ((rc = read_counters(c, h))) return rc;
  std::swap(c->rg, c->rg2); std::swap(c->rp, c->rp2); std::swap(c->rb, c->rb2);
  c->n_slots = h.n_sorted;
  c->P.n_slots = h.n_sorted;
  const uint32_t zero = 0;
  HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &zero, sizeof zero, hipMemcpyHostToDevice));
  return MPMHIP_OK;
}

static int do_p2g(mpmhip_ctx *c) {
  if (!c->affine_valid) {
    hipLaunchKernelGGL(k_affine, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, c->rg, c->rp, c->rb,
                       c->d_groups);
    c->affine_valid = true;
  }
  // one wavefront per block (all 27 nodes, all particles) measured fastest at 256^3 / 8 M: 0.187 ms against
  // 0.237 (two waves splitting the nodes) and 0.225 (two waves splitting the particles); knob: 10*NS + PS
  auto kern = k_p2g<1, 1, 2>;
  int nt = 64;
  switch (c->p2g_split) {
    case 21: kern = k_p2g<2, 1, 3>; nt = 128; break;
    case 12: kern = k_p2g<1, 2, 2>; nt = 128; break;
    default: break;
  }
  hipLaunchKernelGGL(kern, dim3(c->p2g_wgs), dim3(nt), 0, c->stream, c->P,
                     (const float4 *)c->rp, c->cnt, c->act_blk, c->cell_start, c->perm, c->d_groups, c->tiles);
  return launch_check(c, "p2g");
}
static int do_grid(mpmhip_ctx *c, int mode) {
  const bool per_cand = mode == 0 && c->n_slots < (2 << 20);  // small per-GPU problem: latency-bound, see k_grid
  auto kern = mode == 0 ? (per_cand ? k_grid<0, true> : k_grid<0, false>)
                        : (mode == 1 ? k_grid<1, false>
                                     : (mode == 2 ? k_grid<2, false> : (mode == 3 ? k_grid<3, false> : k_grid<4, false>)));
  hipLaunchKernelGGL(kern, dim3(per_cand ? 16384 : 4096), dim3(256), 0, c->stream, c->P, c->cnt, c->act_blk, c->bits, c->wprefix, c->tiles,
                     c->gridv, c->fat_slot, c->dense, c->T, (const DevBox *)c->d_boxes, c->LS);
  return launch_check(c, "grid");
}
static int do_g2p(mpmhip_ctx *c) {
  auto kern = k_g2p<256, 2, false>;
  switch (c->g2p_minw) {  // tuning knob: waves/SIMD target x (rolled gather loop ? 10 : 0)
    case 12: kern = k_g2p<256, 2, true>; break;
    case 3: kern = k_g2p<256, 3, false>; break;
    case 13: kern = k_g2p<256, 3, true>; break;
    case 14: kern = k_g2p<256, 4, true>; break;
    default: break;
  }
  hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, c->stream, c->P, (float4 *)c->rg, (float4 *)c->rp, (float4 *)c->rb,
                     c->cnt, c->act_blk, c->act_start, c->perm, c->d_groups, c->gridv, c->fat_slot, c->cnt, c->key,
                     c->blk_flag, c->LS);
  c->sorted = false;       // positions moved
  c->keys_valid = true;    // ... and their keys / block flags are ready for the next sort
  c->affine_valid = true;  // A was produced together with F
  if (!c->P.store_b) c->b_stale = true;
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
  return rc ? rc : do_g2p(c);
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
      if ((c->ev_level == 2 && k != PH_G2P) || (c->ev_level == 3 && k != PH_P2G)) continue;
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[k], c->ev_pool[i].e[k + 1]));
      c->phase_ms[k] += ms;
    }
    c->prof_substeps++;
  }
  c->ev_used = 0;
  return MPMHIP_OK;
}

static int do_halo_pack(mpmhip_ctx *c) {
  if (c->T.n_boxes == 0) return MPMHIP_OK;
  int nb = (int)((c->T.box_nodes + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_halo_pack, dim3(nb), dim3(256), 0, c->stream, c->P, c->T, (const DevBox *)c->d_boxes, c->bits,
                     c->wprefix, (const float4 *)c->tiles);
  return launch_check(c, "halo_pack");
}

int mpmhip_substep_begin(mpmhip_ctx *c) {  // sort, P2G, halo pack
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->cur_ev) return fail(c, MPMHIP_EINVAL, "substep_begin called twice without substep_end");
  int rc;
  mpmhip_ctx::Ev *ev = nullptr;
  const int lvl = c->profiling;
  if (lvl) {
    if (c->ev_used >= 4096 && (rc = collect_events(c))) return rc;
    if ((rc = get_events(c, &ev))) return rc;
    if (lvl == 1) HIPCHK(c, hipEventRecord(ev->e[0], c->stream));
  }
  if ((rc = do_sort(c))) return rc;
  if (ev && (lvl == 1 || lvl == 3)) HIPCHK(c, hipEventRecord(ev->e[1], c->stream));
  if ((rc = do_p2g(c))) return rc;
  if (ev && lvl == 3) HIPCHK(c, hipEventRecord(ev->e[2], c->stream));
  if ((rc = do_halo_pack(c))) return rc;
  if (ev && lvl == 1) HIPCHK(c, hipEventRecord(ev->e[2], c->stream));
  c->cur_ev = ev;
  c->in_substep = true;
  return MPMHIP_OK;
}

int mpmhip_substep_end(mpmhip_ctx *c) {  // grid (+ halo sum), G2P
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->in_substep) return fail(c, MPMHIP_EINVAL, "substep_end without substep_begin");
  int rc;
  mpmhip_ctx::Ev *ev = c->cur_ev;
  c->cur_ev = nullptr;
  c->in_substep = false;
  const int lvl = c->profiling;
  if (ev && lvl == 1) HIPCHK(c, hipEventRecord(ev->e[3], c->stream));
  if ((rc = do_grid(c, 0))) return rc;
  if (ev && (lvl == 1 || lvl == 2)) HIPCHK(c, hipEventRecord(ev->e[4], c->stream));
  if ((rc = do_g2p(c))) return rc;
  if (ev && (lvl == 1 || lvl == 2)) HIPCHK(c, hipEventRecord(ev->e[5], c->stream));
  c->t += c->P.dt;  // src/mpm.cpp:573
  c->substeps++;
  return MPMHIP_OK;
}

int mpmhip_substep(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  if (c->T.n_boxes > 0)
    return fail(c, MPMHIP_EINVAL, "this ctx has halo boxes: drive it with substep_begin / exchange / substep_end");
  int rc = mpmhip_substep_begin(c);
  return rc ? rc : mpmhip_substep_end(c);
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
  Counters h;
  return read_counters(c, h);
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
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->dense, src, nodes * sizeof(float4), hipMemcpyHostToDevice));
  return do_grid(c, 2);
}

// ------------------------------------------------------------------------------------------------ snapshots
// Whole-state save / load (reference: TC_IO serialization of MPM<dim> + ParticleAllocator, src/mpm.h:38-54,134-169,
// general_action "save"/"load" src/mpm.cpp:940-960).  The blob holds the raw records — including the P2G affine
// matrices — so a restart continues exactly where the run stopped (up to the in-cell summation order).
struct SnapHeader {
  char magic[8];  // "MPMHIP01"
  uint32_t abi, n_groups;
  int64_t n_slots, substeps;
  int32_t next_pid, b_stale, store_b, res[3];
  float t, request_t, dx, dt;
  uint32_t n_dead, pad;
};

static size_t snapshot_bytes(const mpmhip_ctx *c) {
  return sizeof(SnapHeader) + sizeof(GroupParams) * c->groups.size() +
         (size_t)c->n_slots * (sizeof(RecG) + sizeof(RecP) + sizeof(float) * BW);
}

int64_t mpmhip_snapshot_size(mpmhip_ctx *c) { return c ? (int64_t)snapshot_bytes(c) : MPMHIP_EINVAL; }

int mpmhip_snapshot_save(mpmhip_ctx *c, void *dst, size_t cap) {
  if (!c || !dst) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "snapshot inside a substep");
  if (cap < snapshot_bytes(c)) return fail(c, MPMHIP_ECAPACITY, "snapshot buffer too small: %zu < %zu", cap, snapshot_bytes(c));
  Counters hc;
  int rc = read_counters(c, hc);
  if (rc) return rc;
  if (!c->affine_valid) {  // make A current first (fresh uploads), so that the blob is self-consistent
    hipLaunchKernelGGL(k_affine, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, c->rg, c->rp, c->rb, c->d_groups);
    c->affine_valid = true;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  SnapHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "MPMHIP01", 8);
  h.abi = MPMHIP_ABI_VERSION; h.n_groups = (uint32_t)c->groups.size();
  h.n_slots = c->n_slots; h.substeps = c->substeps; h.next_pid = c->next_pid;
  h.b_stale = c->b_stale; h.store_b = c->P.store_b;
  for (int k = 0; k < 3; k++) h.res[k] = c->P.res[k];
  h.t = c->t; h.request_t = c->request_t; h.dx = c->P.dx; h.dt = c->P.dt; h.n_dead = hc.n_dead;
  char *p = (char *)dst;
  memcpy(p, &h, sizeof h); p += sizeof h;
  memcpy(p, c->groups.data(), sizeof(GroupParams) * c->groups.size()); p += sizeof(GroupParams) * c->groups.size();
  const size_t n = (size_t)c->n_slots;
  if (n) {
    HIPCHK(c, hipMemcpy(p, c->rg, sizeof(RecG) * n, hipMemcpyDeviceToHost)); p += sizeof(RecG) * n;
    HIPCHK(c, hipMemcpy(p, c->rp, sizeof(RecP) * n, hipMemcpyDeviceToHost)); p += sizeof(RecP) * n;
    HIPCHK(c, hipMemcpy(p, c->rb, sizeof(float) * BW * n, hipMemcpyDeviceToHost));
  }
  return MPMHIP_OK;
}

int mpmhip_snapshot_load(mpmhip_ctx *c, const void *src, size_t size) {
  if (!c || !src || size < sizeof(SnapHeader)) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  SnapHeader h;
  memcpy(&h, src, sizeof h);
  if (memcmp(h.magic, "MPMHIP01", 8) != 0 || h.abi != MPMHIP_ABI_VERSION) return fail(c, MPMHIP_EINVAL, "not a libmpmhip snapshot of this ABI version");
  for (int k = 0; k < 3; k++)
    if (h.res[k] != c->P.res[k]) return fail(c, MPMHIP_EINVAL, "snapshot is of a %dx%dx%d grid", h.res[0], h.res[1], h.res[2]);
  if (h.dx != c->P.dx) return fail(c, MPMHIP_EINVAL, "snapshot has delta_x = %g, the ctx %g", h.dx, c->P.dx);
  if (h.n_slots < 0 || h.n_slots > c->cap) return fail(c, MPMHIP_ECAPACITY, "snapshot holds %lld particle slots, capacity is %lld", (long long)h.n_slots, (long long)c->cap);
  if ((int)h.n_groups > c->groups_cap) return fail(c, MPMHIP_ECAPACITY, "snapshot holds %u groups", h.n_groups);
  const size_t n = (size_t)h.n_slots;
  if (size < sizeof h + sizeof(GroupParams) * h.n_groups + n * (sizeof(RecG) + sizeof(RecP) + sizeof(float) * BW))
    return fail(c, MPMHIP_EINVAL, "snapshot is truncated");
  const char *p = (const char *)src + sizeof h;
  c->groups.assign((const GroupParams *)p, (const GroupParams *)p + h.n_groups); p += sizeof(GroupParams) * h.n_groups;
  if (h.n_groups) HIPCHK(c, hipMemcpy(c->d_groups, c->groups.data(), sizeof(GroupParams) * h.n_groups, hipMemcpyHostToDevice));
  if (n) {
    HIPCHK(c, hipMemcpy(c->rg, p, sizeof(RecG) * n, hipMemcpyHostToDevice)); p += sizeof(RecG) * n;
    HIPCHK(c, hipMemcpy(c->rp, p, sizeof(RecP) * n, hipMemcpyHostToDevice)); p += sizeof(RecP) * n;
    HIPCHK(c, hipMemcpy(c->rb, p, sizeof(float) * BW * n, hipMemcpyHostToDevice));
  }
  c->n_slots = h.n_slots; c->P.n_slots = (uint32_t)h.n_slots;
  c->substeps = h.substeps; c->next_pid = h.next_pid;
  c->t = h.t; c->request_t = h.request_t;
  // A travels in the records; apic_b is current only if the saving ctx kept it up to date
  c->affine_valid = true;
  c->b_stale = h.b_stale != 0 || (h.store_b == 0);
  if (c->P.store_b && c->b_stale) {  // this ctx keeps apic_b: rebuild it from A once
    int rc = ensure_b_current(c);
    if (rc) return rc;
  }
  c->sorted = c->keys_valid = false;  // keys and block flags are rebuilt by the next sort
  HIPCHK(c, hipMemset(c->blk_flag, 0, (size_t)c->P.nbw * 32));
  Counters hc;
  memset(&hc, 0, sizeof hc);
  hc.n_dead = h.n_dead;
  HIPCHK(c, hipMemcpy(c->cnt, &hc, sizeof hc, hipMemcpyHostToDevice));
  return MPMHIP_OK;
}

// MPM<dim>::calculate_energy (src/mpm.cpp:1078-1110): sort + P2G, kinetic energy of the grid, potential energy
// of the particles.  Leaves the ctx sorted with fresh P2G tiles (like the reference, which leaves its grid rasterized).
int mpmhip_calculate_energy(mpmhip_ctx *c, double *kinetic, double *potential) {
  if (!c || !kinetic || !potential) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->T.n_boxes > 0) return fail(c, MPMHIP_ENOTIMPL, "calculate_energy on a tiled ctx (sum the ranks' shares on the caller side)");
  int rc;
  if ((rc = do_sort(c))) return rc;
  if ((rc = do_p2g(c))) return rc;
  if (!c->d_energy) HIPCHK(c, dmalloc(&c->d_energy, 4));
  HIPCHK(c, hipMemsetAsync(c->d_energy, 0, 4 * sizeof(double), c->stream));
  float4 *const dense_saved = c->dense;
  c->dense = reinterpret_cast<float4 *>(c->d_energy);  // k_grid<4> accumulates into its `dense` argument
  rc = do_grid(c, 4);
  c->dense = dense_saved;
  if (rc) return rc;
  double *acc = c->d_energy;
  hipLaunchKernelGGL(k_potential_energy, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const RecG *)c->rg,
                     (const GroupParams *)c->d_groups, acc + 1);
  if ((rc = launch_check(c, "potential_energy"))) return rc;
  double h[3];
  HIPCHK(c, hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *kinetic = h[0];
  *potential = h[1];
  if (h[2] != 0.0)
    return fail(c, MPMHIP_ENOTIMPL, "%.0f particles are of a type without potential_energy() (reference: TC_NOT_IMPLEMENTED); "
                "kinetic energy is valid", h[2]);
  return MPMHIP_OK;
}

int mpmhip_set_profiling(mpmhip_ctx *c, int32_t level) {
  if (!c || level < 0 || level > 3) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "set_profiling inside a substep");
  int rc = collect_events(c);  // pending events belong to the old level
  c->profiling = level;
  c->ev_level = level;
  return rc;
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
  if ((rc = read_counters(c, h))) return rc;
  int w = snprintf(json, cap,
                   "{\"substeps\":%lld,\"particles\":%lld,\"active_blocks\":%u,\"phases\":{\"sort\":%.6f,\"p2g\":%.6f,"
                   "\"exchange\":%.6f,\"grid\":%.6f,\"g2p\":%.6f}}",
                   (long long)c->prof_substeps, (long long)(c->n_slots - h.n_dead), h.n_active, c->phase_ms[PH_SORT],
                   c->phase_ms[PH_P2G], c->phase_ms[PH_EXCH], c->phase_ms[PH_GRID], c->phase_ms[PH_G2P]);
  return (w < 0 || (size_t)w >= cap) ? fail(c, MPMHIP_EINVAL, "profile buffer too small") : MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ tiling (host)
int mpmhip_set_partition(mpmhip_ctx *c, int32_t rank, const int32_t dims[3], const int32_t *cuts_x,
                         const int32_t *cuts_y, const int32_t *cuts_z, int32_t margin) {
  if (!c || !dims || !cuts_x || !cuts_y || !cuts_z) return MPMHIP_EINVAL;
  const int32_t *cuts[3] = {cuts_x, cuts_y, cuts_z};
  Tiling T;
  memset(&T, 0, sizeof T);
  int world = 1;
  for (int a = 0; a < 3; a++) {
    if (dims[a] < 1 || dims[a] > MPMHIP_MAX_PARTS) return fail(c, MPMHIP_EINVAL, "dims[%d]=%d outside [1,%d]", a, dims[a], MPMHIP_MAX_PARTS);
    T.dims[a] = dims[a];
    world *= dims[a];
    for (int k = 0; k <= dims[a]; k++) {
      T.cuts[a][k] = cuts[a][k];
      if (k > 0 && cuts[a][k] <= cuts[a][k - 1]) return fail(c, MPMHIP_EINVAL, "cuts of axis %d are not increasing", a);
    }
    if (cuts[a][0] != 0 || cuts[a][dims[a]] < c->P.res[a]) return fail(c, MPMHIP_EINVAL, "cuts of axis %d do not cover [0,res)", a);
  }
  if (rank < 0 || rank >= world) return fail(c, MPMHIP_EINVAL, "rank %d of %d", rank, world);
  if (margin < 1) return fail(c, MPMHIP_EINVAL, "margin must be >= 1 cell");
  const int pc[3] = {rank / (dims[1] * dims[2]), (rank / dims[2]) % dims[1], rank % dims[2]};
  for (int a = 0; a < 3; a++) { T.lo[a] = T.cuts[a][pc[a]]; T.hi[a] = T.cuts[a][pc[a] + 1]; }
  T.enabled = 1; T.rank = rank; T.margin = margin;
  c->T = T;  // halo boxes (if any) must be set again
  return MPMHIP_OK;
}

int mpmhip_set_halo(mpmhip_ctx *c, int32_t n, const mpmhip_halo_box *boxes) {
  if (!c || n < 0 || n > MPMHIP_MAX_HALO_BOXES || (n > 0 && !boxes)) return MPMHIP_EINVAL;
  if (n > 0 && !c->T.enabled) return fail(c, MPMHIP_EINVAL, "set_halo needs set_partition first");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  Tiling &T = c->T;
  T.n_boxes = 0; T.box_nodes = 0;
  if (n == 0) return MPMHIP_OK;
  // node box this rank's particles can touch: base cells in [lo-margin, hi+margin), stencil base..base+2
  int nlo[3], nhi[3];
  for (int a = 0; a < 3; a++) {
    nlo[a] = std::max(0, T.lo[a] - T.margin);
    nhi[a] = std::min(c->P.res[a] + 1, T.hi[a] + T.margin + 2);
    T.int_lo[a] = nlo[a]; T.int_hi[a] = nhi[a];
  }
  std::vector<DevBox> hb((size_t)n);
  uint64_t off = 0;
  bool empty_interior = false;
  for (int i = 0; i < n; i++) {
    const mpmhip_halo_box &b = boxes[i];
    if (i > 0 && b.peer < boxes[i - 1].peer) return fail(c, MPMHIP_EINVAL, "halo boxes must be sorted by peer rank");
    if (b.peer == T.rank || !b.send || !b.recv) return fail(c, MPMHIP_EINVAL, "halo box %d: bad peer or null buffer", i);
    bool proper = false;
    for (int a = 0; a < 3; a++) {
      if (b.lo[a] < 0 || b.hi[a] <= b.lo[a] || b.hi[a] > c->P.res[a] + 1) return fail(c, MPMHIP_EINVAL, "halo box %d: bad extent on axis %d", i, a);
      hb[i].lo[a] = b.lo[a]; hb[i].dim[a] = b.hi[a] - b.lo[a];
      // shrink the overlap-free interior along every axis where the box is a proper sub-range of the node box
      if (b.lo[a] <= nlo[a] && b.hi[a] >= nhi[a]) continue;
      proper = true;
      if (b.lo[a] <= nlo[a]) T.int_lo[a] = std::max(T.int_lo[a], b.hi[a]);
      else if (b.hi[a] >= nhi[a]) T.int_hi[a] = std::min(T.int_hi[a], b.lo[a]);
      else empty_interior = true;
    }
    if (!proper) empty_interior = true;
    hb[i].peer = b.peer; hb[i].off = (uint32_t)off;
    hb[i].send = (float4 *)b.send; hb[i].recv = (const float4 *)b.recv;
    off += (uint64_t)hb[i].dim[0] * hb[i].dim[1] * hb[i].dim[2];
    if (off >= (1ull << 31)) return fail(c, MPMHIP_EINVAL, "halo boxes too large");
  }
  if (empty_interior) for (int a = 0; a < 3; a++) T.int_hi[a] = T.int_lo[a];
  if (!c->d_boxes) HIPCHK(c, dmalloc(&c->d_boxes, (size_t)MPMHIP_MAX_HALO_BOXES));
  HIPCHK(c, hipMemcpy(c->d_boxes, hb.data(), sizeof(DevBox) * n, hipMemcpyHostToDevice));
  T.n_boxes = n; T.box_nodes = (uint32_t)off;
  return MPMHIP_OK;
}

int mpmhip_halo_pack(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "halo_pack");
  return rc ? rc : do_halo_pack(c);
}

static int ensure_counts(mpmhip_ctx *c, int world) {
  if (!c->T.enabled) return fail(c, MPMHIP_EINVAL, "no partition set");
  if (world != c->T.dims[0] * c->T.dims[1] * c->T.dims[2]) return fail(c, MPMHIP_EINVAL, "world=%d does not match the partition", world);
  if (c->counts_cap < world) {
    hipFree(c->d_counts);
    c->d_counts = nullptr;
    HIPCHK(c, dmalloc(&c->d_counts, (size_t)world + 6));  // + the 6 bounds of mpmhip_migration_scan
    c->counts_cap = world;
  }
  return MPMHIP_OK;
}

// one pass over the particles, one synchronisation: leaver counts per destination + bounding box of the base cells
int mpmhip_migration_scan(mpmhip_ctx *c, int32_t world, int64_t *counts, int32_t lo[3], int32_t hi[3]) {
  if (!c || !counts) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure_counts(c, world);
  if (rc) return rc;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "migration inside a substep");
  if ((size_t)world + 6 + sizeof(Counters) / 4 > 65536 / 4) return fail(c, MPMHIP_EINVAL, "world too large");
  hipLaunchKernelGGL(k_scan_init, dim3((world + 6 + 255) / 256), dim3(256), 0, c->stream, c->d_counts, world);
  int grid = particle_grid(c->n_slots);
  if (grid > 128) grid = 128;  // few workgroups: 6 same-address atomics each for the bounds
  hipLaunchKernelGGL(k_leaver_count, dim3(grid), dim3(256), 0, c->stream, c->P, c->T, (const float4 *)c->rg, c->d_counts,
                     reinterpret_cast<int *>(c->d_counts + world), c->cnt);
  if ((rc = launch_check(c, "leaver_count"))) return rc;
  uint32_t *h = c->h_pinned + sizeof(Counters) / 4;  // behind the counters read_counters() fetches
  HIPCHK(c, hipMemcpyAsync(h, c->d_counts, sizeof(uint32_t) * ((size_t)world + 6), hipMemcpyDeviceToHost, c->stream));
  Counters hc;
  if ((rc = read_counters(c, hc))) return rc;  // the ONE synchronisation; reports the margin violation
  for (int i = 0; i < world; i++) counts[i] = h[i];
  for (int k = 0; k < 3; k++) {
    if (lo) lo[k] = (int32_t)h[world + k];
    if (hi) hi[k] = (int32_t)h[world + 3 + k];
  }
  return MPMHIP_OK;
}

int mpmhip_leaver_counts(mpmhip_ctx *c, int32_t world, int64_t *counts) {
  return mpmhip_migration_scan(c, world, counts, nullptr, nullptr);
}

int mpmhip_export_leavers(mpmhip_ctx *c, int32_t world, const int64_t *counts, void *dev_records) {
  if (!c || !counts) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure_counts(c, world);
  if (rc) return rc;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "migration inside a substep");
  std::vector<uint32_t> cur((size_t)world);
  uint64_t off = 0;
  for (int i = 0; i < world; i++) { cur[i] = (uint32_t)off; off += (uint64_t)counts[i]; }
  if (off == 0) return MPMHIP_OK;
  if (!dev_records) return MPMHIP_EINVAL;
  uint32_t *pin = c->h_pinned + 1024;  // stays untouched until the next migration: no synchronisation needed
  memcpy(pin, cur.data(), sizeof(uint32_t) * world);
  HIPCHK(c, hipMemcpyAsync(c->d_counts, pin, sizeof(uint32_t) * world, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_leaver_pack, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, c->T, (float4 *)c->rg,
                     (const float4 *)c->rp, (const float4 *)c->rb, c->key, c->d_counts, (float4 *)dev_records, c->cnt);
  return launch_check(c, "leaver_pack");
}

int mpmhip_import_particles(mpmhip_ctx *c, int64_t n, const void *dev_records) {
  if (!c || n < 0 || (n > 0 && !dev_records)) return MPMHIP_EINVAL;
  if (n == 0) return MPMHIP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep || c->sorted) return fail(c, MPMHIP_EINVAL, "import_particles between sort and G2P");
  if (c->n_slots + n > c->cap)
    return fail(c, MPMHIP_ECAPACITY, "particle capacity exceeded on import: %lld + %lld > %lld (request_compaction or a larger max_particles)",
                (long long)c->n_slots, (long long)n, (long long)c->cap);
  hipLaunchKernelGGL(k_import, dim3(particle_grid(n)), dim3(256), 0, c->stream, c->P, (uint32_t)n, (uint32_t)c->n_slots,
                     (const float4 *)dev_records, (float4 *)c->rg, (float4 *)c->rp, (float4 *)c->rb, c->key, c->blk_flag,
                     c->cnt);
  c->n_slots += n;
  c->P.n_slots = (uint32_t)c->n_slots;
  return launch_check(c, "import");
}

int mpmhip_active_bounds(mpmhip_ctx *c, int32_t lo[3], int32_t hi[3]) {
  if (!c || !lo || !hi) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_bounds) HIPCHK(c, dmalloc(&c->d_bounds, 6));
  const int init[6] = {1 << 30, 1 << 30, 1 << 30, -1, -1, -1};
  int h[6];
  HIPCHK(c, hipMemcpyAsync(c->d_bounds, init, sizeof init, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // `init` lives on this stack frame
  hipLaunchKernelGGL(k_active_bounds, dim3(64), dim3(256), 0, c->stream, c->P, (const Counters *)c->cnt,
                     (const uint32_t *)c->act_blk, c->d_bounds);
  int rc = launch_check(c, "active_bounds");
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(h, c->d_bounds, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 3; k++) { lo[k] = h[k]; hi[k] = h[3 + k]; }
  return MPMHIP_OK;
}

int64_t mpmhip_num_slots(mpmhip_ctx *c) { return c ? c->n_slots : MPMHIP_EINVAL; }

int mpmhip_request_compaction(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  c->compact_requested = true;
  return MPMHIP_OK;
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
  if (material < MPMHIP_VISCO || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
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
                            const float *cdg, float *F, float *aux, float *next_force) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  GroupParams g;
  int rc = make_group(c, material, params, g);
  if (rc) return rc;
  float *dC, *dF, *dA, *dO = nullptr;
  HIPCHK(c, dmalloc(&dC, 9 * n)); HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dA, n));
  if (next_force) HIPCHK(c, dmalloc(&dO, 9 * n));
  HIPCHK(c, hipMemcpy(dC, cdg, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dA, aux, sizeof(float) * n, hipMemcpyHostToDevice));
  rc = run_debug(c, k_debug_plasticity, g, n, (const float *)dC, dF, dA, dO);
  if (!rc) {
    HIPCHK(c, hipMemcpy(F, dF, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(aux, dA, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (next_force) HIPCHK(c, hipMemcpy(next_force, dO, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
  }
  hipFree(dC); hipFree(dF); hipFree(dA); hipFree(dO);
  return rc;
}

}  // extern "C"
