// taichi_mpm_amd/csrc/mpm_common.h — constants, particle records, kernel parameter blocks, Morton keys, block bitmap helpers
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mpm_math.h"

namespace mpm {

// The 64-byte particle records G2P writes are not touched again before the next substep's P2G / G2P.  When they
// are far larger than the caches (128 B per particle against the 256 MB of infinity cache), storing them with the
// non-temporal hint stops them evicting the tiles and index arrays the following kernels read (C3, 8 M particles:
// 0.682 -> 0.669 ms per substep, most of it in k_p2g); when they fit (C2, 1 M particles: the next P2G finds them
// in cache) the hint costs 5 %, hence the size switch.  The same hint on the record LOADS is harmful at any size
// (k_p2g 0.18 -> 0.32 ms): a lane fetches its record with four 16-byte loads, and the line has to survive between them.
constexpr uint32_t NT_STORE_MIN_SLOTS = 2u << 20;
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_rec(float4 *p, const float4 &v, bool nt) {
  if (nt) {  // wave-uniform
    nt_f4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4 *>(p));
  } else {
    *p = v;
  }
}



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
  uint32_t run_heads;  // k_rank: runs of equal keys in adjacent slots seen by this sort (statistics for the next one)
  uint32_t rank_mode;  // k_rank path of the NEXT sort: 0 one global atomic per run, 1 LDS hash per 1024 slots
  uint32_t n_own;      // entries of the grid pass's owner list (k_cell_table -> k_grid): touched grid blocks of this sort
};

struct Params {
  int res[3];
  float dx, idx, dt;
  float t;            // current_t of the substep in flight (dynamic level sets sample at this time)
  float g[3];
  int particle_gravity;
  float apic_damping, rpic_damping;
  int clean_boundary;
  int kbits;         // Morton bits per axis
  uint32_t nbw;      // bitmap words = 8^kbits / 32
  uint32_t max_blocks;
  uint32_t n_slots;  // particle slots in use (host-known)
  int store_b;       // keep apic_b in the side array
  int particle_collision;  // particle_collision_resolution after G2P (src/mpm.cpp:566-569)
  int clamp_pos;     // generic transfer path (optimized = false): positions clamped into [0, res - eps] (src/transfer.cpp:668-670)
  int test_small_rank;  // TEST KNOB (env MPMHIP_TEST_SMALL_RANK, results valid): a 3-bit rank field in k_rank's packed words
  int ablate;        // only read by -DMPMHIP_ABLATE_BUILD libraries (profiles/ A/B builds; results invalid): env MPMHIP_ABLATE,
                     // 1 no G2P stores, 2 no constitutive update, 4 no 27-tap gather, 8 no P2G merge, 16 / 32 P2G capped at 8 / 6
                     // particles per cell, 64 / 128 P2G writes 150 of 216 tile nodes / the grid pass reads 5 of 8 tiles (quad-tile bound).  The default
                     // library compiles every use of it away (MPM_ABLATE below is the constant false).
  uint32_t *pidc;    // deterministic mode only (else null): creation id per SLOT, 4 bytes beside key[] — whoever writes a slot's key writes
                     // its id here (k_g2p / k_g2p_packed / k_g2p_rigid at the sorted position, k_build_keys, k_import), so k_cell_order
                     // gathers 4-byte words instead of one 64-byte record line per particle (k_sort.h)
};
#ifdef MPMHIP_ABLATE_BUILD
#define MPM_ABLATE(P, bit) (((P).ablate & (bit)) != 0)
#else
#define MPM_ABLATE(P, bit) false
#endif

// multi-GPU tiling (include/mpmhip.h, "Multi-GPU tiling"): partition of the cell space into bricks + halo boxes
struct Tiling {
  int enabled, rank;
  int dims[3];
  int cuts[3][MPMHIP_MAX_PARTS + 1];
  int lo[3], hi[3];          // this rank's brick, cells
  int margin;
  int n_boxes;
  uint32_t box_nodes;        // total nodes over all halo boxes
  uint32_t box_blocks;       // total 4^3-node grid blocks over all halo boxes (a block cut by a box's faces counts whole): k_halo_pack's work items
  int int_lo[3], int_hi[3];  // node box that no halo box intersects (fast path of k_grid)
};
// Exchange/compute overlap of a tiled substep: work is split by whether it can touch a halo node.
//   phase 0: everything (untiled ctx, or overlap off)     phase 1: only work that touches a halo box ("boundary")
//   phase 2: only work that cannot ("interior": its node box [lo, lo+extent)^3 lies inside the overlap-free box)
// Boundary P2G + halo pack run first, the exchange then overlaps interior P2G / grid / G2P, boundary grid / G2P run
// when the peers' sums have arrived.  Extents: P2G tile 6 nodes, grid candidate 4, G2P 8 (all 8 grid blocks its
// tile reads must be interior).
// the part of Tiling the transfer kernels need (kernel arguments live in SGPRs: the full struct costs ~80 of them)
struct PhaseBox {
  int n_boxes;
  int int_lo[3], int_hi[3];
};
__host__ __device__ __forceinline__ PhaseBox phase_box(const Tiling &T) {
  PhaseBox b;
  b.n_boxes = T.n_boxes;
  for (int k = 0; k < 3; k++) { b.int_lo[k] = T.int_lo[k]; b.int_hi[k] = T.int_hi[k]; }
  return b;
}
template <typename TB>
__device__ __forceinline__ bool in_phase(const TB &T, int phase, int lo0, int lo1, int lo2, int extent) {
  if (phase == 0) return true;
  const bool interior = T.n_boxes == 0 ||
                        (lo0 >= T.int_lo[0] && lo0 + extent <= T.int_hi[0] && lo1 >= T.int_lo[1] &&
                         lo1 + extent <= T.int_hi[1] && lo2 >= T.int_lo[2] && lo2 + extent <= T.int_hi[2]);
  return (phase == 2) == interior;
}

struct DevBox {
  int lo[3], dim[3];
  int peer;
  uint32_t off;  // first node of this box in the concatenated (all boxes) node numbering
  uint32_t boff; // first grid block of this box in the concatenated block numbering (k_halo_pack: one wave per block)
  float4 *send;        // where k_halo_pack writes this rank's partial sums: a local send buffer, or — peer-write wires
                       // (tiled_api.h) — the box's place in the PEER's receive buffer, mapped into this process
  const float4 *recv;  // the peer's partial sums, read by k_grid
  uint32_t *flag;      // peer-write wires: this rank's word in the peer's array of halo epochs (else nullptr)
};

// grid blocks (4^3 nodes) a halo box [lo, lo + dim) touches per axis, and in all
__host__ __device__ __forceinline__ int box_blocks_axis(int lo, int dim) { return ((lo + dim - 1) >> 2) - (lo >> 2) + 1; }
__host__ __device__ __forceinline__ uint32_t box_blocks_of(const int lo[3], const int dim[3]) {
  return (uint32_t)box_blocks_axis(lo[0], dim[0]) * (uint32_t)box_blocks_axis(lo[1], dim[1]) * (uint32_t)box_blocks_axis(lo[2], dim[2]);
}

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


}  // namespace mpm
