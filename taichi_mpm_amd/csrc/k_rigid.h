// taichi_mpm_amd/csrc/k_rigid.h — CPIC rigid coupling: bodies, the grid's colored distance field, the particles' colours
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
//
// Reference: src/rigid_transfer.cpp (rasterize_rigid_boundary :17-115, gather_cdf :121-275), the rigid branches of
// src/transfer.cpp (block_op_rigid :367-463 and :706-835), src/mpm_rigid_body.cpp (boundary particles :130-252,
// advect_rigid_bodies :255-286), src/mpm_fwd.h:69-105 (GridState::states = 24 colour-tag bits, 2 per body, + body id).
//
// Layout.  A body's surface is sampled by "boundary particles" (offset in the body frame + the triangle they sit on);
// they are NOT material particles here: they live in their own array and never enter the sort.  The colored distance
// field exists only near those samples, so it is kept in 4^3-node PAGES handed out from a pool on first touch
// (cdf.slot[morton(block)] -> page, one CAS per new block) and cleared page by page before the next substep — no pass
// over the grid, no dependence on the particle blocks.  Per node: `mind` = (distance bits << 32 | body id + 1), updated
// with one 64-bit atomicMin (= the reference's "closer triangle wins" under its per-node spinlock; ties go to the lower
// body id instead of the first writer), and `tags` = the colour bits, updated with atomicOr.
// A material particle's colour word travels in the spare word of its RecG record; the per-substep results of
// gather_cdf (boundary normal / distance / near flag) go to a side array indexed by slot.
#pragma once
#include "mpm_common.h"
#include "k_joints.h"

namespace mpm {

constexpr int MAX_RIGID = 12;                    // GridState::max_num_rigid_bodies (body 0 = background, no surface)
constexpr uint32_t CDF_TAG_MASK = 0x00FFFFFFu;   // src/mpm_fwd.h:78-82
constexpr uint32_t CDF_STATE_MASK = 0xAAAAAAAAu; // src/mpm.h:36 (the "has colour" bit of every body)
constexpr unsigned long long CDF_EMPTY = ~0ull;
constexpr uint32_t CDF_LOCKED = 0xFFFFFFFEu;  // cdf.slot value while a page is being handed out (inside pass 0 of the rasterisation only)
constexpr uint32_t CDF_POOLS = 64;            // sub-pools of the page pool, chosen by the block's Morton key

struct RigidBodyDev {
  float pos[3], vel[3], omega[3];
  float R[9];      // body -> world, row-major
  float q[4];      // the same rotation as a quaternion (w, x, y, z)
  float mass, inv_mass;
  float inv_I[9];  // body frame, row-major
  float fric[2];
  float lin_damp, ang_damp;
  float axis[3];   // rotation_axis (all zero: unrestricted)
  int scripted;    // bit 0: position follows a script, bit 1: rotation follows a script
  float tmp_imp[3], tmp_trq[3];  // impulse / torque (about the centre of mass) collected by a transfer (apply_tmp_impulse)
};
struct RigidSample { float off[3]; int body; int elem; };  // boundary particle: offset from the centre of mass, body frame
struct RigidStep {  // what the host knows about a scripted body for one substep: poses at t and t + dt
  int has_pos, has_rot;
  float p0[3], p1[3], q0[4], q1[4];
};
struct RigidSteps { RigidStep s[MAX_RIGID]; };

struct CdfDev {
  uint32_t *slot;            // [8^kbits] Morton(block of 4^3 nodes) -> page, INVALID = none
  uint32_t *page_key;        // [max_pages] page -> Morton key (for the clear pass)
  unsigned long long *mind;  // [max_pages * 64]
  uint32_t *tags;            // [max_pages * 64]
  uint32_t *n_pages;         // [CDF_POOLS] pages handed out this substep, per sub-pool (one counter would serialise every hand-out)
  uint32_t max_pages;        // CDF_POOLS * pool_cap
  uint32_t pool_cap;         // pages per sub-pool; sub-pool q owns the pages [q pool_cap, (q + 1) pool_cap)
  uint32_t *rpage;           // bitmap over the REFERENCE's 4x4x8-node blocks: its rigid_page_map (src/mpm.cpp:1026-1076)
  int rpd[3];
  int nb_axis;               // blocks per axis of the Morton space (1 << kbits): node coordinates beyond it have no page
  uint32_t *error;           // the ctx's sticky error word (Counters::error); bit 2 (value 4): page pool exhausted
};
struct BndRec { float n[3]; float dist; uint32_t near; uint32_t epoch; float pad[2]; };  // gather_cdf's per-particle output (32 bytes);
                                                                                        // epoch: the gather that wrote it

__device__ __forceinline__ void rot_apply(const float R[9], const float v[3], float o[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2];
}
__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
// RigidBody::get_velocity_at
__device__ __forceinline__ void rigid_velocity_at(const RigidBodyDev &b, const float p[3], float v[3]) {
  const float r[3] = {p[0] - b.pos[0], p[1] - b.pos[1], p[2] - b.pos[2]};
  float w[3];
  cross3(b.omega, r, w);
  v[0] = b.vel[0] + w[0]; v[1] = b.vel[1] + w[1]; v[2] = b.vel[2] + w[2];
}
// What the transfer kernels read of a body per (particle, node) pair — pose centre, velocities, friction — mirrored in LDS
// once per workgroup: read from the body records in global memory inside the per-node loops, every iteration waited a full
// memory round trip (k_p2g_rigid: 320 of 447 us at 8 M particles with a paddle wheel, profiles/r03_p_cpic.txt).
struct RigidLite { float pos[3], vel[3], omega[3], fric[2], pad; };
__device__ __forceinline__ void load_rigid_lite(RigidLite *s, const RigidBodyDev *rb, int tid, int nt) {
  for (int t = tid; t < MAX_RIGID * 11; t += nt) {
    const int b = t / 11, f = t - 11 * b;
    const RigidBodyDev &B = rb[b];
    const float v = f < 3 ? B.pos[f] : (f < 6 ? B.vel[f - 3] : (f < 9 ? B.omega[f - 6] : B.fric[f - 9]));
    reinterpret_cast<float *>(s + b)[f] = v;
  }
}
__device__ __forceinline__ void rigid_velocity_at(const RigidLite &b, const float p[3], float v[3]) {
  const float r[3] = {p[0] - b.pos[0], p[1] - b.pos[1], p[2] - b.pos[2]};
  float w[3];
  cross3(b.omega, r, w);
  v[0] = b.vel[0] + w[0]; v[1] = b.vel[1] + w[1]; v[2] = b.vel[2] + w[2];
}
// RigidBody::apply_tmp_impulse.  The reference adds every (particle, node) impulse to the body under a spinlock; here a
// lane first sums its own impulses (and their torques about the body's centre) in registers, a wave then reduces the sums
// of its lanes per body, and ONE lane issues the six float atomics: all impulses of a scene land on the same six words of
// a body, and per-impulse atomics serialise there (measured: 39 ms instead of 0.17 ms per substep with 120 k coloured
// particles around one body).  The conversion to velocities happens once, in k_rigid_apply_tmp.
struct ImpulseAcc {
  float imp[3], trq[3];
  int body;  // -1: empty
};
__device__ __forceinline__ void acc_init(ImpulseAcc &A) { A.body = -1; A.imp[0] = A.imp[1] = A.imp[2] = A.trq[0] = A.trq[1] = A.trq[2] = 0.0f; }
__device__ __forceinline__ void acc_flush_lane(ImpulseAcc &A, RigidBodyDev *rb) {  // (a lane met a second body: rare)
  if (A.body >= 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { atomicAdd(&rb[A.body].tmp_imp[k], A.imp[k]); atomicAdd(&rb[A.body].tmp_trq[k], A.trq[k]); }
  }
  acc_init(A);
}
// (centre: the body's centre of mass — from the caller's LDS mirror where it has one)
__device__ __forceinline__ void acc_add(ImpulseAcc &A, RigidBodyDev *rb, int body, const float imp[3], const float at[3],
                                        const float *centre = nullptr) {
  if (A.body != body) { acc_flush_lane(A, rb); A.body = body; }
  const float *cpos = centre ? centre : rb[body].pos;
  const float r[3] = {at[0] - cpos[0], at[1] - cpos[1], at[2] - cpos[2]};
  float t[3];
  cross3(r, imp, t);
#pragma unroll
  for (int k = 0; k < 3; k++) { A.imp[k] += imp[k]; A.trq[k] += t[k]; }
}
// all 64 lanes of the wave must call this together
__device__ __forceinline__ void acc_flush_wave(ImpulseAcc &A, RigidBodyDev *rb) {
  unsigned long long pending = __ballot(A.body >= 0);
  while (pending) {  // wave-uniform: one round per distinct body among the lanes
    const int leader = __ffsll((long long)pending) - 1;
    const int body = __shfl(A.body, leader);
    const bool mine = A.body == body;
    float v[6];
#pragma unroll
    for (int k = 0; k < 3; k++) { v[k] = mine ? A.imp[k] : 0.0f; v[3 + k] = mine ? A.trq[k] : 0.0f; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] += __shfl_xor(v[k], off);
    if ((int)(threadIdx.x & 63) == leader) {
#pragma unroll
      for (int k = 0; k < 3; k++) { atomicAdd(&rb[body].tmp_imp[k], v[k]); atomicAdd(&rb[body].tmp_trq[k], v[3 + k]); }
    }
    if (mine) acc_init(A);
    pending = __ballot(A.body >= 0);
  }
}

// node (i, j, k) of the colored distance field: tags (24 bits), body id of the closest triangle (-1: none), distance (world)
__device__ __forceinline__ void cdf_node(const CdfDev &C, const Params &P, int i, int j, int k, uint32_t &tags, int &rid, float &dist) {
  tags = 0; rid = -1; dist = 0.0f;
  if (i < 0 || j < 0 || k < 0 || (i >> 2) >= C.nb_axis || (j >> 2) >= C.nb_axis || (k >> 2) >= C.nb_axis) return;
  const uint32_t pg = C.slot[morton3(i >> 2, j >> 2, k >> 2)];
  if (pg == INVALID) return;
  const size_t n = (size_t)pg * 64 + (((i & 3) << 4) | ((j & 3) << 2) | (k & 3));
  tags = C.tags[n];
  const unsigned long long m = C.mind[n];
  if (m != CDF_EMPTY) {
    rid = (int)(m & 0xFFu) - 1;
    dist = __uint_as_float((uint32_t)(m >> 32)) * P.dx;  // (the reference rescales by delta_x after the rasterisation, :77-78)
  }
}
// packed node word of the transfer kernels' LDS tiles: tags | (rid + 1) << 24   (= GridState::states)
__device__ __forceinline__ uint32_t cdf_node_word(const CdfDev &C, int i, int j, int k) {
  if (i < 0 || j < 0 || k < 0 || (i >> 2) >= C.nb_axis || (j >> 2) >= C.nb_axis || (k >> 2) >= C.nb_axis) return 0u;
  const uint32_t pg = C.slot[morton3(i >> 2, j >> 2, k >> 2)];
  if (pg == INVALID) return 0u;
  const size_t n = (size_t)pg * 64 + (((i & 3) << 4) | ((j & 3) << 2) | (k & 3));
  const unsigned long long m = C.mind[n];
  return (C.tags[n] & CDF_TAG_MASK) | (m != CDF_EMPTY ? ((uint32_t)(m & 0xFFu) << 24) : 0u);
}
// the colour test of the transfers (src/transfer.cpp:419-423): true = the node belongs to the other side of a body
__device__ __forceinline__ bool cdf_incompatible(uint32_t node_word, uint32_t pstate) {
  const uint32_t gs = node_word & CDF_TAG_MASK;
  const uint32_t mask = (gs & pstate & CDF_STATE_MASK) >> 1;
  return (gs & mask) != (pstate & mask);
}

// ---------------------------------------------------------------------------------------------- clear
// pages handed out by the previous substep: unlink and reset them (then the host zeroes the counter and the page bitmap)
__global__ __launch_bounds__(256) void k_cdf_clear(CdfDev C) {
  for (uint32_t q = blockIdx.y; q < CDF_POOLS; q += gridDim.y) {
    const uint32_t np = min(C.n_pages[q], C.pool_cap), base = q * C.pool_cap * 64u;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < np * 64u; t += gridDim.x * blockDim.x) {
      C.mind[base + t] = CDF_EMPTY;
      C.tags[base + t] = 0u;
      if ((t & 63u) == 0u) C.slot[C.page_key[(base + t) >> 6]] = INVALID;
    }
  }
}

// ---------------------------------------------------------------------------------------------- page hand-out
// One thread per boundary particle, before the rasterisation: the <= 8 blocks its 3^3 stencil touches get a page (whoever
// turns a block's slot from INVALID to LOCKED takes one and publishes its number; nobody waits), and the reference's rigid
// pages are marked: the block (4x4x8 nodes) of the particle's base node and its neighbours in the POSITIVE directions (the
// reference's loop runs over ind in {-1,0,1}^3 but keeps only 0 <= ind, src/mpm.cpp:1062-1064).
// Per PARTICLE, not per node, and with an uncached look first: hundreds of threads share a block, and a CAS from each of
// them on the one slot word serialises (measured: 258 us for 42 k boundary particles when every (particle, node) thread tried).
__device__ __forceinline__ bool sample_base(const Params &P, const RigidBodyDev *rb, const RigidSample &S, int base[3]) {
  const RigidBodyDev &B = rb[S.body];
  float w[3];
  rot_apply(B.R, S.off, w);
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float X = (w[k] + B.pos[k]) * P.idx;  // get_anchor_point
    ok = ok && X >= 0.5f && X < (float)P.res[k] - 1.5f;
    base[k] = (int)(X - 0.5f);
  }
  return ok;  // (the reference refuses boundary particles near the domain wall at creation, src/mpm_rigid_body.cpp:241-246)
}
__global__ __launch_bounds__(256) void k_cdf_alloc(Params P, CdfDev C, const RigidBodyDev *__restrict__ rb,
                                                   const RigidSample *__restrict__ smp, uint32_t n) {
  // Uniform trip count (every lane takes part in the shuffles).  Neighbouring boundary particles sit on the same triangle
  // row and mostly share pages and blocks: a lane only goes to memory for a page / block its left neighbour lane does not
  // name too — the rigid-page bitmap of a whole body is a few cache lines, and even LOADS of one line from every thread
  // of the launch queue up at its L2 bank (measured: 139 us for 42 k particles without this).
  const uint32_t stride = gridDim.x * blockDim.x, nloop = (n + stride - 1) / stride;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t it = 0; it < nloop; it++) {
    const uint32_t s = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    int base[3] = {0, 0, 0};
    const bool ok = s < n && sample_base(P, rb, smp[s], base);
    {
      const int bx = base[0] >> 2, by = base[1] >> 2, bz = base[2] >> 3;
#pragma unroll
      for (int o = 0; o < 8; o++) {
        const int px = bx + (o >> 2), py = by + ((o >> 1) & 1), pz = bz + (o & 1);
        uint32_t bit = INVALID;
        if (ok && px >= 0 && py >= 0 && pz >= 0 && px < C.rpd[0] && py < C.rpd[1] && pz < C.rpd[2]) bit = ((uint32_t)px * C.rpd[1] + py) * C.rpd[2] + pz;
        const uint32_t left = __shfl_up(bit, 1);
        if (bit != INVALID && (lane == 0u || bit != left) &&
            !((__hip_atomic_load(&C.rpage[bit >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 31)) & 1u))
          atomicOr(&C.rpage[bit >> 5], 1u << (bit & 31));
      }
    }
    const int x0 = base[0] >> 2, y0 = base[1] >> 2, z0 = base[2] >> 2;
    const int nx = ((base[0] + 2) >> 2) - x0, ny = ((base[1] + 2) >> 2) - y0, nz = ((base[2] + 2) >> 2) - z0;  // 0 or 1 more block per axis
#pragma unroll
    for (int o = 0; o < 8; o++) {
      const int ax = o >> 2, ay = (o >> 1) & 1, az = o & 1;
      uint32_t bk = INVALID;
      if (ok && ax <= nx && ay <= ny && az <= nz) bk = morton3(x0 + ax, y0 + ay, z0 + az);
      const uint32_t left = __shfl_up(bk, 1);
      if (bk == INVALID || (lane != 0u && bk == left)) continue;
      if (__hip_atomic_load(&C.slot[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != INVALID) continue;
      if (atomicCAS(&C.slot[bk], INVALID, CDF_LOCKED) != INVALID) continue;
      const uint32_t q = bk % CDF_POOLS, mine = atomicAdd(&C.n_pages[q], 1u);
      uint32_t pg = INVALID;
      if (mine >= C.pool_cap) atomicOr(C.error, 4u);  // pool exhausted: sticky error, reported by the next synchronising call
      else { pg = q * C.pool_cap + mine; C.page_key[pg] = bk; }
      __hip_atomic_store(&C.slot[bk], pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------- rasterize
// rasterize_rigid_boundary (src/rigid_transfer.cpp:17-78), one thread per (boundary particle, node); also marks the reference's
// rigid pages (blocks of 4x4x8 nodes) from the block of the particle's base node (src/mpm.cpp:1026-1076)
// One thread per (boundary particle, stencil node); every page exists (k_cdf_alloc ran before).
__global__ __launch_bounds__(256) void k_cdf_rasterize(Params P, CdfDev C, const RigidBodyDev *__restrict__ rb,
                                                       const RigidSample *__restrict__ smp, const float *__restrict__ elems,
                                                       uint32_t n) {
  for (uint32_t tt = blockIdx.x * blockDim.x + threadIdx.x; tt < n * 27u; tt += gridDim.x * blockDim.x) {
    const uint32_t s = tt / 27u, node27 = tt - s * 27u;
    const RigidSample S = smp[s];
    const RigidBodyDev &B = rb[S.body];
    int base[3];
    if (!sample_base(P, rb, S, base)) continue;
    // world-space triangle and the map world -> (edge coordinates, signed distance): inverse of [e1, e2, n]
    float v[3][3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      float t[3];
      rot_apply(B.R, elems + (size_t)S.elem * 9 + 3 * q, t);
#pragma unroll
      for (int k = 0; k < 3; k++) v[q][k] = t[k] + B.pos[k];
    }
    const float e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
    const float e2[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
    float nn[3];
    cross3(e1, e2, nn);
    const float nl = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    nn[0] /= nl; nn[1] /= nl; nn[2] /= nl;
    // M = [e1 e2 n] (columns); inverse by the adjugate, row r of M^-1 = cross of the other two columns / det
    float c12[3], c20[3], c01[3];
    cross3(e2, nn, c12); cross3(nn, e1, c20); cross3(e1, e2, c01);
    const float det = e1[0] * c12[0] + e1[1] * c12[1] + e1[2] * c12[2], id = 1.0f / det;
    const uint32_t body_bits = (uint32_t)S.body * 2u;
    {
      {
        {
          const int a = (int)(node27 / 9u), b = (int)((node27 / 3u) % 3u), c = (int)(node27 % 3u);
          const int gi = base[0] + a, gj = base[1] + b, gk = base[2] + c;
          const float d[3] = {gi * P.dx - v[0][0], gj * P.dx - v[0][1], gk * P.dx - v[0][2]};
          const float u0 = (c12[0] * d[0] + c12[1] * d[1] + c12[2] * d[2]) * id;
          const float u1 = (c20[0] * d[0] + c20[1] * d[1] + c20[2] * d[2]) * id;
          const float u2 = (c01[0] * d[0] + c01[1] * d[1] + c01[2] * d[2]) * id;
          if (!(0.0f <= u0 && 0.0f <= u1 && u0 + u1 <= 1.0f)) continue;
          const bool negative = u2 < 0.0f;
          const float dist = fabsf(u2) * P.idx;
          const uint32_t bk = morton3(gi >> 2, gj >> 2, gk >> 2);
          const uint32_t pg = C.slot[bk];
          if (pg == INVALID) continue;  // (pool exhausted)
          const size_t node = (size_t)pg * 64 + (((gi & 3) << 4) | ((gj & 3) << 2) | (gk & 3));
          atomicMin(&C.mind[node], ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)(S.body + 1));
          atomicOr(&C.tags[node], (2u + (negative ? 1u : 0u)) << body_bits);
        }
      }
    }
  }
}

// is the reference's rigid page holding node-block coordinates (bx, by, bz) of OUR 4^3 blocks set?  (its blocks are
// 4x4x8 nodes: two of ours stacked in z)
__device__ __forceinline__ bool rigid_page_of_block(const CdfDev &C, int bx, int by, int bz) {
  const int pz = bz >> 1;
  if (bx < 0 || by < 0 || pz < 0 || bx >= C.rpd[0] || by >= C.rpd[1] || pz >= C.rpd[2]) return false;
  const uint32_t bit = ((uint32_t)bx * C.rpd[1] + by) * C.rpd[2] + pz;
  return (C.rpage[bit >> 5] >> (bit & 31)) & 1u;
}
// active blocks the transfers handle with the colour-aware kernels: block_op_switch, src/transfer.cpp:570-576 — the
// block's page is in rigid_page_map.  (Blocks outside ignore colours altogether, like the reference's block_op_normal.)
// The flag goes to blk_rigid[a] (k_p2g / k_p2g_rigid) and into the TOP BIT of act_start[a]: k_g2p's chunk iterator loads
// act_start anyway, a second array there would put one more load — and with its vmcnt wait the prefetched records — on
// the iterator's critical path (measured: +20 % on k_g2p).  Only the RIGID variants of the G2P kernels see the bit.
constexpr uint32_t ACT_RIGID_BIT = 0x80000000u;
// `rigid_list[0 .. *n_rigid)` = those blocks, in no particular order (the host clears *n_rigid before the launch): the
// colour-aware kernels walk the list, so every one of their workgroups has a block to work on.
__global__ __launch_bounds__(256) void k_blk_rigid(Params P, const Counters *__restrict__ cnt, const uint32_t *__restrict__ act_blk,
                                                   CdfDev C, uint8_t *__restrict__ blk_rigid, uint32_t *__restrict__ act_start,
                                                   uint32_t *__restrict__ rigid_list, uint32_t *__restrict__ n_rigid) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < na; a += gridDim.x * blockDim.x) {
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    const bool r = rigid_page_of_block(C, bx, by, bz);
    blk_rigid[a] = r ? 1 : 0;
    const uint32_t s = act_start[a] & ~ACT_RIGID_BIT;  // (idempotent: the phase-level API may run this twice per sort)
    act_start[a] = r ? (s | ACT_RIGID_BIT) : s;
    if (r) rigid_list[atomicAdd(n_rigid, 1u)] = a;
  }
}

// ---------------------------------------------------------------------------------------------- gather_cdf
// 4x4 least-squares system in double precision (the host-side reference build solves it in double as well: the taichi
// core's own routine is not available, see oracle/taichi_shim); returns |det|, solves A r = y when it exceeds the guard
__device__ __forceinline__ double solve4(const float A[4][4], const float y[4], float r[4], double guard) {
  // A = X^T W X is symmetric positive semi-definite: L D L^T without pivoting, fully unrolled (registers, no scratch);
  // det A = d0 d1 d2 d3.  Double precision, like the reference build's solve.
  const double a00 = A[0][0], a10 = A[1][0], a11 = A[1][1], a20 = A[2][0], a21 = A[2][1], a22 = A[2][2], a30 = A[3][0], a31 = A[3][1],
               a32 = A[3][2], a33 = A[3][3];
  const double d0 = a00;
  if (!(d0 > 0.0)) return 0.0;
  const double l10 = a10 / d0, l20 = a20 / d0, l30 = a30 / d0;
  const double d1 = a11 - l10 * a10;
  if (!(d1 > 0.0)) return 0.0;
  const double l21 = (a21 - l20 * a10) / d1, l31 = (a31 - l30 * a10) / d1;
  const double d2 = a22 - l20 * a20 - l21 * l21 * d1;
  if (!(d2 > 0.0)) return 0.0;
  const double l32 = (a32 - l30 * a20 - l31 * l21 * d1) / d2;
  const double d3 = a33 - l30 * a30 - l31 * l31 * d1 - l32 * l32 * d2;
  const double det = d0 * d1 * d2 * d3;
  if (!(fabs(det) > guard)) return fabs(det);
  // L z = y, D w = z, L^T x = w
  const double z0 = y[0], z1 = y[1] - l10 * z0, z2 = y[2] - l20 * z0 - l21 * z1, z3 = y[3] - l30 * z0 - l31 * z1 - l32 * z2;
  const double x3 = z3 / d3, x2 = z2 / d2 - l32 * x3, x1 = z1 / d1 - l21 * x2 - l31 * x3, x0 = z0 / d0 - l10 * x1 - l20 * x2 - l30 * x3;
  r[0] = (float)x0; r[1] = (float)x1; r[2] = (float)x2; r[3] = (float)x3;
  return fabs(det);
}

// gather_cdf (src/rigid_transfer.cpp:121-275), one thread per particle slot: the particle gains colours from the grid
// (never changes one it has), then fits (normal, distance) of the boundary to the nodes of its own colour
__device__ __forceinline__ void gather_cdf_one(const Params &P, const CdfDev &C, RecG *__restrict__ rg, BndRec *__restrict__ bnd,
                                               uint32_t *__restrict__ cutting_counter, size_t i, uint32_t epoch, int ox, int oy, int oz,
                                               const uint32_t *s_tag, const int *s_rid, const float *s_d) {
  {
    BndRec out;
    out.n[0] = out.n[1] = out.n[2] = 0.0f; out.dist = 0.0f; out.near = 0u; out.epoch = epoch; out.pad[0] = out.pad[1] = 0.0f;
    if (rg[i].pid < 0) { bnd[i] = out; return; }
    const float pos[3] = {rg[i].x[0] * P.idx, rg[i].x[1] * P.idx, rg[i].x[2] * P.idx};
    {  // rigid_page_map->Test_Page(Linear_Offset(pos.cast<int>()))  (:142-146): elsewhere the colours stay as they are
      const int px = (int)pos[0] >> 2, py = (int)pos[1] >> 2, pz = (int)pos[2] >> 3;
      bool on = px >= 0 && py >= 0 && pz >= 0 && px < C.rpd[0] && py < C.rpd[1] && pz < C.rpd[2];
      if (on) {
        const uint32_t bit = ((uint32_t)px * C.rpd[1] + py) * C.rpd[2] + pz;
        on = (C.rpage[bit >> 5] >> (bit & 31)) & 1u;
      }
      if (!on) { bnd[i] = out; return; }
    }
    uint32_t pstate = rg[i].pad;
    int base[3];
    float w[3][3], rel[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      base[k] = (int)(pos[k] - 0.5f);
      rel[k] = pos[k] - (float)base[k];
      bspline_weights(rel[k], w[k]);
    }
    uint32_t ntag[27];
    int nrid[27];
    float nd[27];
    uint32_t all_b = 0u;
#pragma unroll
    for (int t = 0; t < 27; t++) {
      // (the block's 6^3 nodes are staged in LDS by the workgroup: a particle's stencil lies inside its block's tile)
      const int tn = ((base[0] - ox + t / 9) * TS + (base[1] - oy + (t / 3) % 3)) * TS + (base[2] - oz + t % 3);
      ntag[t] = s_tag[tn]; nrid[t] = s_rid[tn]; nd[t] = s_d[tn];
      all_b |= ntag[t] & CDF_STATE_MASK;
    }
    pstate &= (all_b + (all_b >> 1));  // unset the colours of bodies the particle no longer touches (:164)
    uint32_t to_add = all_b & ~pstate;
    while (to_add) {
      const uint32_t bit = to_add & (0u - to_add);
      to_add ^= bit;
      float wd[2] = {0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 27; t++) {
        if (nrid[t] == -1) continue;
        const float d = nd[t] * P.idx;
        const float weight = (w[0][t / 9] * w[1][(t / 3) % 3]) * w[2][t % 3];
        if (ntag[t] & bit) wd[(ntag[t] & (bit >> 1)) != 0u ? 1 : 0] += d * weight;
      }
      if (wd[0] + wd[1] > 1e-7f) {
        pstate |= bit | ((bit >> 1) * (wd[0] < wd[1] ? 1u : 0u));
        atomicAdd(cutting_counter, 1u);
      }
    }
    rg[i].pad = pstate;
    if (pstate != 0u) {
      float XtX[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, XtY[4] = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 27; t++) {
        if (nrid[t] == -1) continue;
        const uint32_t gs = ntag[t];
        if (gs == 0u) continue;
        const float dp[3] = {rel[0] - (float)(t / 9), rel[1] - (float)((t / 3) % 3), rel[2] - (float)(t % 3)};
        const float xp[4] = {-dp[0], -dp[1], -dp[2], 1.0f};
        const float d = nd[t] * P.idx;
        const float weight = (w[0][t / 9] * w[1][(t / 3) % 3]) * w[2][t % 3];
        const uint32_t mask = (gs & pstate & CDF_STATE_MASK) >> 1;
        float sgn = 0.0f;
        if ((gs & mask) == (pstate & mask)) sgn = 1.0f;  // same colour
        else {                                           // exactly one colour differs: the node counts with the negative distance
          const uint32_t diff = (gs & mask) ^ (pstate & mask);
          if (diff != 0u && (diff & (diff - 1u)) == 0u) sgn = -1.0f;
        }
        if (sgn == 0.0f) continue;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int c = 0; c < 4; c++) XtX[r][c] += (xp[r] * xp[c]) * weight;
        const float yv[4] = {-d * dp[0], -d * dp[1], -d * dp[2], d};
#pragma unroll
        for (int r = 0; r < 4; r++) XtY[r] += (sgn * yv[r]) * weight;
      }
      float r4[4] = {0, 0, 0, 0};
      const double det = solve4(XtX, XtY, r4, 1e-4);  // mpm_reconstruction_guard<3>()
      if (det > 1e-4) {
        out.near = 1u;
        out.dist = r4[3] * P.dx;
        const float l2 = r4[0] * r4[0] + r4[1] * r4[1] + r4[2] * r4[2];
        if (l2 > 1e-4f) {
          const float il = 1.0f / sqrtf(l2);
          out.n[0] = r4[0] * il; out.n[1] = r4[1] * il; out.n[2] = r4[2] * il;
        }
      }
    }
    bnd[i] = out;
  }
}
// One workgroup per active block that can hold such particles: the page test above uses the TRUNCATED position, the block a
// particle is sorted into its base node (half a cell lower), so besides the block's own page its neighbours in the positive
// directions are looked at.  Everything else is skipped without touching its records (the reference leaves those
// particles' colours alone too); their stale boundary records are told apart by the epoch.
__global__ __launch_bounds__(256) void k_gather_cdf(Params P, CdfDev C, RecG *__restrict__ rg, BndRec *__restrict__ bnd,
                                                    uint32_t *__restrict__ cutting_counter, const Counters *__restrict__ cnt,
                                                    const uint32_t *__restrict__ act_blk, const uint32_t *__restrict__ act_start,
                                                    const uint32_t *__restrict__ perm, uint32_t epoch) {
  __shared__ uint32_t s_tag[TN];
  __shared__ int s_rid[TN];
  __shared__ float s_d[TN];
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  for (uint32_t a = blockIdx.x; a < na; a += gridDim.x) {
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    bool cand = false;
#pragma unroll
    for (int o = 0; o < 8; o++) cand = cand || rigid_page_of_block(C, bx + (o >> 2), by + ((o >> 1) & 1), bz + 2 * (o & 1));
    if (!cand) continue;  // workgroup-uniform
    __syncthreads();
    for (int t = threadIdx.x; t < TN; t += blockDim.x)
      cdf_node(C, P, bx * BS + t / (TS * TS), by * BS + (t / TS) % TS, bz * BS + t % TS, s_tag[t], s_rid[t], s_d[t]);
    __syncthreads();
    const uint32_t p0 = act_start[a] & 0x7FFFFFFFu, p1 = act_start[a + 1] & 0x7FFFFFFFu;
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += blockDim.x)
      gather_cdf_one(P, C, rg, bnd, cutting_counter, perm[p], epoch, bx * BS, by * BS, bz * BS, s_tag, s_rid, s_d);
  }
}

// ---------------------------------------------------------------------------------------------- the bodies
// RigidBody::apply_tmp_velocity after a transfer: velocity += impulse / m, omega += R I^-1 R^T torque; sums cleared
// MPM::articulate (src/mpm.h:278-319): the joints' sequential Gauss-Seidel chain (k_joints.h), one lane; the bodies sit in
// LDS while it runs
__global__ __launch_bounds__(64) void k_articulate(RigidBodyDev *rb, int nb, const JointDev *__restrict__ joints, int nj, float dt,
                                                   int iterations) {
  __shared__ JointBody sb[MAX_RIGID];
  __shared__ JointPre pre[MAX_JOINTS];
  const int t = threadIdx.x;
  if (t < nb) {
    const RigidBodyDev &B = rb[t];
    JointBody &J = sb[t];
    for (int k = 0; k < 3; k++) { J.pos[k] = B.pos[k]; J.vel[k] = B.vel[k]; J.omega[k] = B.omega[k]; }
    for (int k = 0; k < 9; k++) { J.R[k] = B.R[k]; J.inv_I[k] = B.inv_I[k]; }
    J.inv_mass = B.inv_mass;
  }
  __syncthreads();
  if (t == 0) articulate(sb, nb, joints, pre, nj, dt, iterations);
  __syncthreads();
  if (t < nb) {
    for (int k = 0; k < 3; k++) { rb[t].vel[k] = sb[t].vel[k]; rb[t].omega[k] = sb[t].omega[k]; }
  }
}

// ------------------------------------------------------------------------------------------------ rigid body <-> level set
// MPM<dim>::rigid_body_levelset_collision (src/mpm_rigid_body.cpp:347-387; config key rigid_body_levelset_collision, called
// between normalize_grid and the grid boundary condition, src/mpm.cpp:535-538).  The reference walks its sorted particle list
// and every boundary particle below the level set gives its body an impulse IMMEDIATELY — the next boundary particle of that body
// sees the changed velocity (a Gauss-Seidel chain; summing the impulses of all penetrating particles instead would multiply the
// response by their number).  Reproduced as such: the boundary particles are kept in the reference's order — its sort key
// (SPGrid offset of the base node, src/mpm.cpp:785-790) with the position in the previous order as the tie-break, i.e. the
// history of stable sorts the reference's list goes through — and ONE lane applies the impulses in that order; finding the
// penetrating particles (pose, level set) is done by the whole workgroup.
__device__ __forceinline__ uint32_t spread_every_third(uint32_t v) {
  uint32_t r = 0;
#pragma unroll
  for (int b = 0; b < 10; b++) r |= ((v >> b) & 1u) << (3 * b);
  return r;
}
// SparseMask::Linear_Offset(i, j, k) >> data_bits for the reference's 4 x 4 x 8-node blocks (page bits: per level z most
// significant, then x, then y — pinned against the reference's own header in tests/test_oracle_kernel.py)
__device__ __forceinline__ uint32_t ref_node_key(int i, int j, int k) {
  const uint32_t blk = (spread_every_third((uint32_t)(k >> 3)) << 2) | (spread_every_third((uint32_t)(i >> 2)) << 1) |
                       spread_every_third((uint32_t)(j >> 2));
  return (blk << 7) | (uint32_t)((((i & 3) << 2) | (j & 3)) << 3 | (k & 7));
}
__device__ __forceinline__ void sample_world(const RigidBodyDev &B, const RigidSample &s, float p[3]) {
  rot_apply(B.R, s.off, p);
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] += B.pos[k];
}
__global__ __launch_bounds__(256) void k_rigid_ls_keys(Params P, const RigidBodyDev *__restrict__ rb, const RigidSample *__restrict__ smp,
                                                       uint32_t n, const uint32_t *__restrict__ rank, unsigned long long *__restrict__ keys,
                                                       uint32_t *__restrict__ vals) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    float p[3];
    sample_world(rb[smp[s].body], smp[s], p);
    int b[3];
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] = max((int)(p[k] * P.idx - 0.5f), 0);  // get_grid_base_pos (src/mpm.h:252-255)
    keys[s] = ((unsigned long long)ref_node_key(b[0], b[1], b[2]) << 32) | rank[s];
    vals[s] = s;
  }
}
struct RigidRestitution { float e[MAX_RIGID]; };
__global__ __launch_bounds__(1024) void k_rigid_ls_collide(Params P, const LevelSetDev *__restrict__ ls, RigidBodyDev *rb, int nb,
                                                           const RigidSample *__restrict__ smp, const uint32_t *__restrict__ sorted,
                                                           uint32_t n, uint32_t *__restrict__ rank, RigidRestitution rest) {
  struct Hit { int body; float r[3], g[3]; };
  __shared__ JointBody sb[MAX_RIGID];
  __shared__ Hit hits[1024];
  __shared__ int wave_n[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < nb) {
    const RigidBodyDev &B = rb[t];
    JointBody &J = sb[t];
    for (int k = 0; k < 3; k++) { J.pos[k] = B.pos[k]; J.vel[k] = B.vel[k]; J.omega[k] = B.omega[k]; }
    for (int k = 0; k < 9; k++) { J.R[k] = B.R[k]; J.inv_I[k] = B.inv_I[k]; }
    J.inv_mass = B.inv_mass;
    j_to_world(J.R, J.inv_I, J.Iw);
  }
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t j = base + t;
    bool hit = false;
    Hit h;
    h.body = 0;
    if (j < n) {
      const uint32_t s = sorted[j];
      rank[s] = j;  // the position in this substep's order is the next substep's tie-break
      const RigidSample sm = smp[s];
      float p[3];
      sample_world(rb[sm.body], sm, p);
      float phi, g[3] = {0, 0, 0};
      if (levelset_eval(*ls, P.t, p, P.idx, phi, g) && phi < 0.0f) {
        hit = true;
        h.body = sm.body;
        for (int k = 0; k < 3; k++) { h.r[k] = p[k] - rb[sm.body].pos[k]; h.g[k] = g[k]; }
      }
    }
    // ordered compaction of the hits of this batch
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
        JointBody &B = sb[H.body];
        float v10[3];
        j_velocity_at(B, H.r, v10);
        const float v0 = j_dot(H.g, v10);
        const float den = j_impulse_contribution(B, H.r, H.g);
        const float J = -((1.0f + rest.e[H.body]) * v0) * (1.0f / den);
        if (!(J >= 0.0f)) continue;  // (J < 0: separating; a body of infinite mass and inertia gives 0 / 0)
        const float imp[3] = {J * H.g[0], J * H.g[1], J * H.g[2]};
        j_apply_impulse(B, imp, H.r);
        j_velocity_at(B, H.r, v10);
        const float vn = j_dot(H.g, v10);
        float tao[3] = {v10[0] - H.g[0] * vn, v10[1] - H.g[1] * vn, v10[2] - H.g[2] * vn};
        if (fmaxf(fabsf(tao[0]), fmaxf(fabsf(tao[1]), fabsf(tao[2]))) > 1e-7f) {
          const float il = 1.0f / sqrtf(j_dot(tao, tao));
          for (int k = 0; k < 3; k++) tao[k] *= il;
          const float fr = rb[H.body].fric[0];
          float jj = -j_dot(v10, tao) / j_impulse_contribution(B, H.r, tao);
          jj = fminf(fmaxf(jj, fr * -J), fr * J);
          const float fi[3] = {jj * tao[0], jj * tao[1], jj * tao[2]};
          j_apply_impulse(B, fi, H.r);
        }
      }
    }
    __syncthreads();
  }
  if (t >= 1 && t < nb) {
    for (int k = 0; k < 3; k++) { rb[t].vel[k] = sb[t].vel[k]; rb[t].omega[k] = sb[t].omega[k]; }
  }
}

__global__ void k_rigid_apply_tmp(RigidBodyDev *rb, int nb) {
  const int b = threadIdx.x;
  if (b < 1 || b >= nb) return;
  RigidBodyDev &B = rb[b];
  float t[3], u[3], w[3];
#pragma unroll
  for (int k = 0; k < 3; k++) B.vel[k] += B.tmp_imp[k] * B.inv_mass;
  // world inverse inertia applied to the torque: R (I^-1 (R^T torque))
#pragma unroll
  for (int c = 0; c < 3; c++) t[c] = B.R[c] * B.tmp_trq[0] + B.R[3 + c] * B.tmp_trq[1] + B.R[6 + c] * B.tmp_trq[2];
  rot_apply(B.inv_I, t, u);
  rot_apply(B.R, u, w);
#pragma unroll
  for (int k = 0; k < 3; k++) { B.omega[k] += w[k]; B.tmp_imp[k] = 0.0f; B.tmp_trq[k] = 0.0f; }
}

__device__ __forceinline__ void quat_to_R(const float q[4], float R[9]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
__device__ __forceinline__ void enforce_axis(RigidBodyDev &B) {  // enforce_angular_velocity_parallel_to (world-frame axis)
  const float am = fmaxf(fabsf(B.axis[0]), fmaxf(fabsf(B.axis[1]), fabsf(B.axis[2])));
  if (!(am > 0.1f)) return;
  const float il = 1.0f / sqrtf(B.axis[0] * B.axis[0] + B.axis[1] * B.axis[1] + B.axis[2] * B.axis[2]);
  const float a[3] = {B.axis[0] * il, B.axis[1] * il, B.axis[2] * il};
  const float d = a[0] * B.omega[0] + a[1] * B.omega[1] + a[2] * B.omega[2];
  B.omega[0] = a[0] * d; B.omega[1] = a[1] * d; B.omega[2] = a[2] * d;
}
// advect_rigid_bodies (src/mpm_rigid_body.cpp:255-286): RigidBody::advance (scripted: the pose of the script at t + dt and
// the secant velocity of the step; free: damping, explicit Euler, exponential map — the conventions of the rigid body
// the reference build uses, oracle/taichi_shim/taichi/dynamics/rigid_body_shim.h), then gravity as an impulse
__global__ void k_rigid_advect(RigidBodyDev *rb, int nb, RigidSteps steps, float dt, float g0, float g1, float g2) {
  const int b = threadIdx.x;
  if (b < 1 || b >= nb) return;
  RigidBodyDev &B = rb[b];
  const RigidStep &S = steps.s[b];
  enforce_axis(B);
  if (S.has_pos) {
#pragma unroll
    for (int k = 0; k < 3; k++) { B.vel[k] = (S.p1[k] - S.p0[k]) / dt; B.pos[k] = S.p1[k]; }
  } else {
    const float f = expf(-B.lin_damp * dt);
#pragma unroll
    for (int k = 0; k < 3; k++) { B.vel[k] *= f; B.pos[k] += B.vel[k] * dt; }
  }
  if (S.has_rot) {
    const float q0c[4] = {S.q0[0], -S.q0[1], -S.q0[2], -S.q0[3]};
    float dq[4];
    quat_mul(S.q1, q0c, dq);
    if (dq[0] < 0.0f) { dq[0] = -dq[0]; dq[1] = -dq[1]; dq[2] = -dq[2]; dq[3] = -dq[3]; }
    const float s = sqrtf(dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    const float ang = 2.0f * atan2f(s, dq[0]);
#pragma unroll
    for (int k = 0; k < 3; k++) B.omega[k] = s > 1e-12f ? dq[1 + k] * (ang / (s * dt)) : 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) B.q[k] = S.q1[k];
  } else {
    const float f = expf(-B.ang_damp * dt);
#pragma unroll
    for (int k = 0; k < 3; k++) B.omega[k] *= f;
    const float len = sqrtf(B.omega[0] * B.omega[0] + B.omega[1] * B.omega[1] + B.omega[2] * B.omega[2]);
    if (len * dt >= 1e-12f) {
      const float h = 0.5f * len * dt, sn = sinf(h) / len;
      const float d[4] = {cosf(h), sn * B.omega[0], sn * B.omega[1], sn * B.omega[2]};
      float o[4];
      quat_mul(d, B.q, o);
      const float nl = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
#pragma unroll
      for (int k = 0; k < 4; k++) B.q[k] = o[k] * nl;
    }
  }
  quat_to_R(B.q, B.R);
  // rigid.apply_impulse(gravity * mass * dt, rigid.position): no torque; a scripted translation has inv_mass = 0
  B.vel[0] += g0 * B.mass * dt * B.inv_mass; B.vel[1] += g1 * B.mass * dt * B.inv_mass; B.vel[2] += g2 * B.mass * dt * B.inv_mass;
  enforce_axis(B);
}

// world positions of the boundary particles (download / tests)
__global__ __launch_bounds__(256) void k_rigid_sample_positions(const RigidBodyDev *__restrict__ rb, const RigidSample *__restrict__ smp,
                                                                uint32_t n, float *__restrict__ out) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const RigidBodyDev &B = rb[smp[s].body];
    float w[3];
    rot_apply(B.R, smp[s].off, w);
    out[3 * s] = w[0] + B.pos[0]; out[3 * s + 1] = w[1] + B.pos[1]; out[3 * s + 2] = w[2] + B.pos[2];
  }
}
// dense views of the colored distance field (parity / download only): states word and distance of every node
__global__ __launch_bounds__(256) void k_cdf_dense(Params P, CdfDev C, uint32_t *__restrict__ states, float *__restrict__ dist) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < C.max_pages * 64u; t += gridDim.x * blockDim.x) {
    const uint32_t pg = t >> 6, l = t & 63u;
    if (pg % C.pool_cap >= min(C.n_pages[pg / C.pool_cap], C.pool_cap)) continue;  // not handed out this substep
    int bx, by, bz;
    demorton3(C.page_key[pg], bx, by, bz);
    if (C.slot[C.page_key[pg]] != pg) continue;
    const int gi = bx * 4 + (int)(l >> 4), gj = by * 4 + (int)((l >> 2) & 3), gk = bz * 4 + (int)(l & 3);
    if (gi > P.res[0] || gj > P.res[1] || gk > P.res[2]) continue;
    const size_t idx = ((size_t)gi * (P.res[1] + 1) + gj) * (P.res[2] + 1) + gk;
    const unsigned long long m = C.mind[t];
    states[idx] = (C.tags[t] & CDF_TAG_MASK) | (m != CDF_EMPTY ? ((uint32_t)(m & 0xFFu) << 24) : 0u);
    dist[idx] = m != CDF_EMPTY ? __uint_as_float((uint32_t)(m >> 32)) * P.dx : 0.0f;
  }
}

}  // namespace mpm
