// taichi_mpm_amd/csrc/k_g2p.h — G2P (resample_optimized, src/transfer.cpp:837-954) with the fused constitutive update
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

#ifndef MPM_FENCE_A
#define MPM_FENCE_A
#endif
#ifndef MPM_FENCE_B
#define MPM_FENCE_B
#endif
namespace mpm {

// ------------------------------------------------------------------------------------------------ G2P
// resample_optimized / block_op_normal (src/transfer.cpp:837-954), one workgroup per active block, one
// particle per lane through the sorted index.  Also produces, for the NEXT substep: the affine matrix A of
// P2G (stress of the updated F from the same eigen-solve as the plasticity) and the sort key of the new position.
// The updated records are written to the OTHER record buffers at the particle's SORTED POSITION (the host swaps the
// buffers behind the launch): every substep leaves the records in the order of its own sort, i.e. one substep stale —
// the physical reorder of the reference's sort_allocator (src/mpm.cpp:752-768) for free, every substep instead of every
// reorder_interval, with fully sequential record stores; particles deleted by an earlier substep drop out, so the live
// records always occupy the slots [0, n_sorted).
constexpr int G2P_LDS_GROUPS = MPMHIP_MAX_GROUPS;  // the ctx's group capacity: the whole table is mirrored in LDS (5 KiB)
// RIGID: CPIC rigid bodies exist — blocks flagged by k_blk_rigid (top bit of act_start) are left to k_g2p_rigid and the
// records' spare word (the particle's colour) is carried along.  A compile-time switch: the instantiation without it is the kernel tuned above,
// instruction for instruction.
// MATS: the set of material types the ctx's groups use (mpm_math.h: plasticity_and_force) — the full set, or one material.
template <int NT, int MINW, bool ROLL, bool STORE_B, bool RIGID = false, uint32_t MATS = MAT_ALL>
__global__ __launch_bounds__(NT, MINW) void k_g2p(Params P, const float4 *__restrict__ rg, float4 *__restrict__ rg_out,
                                                  float4 *__restrict__ rp_out, float4 *__restrict__ rb_out,
                                                  const Counters *__restrict__ cnt,
                                                  const uint32_t *__restrict__ act_blk,
                                                  const uint32_t *__restrict__ act_start,
                                                  const uint32_t *__restrict__ perm,
                                                  const GroupParams *__restrict__ groups,
                                                  const float4 *__restrict__ gridv,
                                                  const uint32_t *__restrict__ fat_slot, Counters *cnt_w,
                                                  uint32_t *__restrict__ key, uint8_t *__restrict__ blk_flag,
                                                  const LevelSetDev *__restrict__ ls, PhaseBox T, int phase) {
  __shared__ float4 tile[TN];
  __shared__ GroupParams sgroups[G2P_LDS_GROUPS];
  for (int t = threadIdx.x; t < G2P_LDS_GROUPS * (int)(sizeof(GroupParams) / 4); t += NT)
    reinterpret_cast<uint32_t *>(sgroups)[t] = reinterpret_cast<const uint32_t *>(groups)[t];  // (the table holds >= 256 rows)
  __syncthreads();
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
      bool rigid_block = false;
      if constexpr (RIGID) {  // the top bit flags a block near a rigid body (k_blk_rigid): k_g2p_rigid takes it
        rigid_block = (c.p & 0x80000000u) != 0u;
        c.p &= 0x7FFFFFFFu; c.p1 &= 0x7FFFFFFFu;
      }
      bool mine = c.p < c.p1 && !rigid_block;  // (empty block: all its particles migrated away)
      if (mine && phase != 0) {
        int bx, by, bz;
        demorton3(act_blk[c.a], bx, by, bz);
        mine = in_phase(T, phase, bx * BS, by * BS, bz * BS, 2 * BS);
      }
      if (mine) break;
      c.a += gridDim.x;
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
  const bool nt_store = P.n_slots >= NT_STORE_MIN_SLOTS;  // see st_rec
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
  // the outgoing records of a lane (lanes without a particle leave stale values here: they are never stored)
  float4 G0, G1, G2, G3, Q0, Q1, Q2, Q3, B0, B1, B2;
  G0 = G1 = G2 = G3 = Q0 = Q1 = Q2 = Q3 = B0 = B1 = B2 = make_float4(0, 0, 0, 0);
  // The loop below keeps loads in flight across its back edge, and the compiler's wait-count insertion merges the state of
  // the entry edge with that of the back edge: a prologue load still pending at entry (the index i_nx) makes it put a
  // vmcnt(0) in front of the prefetch INSIDE the loop — where it also waits for the previous chunk's record stores (the
  // counter is shared and in-order): measured 0.305 -> 0.364 ms at C3 when an unrelated edit changed the schedule.  Draining
  // the prologue's loads here, once per workgroup, makes the entry state empty whatever the schedule.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  while (cur.a < na) {
    if (cur.a != tile_a) {
      // LDS-only barriers (lgkmcnt(0) + s_barrier): __syncthreads() would also wait on vmcnt, i.e. on the
      // acknowledgements of the previous chunk's record stores
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();  // everyone is done with the previous tile
      int bx, by, bz;
      demorton3(act_blk[cur.a], bx, by, bz);
      for (int t = tid; t < TN; t += NT) {
        const int tx = t / (TS * TS), ty = (t / TS) % TS, tz = t % TS;
        const int qx = tx >> 2, qy = ty >> 2, qz = tz >> 2;
        const uint32_t fs = fat_slot[morton3(bx + qx, by + qy, bz + qz)];
        tile[t] = gridv[(size_t)fs * BC + (((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3))];
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
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
    auto particle = [&](const GroupParams &g) __attribute__((always_inline)) {
#include "k_g2p_particle.inc"
    };
    // Group parameters are read at use (keeps ~20 VGPRs free) from the workgroup's LDS copy of the table: DS reads
    // wait on lgkmcnt, whereas vector loads in the middle of the arithmetic wait on vmcnt and with it on the
    // prefetched records of the next chunk (the counter is in-order), which would undo the prefetch.
    if (i_cur != INVALID) particle(sgroups[__float_as_uint(g3.y) & (G2P_LDS_GROUPS - 1)]);
    // Let the prefetched records of the next chunk land BEFORE this chunk's stores go out: on gfx9-family parts
    // loads and stores share one in-order-per-type counter (vmcnt), so a later wait for those loads would be a
    // vmcnt(0) that also waits for the stores' acknowledgements — here the loads have had the whole arithmetic to
    // arrive, and the stores then drain behind the next chunk's arithmetic.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    // transposed stores through this wave's LDS slab (DS operations of one wave execute in program order)
    xs[lane] = out_slot;
    xp[lane * 5 + 0] = G0; xp[lane * 5 + 1] = G1; xp[lane * 5 + 2] = G2; xp[lane * 5 + 3] = G3;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int src = 16 * k + (lane >> 2), q = lane & 3;
      const uint32_t sl = xs[src];
      const float4 val = xp[src * 5 + q];
      if (sl != INVALID) st_rec(rg_out + (size_t)sl * 4 + q, val, nt_store);
    }
    __builtin_amdgcn_wave_barrier();
    xp[lane * 5 + 0] = Q0; xp[lane * 5 + 1] = Q1; xp[lane * 5 + 2] = Q2; xp[lane * 5 + 3] = Q3;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int src = 16 * k + (lane >> 2), q = lane & 3;
      const uint32_t sl = xs[src];
      const float4 val = xp[src * 5 + q];
      if (sl != INVALID) st_rec(rp_out + (size_t)sl * 4 + q, val, nt_store);
    }
    if constexpr (STORE_B) {  // (compile-time: in the default folded mode the apic_b registers do not exist)
      __builtin_amdgcn_wave_barrier();
      xp[lane * 5 + 0] = B0; xp[lane * 5 + 1] = B1; xp[lane * 5 + 2] = B2;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int e = 64 * k + lane, src = e / 3, q = e - 3 * src;
        const uint32_t sl = xs[src];
        const float4 val = xp[src * 5 + q];
        if (sl != INVALID) rb_out[(size_t)sl * 3 + q] = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
    flag_block(blk_flag, bkey);
    cur = nx; nx = nn;
    i_cur = i_nx; i_nx = i_nn;
    g0 = n0; g1 = n1; g2 = n2; g3 = n3;
  }
  // slots behind the live range (particles deleted by earlier substeps have dropped out): dead for every consumer
  for (uint32_t t = cnt->n_sorted + blockIdx.x * NT + tid; t < P.n_slots; t += gridDim.x * NT) {
    key[t] = INVALID;
    rg_out[(size_t)t * 4 + 3] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);  // pid = -1
  }
}


}  // namespace mpm
