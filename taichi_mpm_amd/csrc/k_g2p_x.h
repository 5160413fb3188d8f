// taichi_mpm_amd/csrc/k_g2p_x.h — G2P with chunks that run ACROSS block boundaries
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
//
// k_g2p (k_g2p.h) walks chunks of 256 sorted positions INSIDE one active block: a block of 364 particles (the average after the
// benchmark's floor impact) is one full chunk and one with 108 busy lanes, and the time of the kernel follows the number of
// chunks, not of particles (lattice 35 k chunks 0.30 ms, after impact 42 k chunks 0.37 ms: 8.6 ns per chunk in both).  Here a
// chunk is 256 CONSECUTIVE sorted positions whatever blocks they belong to — always ceil(n / 256) chunks, every lane busy:
//   * a workgroup takes a contiguous range of chunks, so the 6^3 velocity tiles of the blocks it walks through stay in LDS:
//     a ring of NTILE tiles (slot = block index mod NTILE), a tile is loaded when a chunk first needs it;
//   * the sort hands over `chunk_blk[k]` = the active block holding sorted position 256 k (k_cell_table); the starts and keys
//     of that block and the next NTILE are workgroup-uniform (scalar loads), a lane finds its block by comparing its position
//     with the starts;
//   * a chunk that touches more than NTILE blocks (sparse spray: blocks of a few particles) is processed in rounds of NTILE
//     blocks, the lanes of later blocks keeping their records in registers meanwhile;
//   * the per-particle arithmetic is k_g2p's, textually (k_g2p_particle.inc), with the block origin and the tile per lane.
// Everything else — record prefetch two chunks ahead, transposed stores at the sorted positions into the other record set, keys
// and block flags for the next sort — is k_g2p's.  No rigid-body variant and no tiling phases: those take k_g2p.
#pragma once
#include "k_g2p.h"

namespace mpm {

constexpr int G2PX_NTILE = 4;

template <int MINW, bool STORE_B, uint32_t MATS = MAT_ALL>
__global__ __launch_bounds__(256, MINW) void k_g2p_x(Params P, const float4 *__restrict__ rg, float4 *__restrict__ rg_out,
                                                     float4 *__restrict__ rp_out, float4 *__restrict__ rb_out,
                                                     const Counters *__restrict__ cnt, const uint32_t *__restrict__ act_blk,
                                                     const uint32_t *__restrict__ act_start,
                                                     const uint32_t *__restrict__ chunk_blk,
                                                     const uint32_t *__restrict__ perm, const GroupParams *__restrict__ groups,
                                                     const float4 *__restrict__ gridv, const uint32_t *__restrict__ fat_slot,
                                                     Counters *cnt_w, uint32_t *__restrict__ key, uint8_t *__restrict__ blk_flag,
                                                     const LevelSetDev *__restrict__ ls) {
  constexpr int NT = 256;
  constexpr bool RIGID = false;
  __shared__ float4 tiles[G2PX_NTILE][TN];
  __shared__ GroupParams sgroups[G2P_LDS_GROUPS];
  __shared__ float4 xpose[NT / 64][64 * 5];
  __shared__ uint32_t xslot[NT / 64][64];
  for (int t = threadIdx.x; t < G2P_LDS_GROUPS * (int)(sizeof(GroupParams) / 4); t += NT)
    reinterpret_cast<uint32_t *>(sgroups)[t] = reinterpret_cast<const uint32_t *>(groups)[t];
  __syncthreads();
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t n = cnt->n_sorted;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  float4 *xp = xpose[wave];
  uint32_t *xs = xslot[wave];
  const float scale = -4.0f * P.idx * P.dt;  // :938
  const bool nt_store = P.n_slots >= NT_STORE_MIN_SLOTS;  // see st_rec
  const uint32_t nchunks = (n + NT - 1) / NT;
  const uint32_t per = (nchunks + gridDim.x - 1) / gridDim.x;
  const uint32_t k0 = min(blockIdx.x * per, nchunks), k1 = min(k0 + per, nchunks);
  static_assert(G2PX_NTILE == 4, "the slot bookkeeping below is written for four tiles");
  uint32_t res0 = INVALID, res1 = INVALID, res2 = INVALID, res3 = INVALID;  // which block each tile slot holds (workgroup-uniform)
  auto lane_slot = [&](uint32_t k) -> uint32_t {
    const uint32_t p = k * NT + tid;
    return (k < k1 && p < n) ? perm[p] : INVALID;
  };
  uint32_t i_cur = lane_slot(k0);
  float4 g0, g1, g2, g3;
  if (i_cur != INVALID) {
    const size_t i = i_cur;
    g0 = rg[i * 4 + 0]; g1 = rg[i * 4 + 1]; g2 = rg[i * 4 + 2]; g3 = rg[i * 4 + 3];
  }
  uint32_t i_nx = lane_slot(k0 + 1);
  float4 G0, G1, G2, G3, Q0, Q1, Q2, Q3, B0, B1, B2;
  G0 = G1 = G2 = G3 = Q0 = Q1 = Q2 = Q3 = B0 = B1 = B2 = make_float4(0, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prologue's loads are in (k_g2p.h: why)
  for (uint32_t k = k0; k < k1; k++) {
    // prefetch: records of the next chunk, index of the one after
    float4 n0, n1, n2, n3;
    if (i_nx != INVALID) {
      const size_t i = i_nx;
      n0 = rg[i * 4 + 0]; n1 = rg[i * 4 + 1]; n2 = rg[i * 4 + 2]; n3 = rg[i * 4 + 3];
    }
    const uint32_t i_nn = lane_slot(k + 2);
    struct { uint32_t p; } cur;  // (the shared particle arithmetic writes key[cur.p + tid])
    cur.p = k * NT;
    const uint32_t p_mine = cur.p + tid;
    bool pending = i_cur != INVALID;  // this lane's particle has not been processed yet
    uint32_t bkey = INVALID, out_slot = INVALID;
    uint32_t base = chunk_blk[k];  // (workgroup-uniform) the block holding the chunk's first position
    const uint32_t p_end = min(cur.p + NT, n);
    for (;;) {  // rounds of NTILE blocks: one round unless the chunk touches more than NTILE blocks
      // ---- this round's blocks (all of this is workgroup-uniform): where they start in the sorted index, which are needed
      uint32_t st[G2PX_NTILE + 1];
#pragma unroll
      for (int j = 0; j <= G2PX_NTILE; j++) st[j] = (base + j <= na) ? act_start[min(base + (uint32_t)j, na)] : 0xFFFFFFFFu;
      bool load[G2PX_NTILE];
      bool any_load = false;
#pragma unroll
      for (int j = 0; j < G2PX_NTILE; j++) {
        const uint32_t a = base + j, sl = a & 3u;
        const uint32_t held = sl == 0u ? res0 : (sl == 1u ? res1 : (sl == 2u ? res2 : res3));
        load[j] = a < na && st[j] < p_end && held != a;
        any_load = any_load || load[j];
      }
      if (any_load) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // everyone is done with the tiles of the previous chunk
#pragma unroll
        for (int j = 0; j < G2PX_NTILE; j++) {
          if (!load[j]) continue;
          const uint32_t a = base + j, sl = a & 3u;
          int bx, by, bz;
          demorton3(act_blk[a], bx, by, bz);
          float4 *tl = tiles[sl];
          for (int t = tid; t < TN; t += NT) {
            const int tx = t / (TS * TS), ty = (t / TS) % TS, tz = t % TS;
            const uint32_t fs = fat_slot[morton3(bx + (tx >> 2), by + (ty >> 2), bz + (tz >> 2))];
            tl[t] = gridv[(size_t)fs * BC + (((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3))];
          }
          if (sl == 0u) res0 = a; else if (sl == 1u) res1 = a; else if (sl == 2u) res2 = a; else res3 = a;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
      }
      // ---- this lane's block: base + (number of later block starts at or before its position)
      int jb = 0;
#pragma unroll
      for (int j = 1; j <= G2PX_NTILE; j++) jb += (int)(st[j] <= p_mine);
      const bool later = pending && jb >= G2PX_NTILE;  // a block of a later round
      const bool now = pending && !later;
      // its first cell (grid units) and its tile
      float ox = 0.0f, oy = 0.0f, oz = 0.0f;
#pragma unroll
      for (int j = 0; j < G2PX_NTILE; j++) {
        if (base + j >= na) break;  // uniform
        int bx, by, bz;
        demorton3(act_blk[base + j], bx, by, bz);  // (scalar: the address is uniform)
        if (jb == j) { ox = (float)(bx * BS); oy = (float)(by * BS); oz = (float)(bz * BS); }
      }
      const float4 *tile = tiles[(base + (uint32_t)min(jb, G2PX_NTILE - 1)) & 3u];
      auto particle = [&](const GroupParams &g) __attribute__((always_inline)) {
#include "k_g2p_particle.inc"
      };
      if (now) particle(sgroups[__float_as_uint(g3.y) & (G2P_LDS_GROUPS - 1)]);
      if (now) pending = false;
      if (!(st[G2PX_NTILE] < p_end)) break;  // (uniform) no block of the chunk lies behind this round's
      base += G2PX_NTILE;
    }
    // Let the prefetched records of the next chunk land BEFORE this chunk's stores go out (k_g2p.h)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    // transposed stores through this wave's LDS slab
    xs[lane] = out_slot;
    xp[lane * 5 + 0] = G0; xp[lane * 5 + 1] = G1; xp[lane * 5 + 2] = G2; xp[lane * 5 + 3] = G3;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
      const int src = 16 * q4 + (lane >> 2), q = lane & 3;
      const uint32_t sl = xs[src];
      const float4 val = xp[src * 5 + q];
      if (sl != INVALID) st_rec(rg_out + (size_t)sl * 4 + q, val, nt_store);
    }
    __builtin_amdgcn_wave_barrier();
    xp[lane * 5 + 0] = Q0; xp[lane * 5 + 1] = Q1; xp[lane * 5 + 2] = Q2; xp[lane * 5 + 3] = Q3;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
      const int src = 16 * q4 + (lane >> 2), q = lane & 3;
      const uint32_t sl = xs[src];
      const float4 val = xp[src * 5 + q];
      if (sl != INVALID) st_rec(rp_out + (size_t)sl * 4 + q, val, nt_store);
    }
    if constexpr (STORE_B) {
      __builtin_amdgcn_wave_barrier();
      xp[lane * 5 + 0] = B0; xp[lane * 5 + 1] = B1; xp[lane * 5 + 2] = B2;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q3 = 0; q3 < 3; q3++) {
        const int e = 64 * q3 + lane, src = e / 3, q = e - 3 * src;
        const uint32_t sl = xs[src];
        const float4 val = xp[src * 5 + q];
        if (sl != INVALID) rb_out[(size_t)sl * 3 + q] = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
    flag_block(blk_flag, bkey);
    i_cur = i_nx; i_nx = i_nn;
    g0 = n0; g1 = n1; g2 = n2; g3 = n3;
  }
  // slots behind the live range (particles deleted by earlier substeps have dropped out): dead for every consumer
  for (uint32_t t = n + blockIdx.x * NT + tid; t < P.n_slots; t += gridDim.x * NT) {
    key[t] = INVALID;
    rg_out[(size_t)t * 4 + 3] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);  // pid = -1
  }
}

}  // namespace mpm
