// taichi_mpm_amd/csrc/k_g2p_packed.h — G2P over PACKED chunks: the same particle arithmetic as k_g2p (g2p_particle, k_g2p.h), another walk
// over the particles.  Part of libmpmhip.
//
// k_g2p walks chunks of 256 sorted positions INSIDE one block, so a block's last chunk is as full as the block's size allows: on
// the lattice the reference's benchmark seeds (512 particles per block) every chunk is full, 400 substeps after the impact a
// block holds 348 particles on average — 256 + 92: the lanes are 68 % used and there are 1.47 times as many chunks.  Here a chunk
// is 256 CONSECUTIVE positions of the sorted index, whatever blocks they belong to: every chunk is full, every workgroup's share
// of the work is equal.  A chunk can touch several blocks; the workgroup keeps K block tiles in LDS (slot = block index mod K,
// tagged with the block they hold, so a tile survives from chunk to chunk while positions stay in its block), a lane finds its
// block among the chunk's at most K by comparing its position with their starts (k_cell_table wrote, per chunk, the block that
// holds the chunk's first position), and reads its tile and the tile's origin from its slot.  A chunk that touches more than K
// blocks (K consecutive blocks with fewer than 256 particles between them: spray) is done in several passes.
// A workgroup takes runs of G2P_RUN consecutive chunks (tiles are reused along a run), the runs dealt round-robin over the launch, so
// that the workgroups resident together work on neighbouring blocks (their tiles share grid blocks in the L2).  The host launches
// four rounds of the device's resident set and picks this walk when the blocks are NOT full (mpmhip.hip: g2p_is_packed).
#pragma once
#include "k_g2p.h"

namespace mpm {

constexpr int G2P_PK = 4;  // block tiles resident per workgroup
// consecutive chunks a workgroup takes at a time (a compile-time power of two: the division sits in the chunk walk).  Measured at C3,
// lattice / after impact, against k_g2p's 283 / 363 us on the same box: 2 -> +5 / -15, 4 -> +-0 / -16; 3 and 8 (run-time value) worse
constexpr uint32_t G2P_RUN = 4;

template <int NT, int MINW, bool STORE_B, uint32_t MATS>
__global__ __launch_bounds__(NT, MINW) void k_g2p_packed(Params P, const float4 *__restrict__ rg, float4 *__restrict__ rg_out,
                                                         float4 *__restrict__ rp_out, float4 *__restrict__ rb_out,
                                                         const Counters *__restrict__ cnt, const uint32_t *__restrict__ act_blk,
                                                         const uint32_t *__restrict__ act_start, const uint32_t *__restrict__ perm,
                                                         const GroupParams *__restrict__ groups, const float4 *__restrict__ gridv,
                                                         const uint32_t *__restrict__ fat_slot, Counters *cnt_w,
                                                         uint32_t *__restrict__ key, uint8_t *__restrict__ blk_flag,
                                                         const LevelSetDev *__restrict__ ls, const uint32_t *__restrict__ chunk_blk
#ifdef MPMHIP_TIMING_BUILD  // (variant library only, profiles/g2p_tile_time.py: wall-clock time a workgroup spends making tiles resident)
                                                         , unsigned long long *__restrict__ tlog
#endif
                                                         ) {
  static_assert(NT == 256, "chunk_blk is written for 256-position chunks");
#ifdef MPMHIP_TIMING_BUILD
  const unsigned long long t_enter = wall_clock64();
  unsigned long long t_tile = 0, n_tile = 0, n_chunk = 0;
#endif
  __shared__ float4 tile[G2P_PK * TN];
  __shared__ float s_org[G2P_PK][4];    // origin (grid node of the tile's node (0, 0, 0)) of the block in each slot
  __shared__ GroupParams sgroups[G2P_LDS_GROUPS];
  for (int t = threadIdx.x; t < G2P_LDS_GROUPS * (int)(sizeof(GroupParams) / 4); t += NT)
    reinterpret_cast<uint32_t *>(sgroups)[t] = reinterpret_cast<const uint32_t *>(groups)[t];
  __syncthreads();
  uint32_t tag[G2P_PK];  // the block whose tile sits in each slot (INVALID: none) — the same in every thread: no LDS, no barrier to read it
#pragma unroll
  for (int s = 0; s < G2P_PK; s++) tag[s] = INVALID;
  __shared__ float4 xpose[NT / 64][64 * 5];  // store staging, one slab per wavefront (see k_g2p)
  __shared__ uint32_t xslot[NT / 64][64];
  const uint32_t na = min(cnt->n_active, P.max_blocks), n_sorted = cnt->n_sorted;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  float4 *xp = xpose[wave];
  uint32_t *xs = xslot[wave];
  const float scale = -4.0f * P.idx * P.dt;  // :938
  // the k-th chunk of this workgroup: runs of G2P_RUN consecutive chunks (a tile serves a run), the runs dealt round-robin — the
  // workgroups that are resident together then work on neighbouring blocks, whose tiles share grid blocks in the L2
  const uint32_t nchunks = (n_sorted + NT - 1) / NT;
  auto chunk_of = [&](uint32_t k) -> uint32_t { return ((k / G2P_RUN) * gridDim.x + blockIdx.x) * G2P_RUN + k % G2P_RUN; };
  auto lane_slot = [&](uint32_t k) -> uint32_t {
    const uint32_t c = chunk_of(k), p = c * NT + tid;
    return (c < nchunks && p < n_sorted) ? perm[p] : INVALID;
  };
  const bool nt_store = P.n_slots >= NT_STORE_MIN_SLOTS;  // see st_rec
  uint32_t k_cur = 0;
  uint32_t i_cur = lane_slot(k_cur);
  float4 g0, g1, g2, g3;
  if (i_cur != INVALID) {
    const size_t i = i_cur;
    g0 = rg[i * 4 + 0]; g1 = rg[i * 4 + 1]; g2 = rg[i * 4 + 2]; g3 = rg[i * 4 + 3];
  }
  uint32_t i_nx = lane_slot(k_cur + 1);
  // chunk metadata (uniform): the block holding the chunk's first position and the starts of the G2P_PK + 1 blocks from there —
  // two dependent (scalar) loads, requested while the chunk before is computed
  struct Meta { uint32_t ab; uint32_t st[G2P_PK + 1]; };
  auto load_meta = [&](uint32_t k) {
    const uint32_t c = chunk_of(k);
    Meta m;
    m.ab = 0;
#pragma unroll
    for (int s = 0; s <= G2P_PK; s++) m.st[s] = 0;
    if (c < nchunks) {
      m.ab = chunk_blk[c];
#pragma unroll
      for (int s = 0; s <= G2P_PK; s++) m.st[s] = act_start[min(m.ab + (uint32_t)s, na)];
    }
    return m;
  };
  Meta m_cur = load_meta(k_cur);
  float4 G0, G1, G2, G3, Q0, Q1, Q2, Q3, B0, B1, B2;
  G0 = G1 = G2 = G3 = Q0 = Q1 = Q2 = Q3 = B0 = B1 = B2 = make_float4(0, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered with nothing pending (see k_g2p)
  while (chunk_of(k_cur) < nchunks) {  // (chunk_of grows with k: the first chunk behind the end ends the walk)
    const uint32_t p0 = chunk_of(k_cur) * NT, p1 = min(p0 + NT, n_sorted), pos = p0 + tid;
    float4 n0, n1, n2, n3;
    uint32_t i_nn = INVALID;
    Meta m_nx;
    uint32_t ab = m_cur.ab;        // the block that holds p0 (uniform)
    uint32_t st[G2P_PK + 1];       // starts of the window's blocks (uniform; behind the last block: the live count)
#pragma unroll
    for (int s = 0; s <= G2P_PK; s++) st[s] = m_cur.st[s];
    bool done = i_cur == INVALID;  // (lanes behind the live range have nothing to do)
    for (bool first_pass = true;; first_pass = false) {  // passes over the chunk's blocks, G2P_PK at a time — nearly always one
      if (!first_pass) {
#pragma unroll
        for (int s = 0; s <= G2P_PK; s++) st[s] = act_start[min(ab + (uint32_t)s, na)];
      }
      // make the tiles of the window's blocks that overlap the chunk resident
      uint32_t need = 0;
#pragma unroll
      for (int s = 0; s < G2P_PK; s++) {
        const uint32_t a = ab + (uint32_t)s;
        if (a < na && st[s] < p1 && st[s + 1] > st[s] && tag[a % G2P_PK] != a) need |= 1u << s;
      }
      if (need) {  // uniform
#ifdef MPMHIP_TIMING_BUILD
        const unsigned long long t0 = wall_clock64();
#endif
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // everyone is done with the tiles these replace
#pragma unroll
        for (int s = 0; s < G2P_PK; s++) {
          if (!((need >> s) & 1u)) continue;
          const uint32_t a = ab + (uint32_t)s, slot = a % G2P_PK;
          int bx, by, bz;
          demorton3(act_blk[a], bx, by, bz);
          for (int t = tid; t < TN; t += NT) {
            const int tx = t / (TS * TS), ty = (t / TS) % TS, tz = t % TS;
            const uint32_t fs = fat_slot[morton3(bx + (tx >> 2), by + (ty >> 2), bz + (tz >> 2))];
            tile[slot * TN + t] = gridv[(size_t)fs * BC + (((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3))];
          }
          if (tid == 0) { s_org[slot][0] = (float)(bx * BS); s_org[slot][1] = (float)(by * BS); s_org[slot][2] = (float)(bz * BS); }
#pragma unroll
          for (int q = 0; q < G2P_PK; q++) if ((uint32_t)q == slot) tag[q] = a;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
#ifdef MPMHIP_TIMING_BUILD
        t_tile += wall_clock64() - t0; n_tile += __popc(need);
#endif
      }
      if (first_pass) {
        // prefetch: records and metadata of the next chunk, index of the one after — BEHIND the tile loads above: the vector-memory
        // counter is in order, a wait for the tile would otherwise wait for these as well
        if (i_nx != INVALID) {
          const size_t i = i_nx;
          n0 = rg[i * 4 + 0]; n1 = rg[i * 4 + 1]; n2 = rg[i * 4 + 2]; n3 = rg[i * 4 + 3];
        }
        i_nn = lane_slot(k_cur + 2);
        m_nx = load_meta(k_cur + 1);
      }
      // this lane's block: the last one of the window that starts at or before its position
      uint32_t bkey = INVALID, out_slot = INVALID;
      const bool in_window = !done && pos >= st[0] && pos < st[G2P_PK];
      if (in_window) {
        uint32_t sl = 0;
#pragma unroll
        for (int s = 1; s < G2P_PK; s++) sl += st[s] <= pos ? 1u : 0u;
        const uint32_t slot = (ab + sl) % G2P_PK;
        g2p_particle<MATS, STORE_B, false>(P, scale, tile + slot * TN, s_org[slot][0], s_org[slot][1], s_org[slot][2], g0, g1, g2, g3,
                                           sgroups[__float_as_uint(g3.y) & (G2P_LDS_GROUPS - 1)], ls, cnt_w, key, pos, bkey, out_slot, G0,
                                           G1, G2, G3, Q0, Q1, Q2, Q3, B0, B1, B2, xp + lane * 5);
        done = true;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prefetched records land before this chunk's stores go out (see k_g2p)
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
      if constexpr (STORE_B) {
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
      if (P.pidc && out_slot != INVALID) P.pidc[out_slot] = __float_as_uint(G3.z);  // (deterministic mode: the id beside the key, see k_g2p)
      flag_block(blk_flag, bkey);
      if (st[G2P_PK] >= p1 || ab + G2P_PK >= na) break;  // uniform: the window reached the end of the chunk
      ab += G2P_PK;
    }
#ifdef MPMHIP_TIMING_BUILD
    n_chunk++;
#endif
    k_cur++;
    m_cur = m_nx;
    i_cur = i_nx; i_nx = i_nn;
    g0 = n0; g1 = n1; g2 = n2; g3 = n3;
  }
#ifdef MPMHIP_TIMING_BUILD
  if (tlog && tid == 0) {
    tlog[4 * (size_t)blockIdx.x] = wall_clock64() - t_enter; tlog[4 * (size_t)blockIdx.x + 1] = t_tile;
    tlog[4 * (size_t)blockIdx.x + 2] = n_tile; tlog[4 * (size_t)blockIdx.x + 3] = n_chunk;
  }
#endif
  // slots behind the live range (particles deleted by earlier substeps have dropped out): dead for every consumer
  for (uint32_t t = n_sorted + blockIdx.x * NT + tid; t < P.n_slots; t += gridDim.x * NT) {
    key[t] = INVALID;
    rg_out[(size_t)t * 4 + 3] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);  // pid = -1
  }
}

}  // namespace mpm
