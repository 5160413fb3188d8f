// taichi_mpm_amd/csrc/k_rigid_transfer.h — P2G / G2P of the blocks near rigid bodies (CPIC colour test)
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
//
// The reference switches per block (block_op_switch, src/transfer.cpp:570-576): blocks whose page is in rigid_page_map
// run block_op_rigid (:367-463 P2G, :706-835 G2P), all others the SIMD block_op_normal.  Same split here: k_p2g /
// k_g2p skip the blocks flagged in blk_rigid (k_blk_rigid), the kernels below take exactly those.  They keep the data
// flow of the fast kernels (one wave per block and lane per cell for P2G; chunks of the sorted index, records written
// at their sorted positions into the other record set for G2P) but are written for clarity, not speed: rigid blocks
// are a thin shell around the bodies.
#pragma once
#include "k_rigid.h"

#ifndef MPM_RIGID_P2G_MINW
#define MPM_RIGID_P2G_MINW 2
#endif
#ifndef MPM_RIGID_G2P_MINW
#define MPM_RIGID_G2P_MINW 2
#endif

namespace mpm {

struct RigidXfer {
  CdfDev C;
  RigidBodyDev *rb;
  const BndRec *bnd;
  const uint8_t *blk_rigid;
  const uint32_t *rigid_list, *n_rigid;  // the flagged blocks as a list (k_blk_rigid)
  const float4 *rp_in;  // the current RecP set (G2P reads the particle's own velocity from it)
  float penalty, pushing_force;
};

// states of the block's 6^3 tile nodes (tags | body id + 1 << 24) into LDS
// (NT threads; a thread's nodes are looked up together — first every page slot, then every tag / distance word — so that a
// 64-thread workgroup pays two memory round trips for its four nodes per lane, not eight)
template <int NT>
__device__ __forceinline__ void load_state_tile(const CdfDev &C, int bx, int by, int bz, uint32_t *stile, int tid) {
  constexpr int R = (TN + NT - 1) / NT;
  uint32_t pg[R], cell[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int t = tid + r * NT;
    pg[r] = INVALID; cell[r] = 0u;
    if (t < TN) {
      const int i = bx * BS + t / (TS * TS), j = by * BS + (t / TS) % TS, k = bz * BS + t % TS;
      if (!(i < 0 || j < 0 || k < 0 || (i >> 2) >= C.nb_axis || (j >> 2) >= C.nb_axis || (k >> 2) >= C.nb_axis)) {
        pg[r] = C.slot[morton3(i >> 2, j >> 2, k >> 2)];
        cell[r] = (uint32_t)(((i & 3) << 4) | ((j & 3) << 2) | (k & 3));
      }
    }
  }
  uint32_t tg[R];
  unsigned long long md[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    tg[r] = 0u; md[r] = CDF_EMPTY;
    if (pg[r] != INVALID) {
      const size_t n = (size_t)pg[r] * 64 + cell[r];
      tg[r] = C.tags[n]; md[r] = C.mind[n];
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int t = tid + r * NT;
    if (t < TN) stile[t] = (tg[r] & CDF_TAG_MASK) | (md[r] != CDF_EMPTY ? ((uint32_t)(md[r] & 0xFFu) << 24) : 0u);  // = cdf_node_word
  }
}

// ------------------------------------------------------------------------------------------------ P2G, rigid blocks
// block_op_rigid of rasterize_optimized (src/transfer.cpp:367-463): a node of the other colour receives nothing; the
// particle's momentum change against the body's surface velocity (friction_project with the particle's boundary
// normal) and its stress term go to the body as an impulse at the node instead (:425-444).
constexpr int P2GR_LIST = 1024;
template <uint32_t MATS = MAT_ALL>  // material set of the ctx (mpm_math.h): the impulse walk evaluates calculate_force()
__global__ __launch_bounds__(64, MPM_RIGID_P2G_MINW) void k_p2g_rigid(Params P, const float4 *__restrict__ rp, const float4 *__restrict__ rg,
                                                  const Counters *__restrict__ cnt, const uint32_t *__restrict__ act_blk,
                                                  const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ perm,
                                                  const GroupParams *__restrict__ groups, float4 *__restrict__ tiles,
                                                  RigidXfer X) {
  __shared__ float4 tile[TN];
  __shared__ uint32_t stile[TN];
  __shared__ RigidLite srb[MAX_RIGID];
  __shared__ uint32_t blist[P2GR_LIST];  // the block's boundary particles: (position in the block's sorted range) << 6 | cell
  __shared__ uint32_t bcount;
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int lane = threadIdx.x;
  load_rigid_lite(srb, X.rb, lane, 64);  // (visible behind the first block's barrier)
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const int nbase = (cx * TS + cy) * TS + cz;
  const uint32_t nr = min(*X.n_rigid, na);
  for (uint32_t li = blockIdx.x; li < nr; li += gridDim.x) {
    const uint32_t a = X.rigid_list[li];
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    for (int t = lane; t < TN; t += 64) tile[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    load_state_tile<64>(X.C, bx, by, bz, stile, lane);
    if (lane == 0) bcount = 0u;
    __syncthreads();
    const int gx = bx * BS + cx, gy = by * BS + cy, gz = bz * BS + cz;  // base node of this lane's cell
    const uint32_t blk0 = __shfl(cell_start[a * BC + lane], 0);  // start of the block in the sorted index
    const uint32_t p0 = cell_start[a * BC + lane], p1 = cell_start[a * BC + lane + 1];
    // Two passes.  The first is k_p2g's scatter (lane = cell, 27 x 4 sums in registers) with the colour test; it also LISTS the
    // particles that have a node on the other side of a body.  The second, after those sums have been merged into the tile
    // and their registers are free, takes the list one LANE PER PARTICLE and hands the momentum change and the stress term of
    // the skipped nodes to the bodies — it evaluates calculate_force() (an eigen-solve), which in one loop with the 108
    // accumulators alive cost 43 spilled registers.  Boundary particles are a minority of a flagged block's particles
    // (walking them cell by cell cost as many rounds as the fullest cell had of them); their records come from L2 again.
    float acc[27][4];
#pragma unroll
    for (int n = 0; n < 27; n++) acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.0f;
    bool any_other = false;
    {
      float4 nq0, nq1, nq2, nq3, nh3;
      size_t inext = 0;
      if (p0 < p1) {
        inext = perm[p0];
        nq0 = rp[inext * 4 + 0]; nq1 = rp[inext * 4 + 1]; nq2 = rp[inext * 4 + 2]; nq3 = rp[inext * 4 + 3];
        nh3 = rg[inext * 4 + 3];
      }
      for (uint32_t p = p0; p < p1; p++) {
        const float4 q0 = nq0, q1 = nq1, q2 = nq2, q3 = nq3, h3 = nh3;
        if (p + 1 < p1) {  // the records of the next particle are in flight while one is computed
          inext = perm[p + 1];
          nq0 = rp[inext * 4 + 0]; nq1 = rp[inext * 4 + 1]; nq2 = rp[inext * 4 + 2]; nq3 = rp[inext * 4 + 3];
          nh3 = rg[inext * 4 + 3];
        }
        const uint32_t pstate = __float_as_uint(h3.w);
        const float mass = q3.w;
        float v[3] = {q0.w, q1.x, q1.y};
        if (P.particle_gravity) { v[0] = fmaf(P.g[0], P.dt, v[0]); v[1] = fmaf(P.g[1], P.dt, v[1]); v[2] = fmaf(P.g[2], P.dt, v[2]); }
        const float r0 = q0.x * P.idx - (float)gx, r1 = q0.y * P.idx - (float)gy, r2 = q0.z * P.idx - (float)gz;
        float w0[3], w1[3], w2[3];
        bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
        const float A00 = q1.z, A01 = q1.w, A02 = q2.x, A10 = q2.y, A11 = q2.z, A12 = q2.w, A20 = q3.x, A21 = q3.y, A22 = q3.z;
        const float mv0 = mass * v[0], mv1 = mass * v[1], mv2 = mass * v[2];
        // which of the 27 nodes belong to the other side of a body for this particle; the ordinary scatter skips them
        uint32_t other = 0u;
#pragma unroll
        for (int n = 0; n < 27; n++) {
          const int i3 = n / 9, j = (n / 3) % 3, k = n % 3;
          if (cdf_incompatible(stile[nbase + (i3 * TS + j) * TS + k], pstate)) other |= 1u << n;
        }
        any_other = any_other || other != 0u;
        if (other != 0u) {
          const uint32_t slot = atomicAdd(&bcount, 1u);
          if (slot < (uint32_t)P2GR_LIST) blist[slot] = ((p - blk0) << 6) | (uint32_t)lane;
        }
        // k_p2g's scatter (k_p2g.h: the contribution stepped from node to node by the columns of the affine matrix, on packed
        // fp32 pairs, the mass riding along) with the weight of the skipped nodes set to zero
        const f2 a0xy = {A00, A10}, a0zw = {A20, 0.0f}, a1xy = {A01, A11}, a1zw = {A21, 0.0f}, a2xy = {A02, A12}, a2zw = {A22, 0.0f};
        f2 cixy = {fmaf(A02, r2, fmaf(A01, r1, fmaf(A00, r0, mv0))), fmaf(A12, r2, fmaf(A11, r1, fmaf(A10, r0, mv1)))};
        f2 cizw = {fmaf(A22, r2, fmaf(A21, r1, fmaf(A20, r0, mv2))), mass};
#pragma unroll
        for (int i3 = 0; i3 < 3; i3++) {
          f2 cjxy = cixy, cjzw = cizw;
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const float wij = w0[i3] * w1[j];
            f2 ckxy = cjxy, ckzw = cjzw;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const int n = (i3 * 3 + j) * 3 + k;
              const f2 w = splat2(((other >> n) & 1u) ? 0.0f : wij * w2[k]);
              f2 axy = {acc[n][0], acc[n][1]}, azw = {acc[n][2], acc[n][3]};
              axy = fma2(w, ckxy, axy); azw = fma2(w, ckzw, azw);
              acc[n][0] = axy.x; acc[n][1] = axy.y; acc[n][2] = azw.x; acc[n][3] = azw.y;
              if (k < 2) { ckxy -= a2xy; ckzw -= a2zw; }
            }
            if (j < 2) { cjxy -= a1xy; cjzw -= a1zw; }
          }
          if (i3 < 2) { cixy -= a0xy; cizw -= a0zw; }
        }
      }
    }
    // ordered, race-free merges into the wave's tile (see k_p2g.h)
#pragma unroll
    for (int n = 0; n < 27; n++) {
      const int node = nbase + ((n / 9) * TS + (n / 3) % 3) * TS + n % 3;
      if (p1 > p0) {
        float4 t = tile[node];
        t.x += acc[n][0]; t.y += acc[n][1]; t.z += acc[n][2]; t.w += acc[n][3];
        tile[node] = t;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    }
    // second pass: the listed particles, one per lane (if the list overflowed: cell by cell, every particle re-tested)
    ImpulseAcc ia;
    acc_init(ia);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const uint32_t nlist = bcount;
    const bool listed = nlist <= (uint32_t)P2GR_LIST;
    uint32_t lt = (uint32_t)lane, pc = p0;
    while (true) {
      uint32_t p = 0u, cell = (uint32_t)lane;
      bool have;
      if (listed) {
        have = lt < nlist;
        if (have) { const uint32_t e = blist[lt]; p = blk0 + (e >> 6); cell = e & 63u; }
        lt += 64u;
      } else {
        have = any_other && pc < p1;
        p = pc++;
      }
      if (!__any(have)) break;
      if (!have) continue;
      const int ccx = (int)(cell >> 4), ccy = (int)((cell >> 2) & 3u), ccz = (int)(cell & 3u);
      const int pbase = (ccx * TS + ccy) * TS + ccz;
      const int px = bx * BS + ccx, py = by * BS + ccy, pz = bz * BS + ccz;  // base node of the particle's cell
      const size_t icur = perm[p];
      const float4 h3 = rg[icur * 4 + 3];
      const uint32_t pstate = __float_as_uint(h3.w);
      uint32_t other = 0u;
#pragma unroll
      for (int n = 0; n < 27; n++) {
        const int i3 = n / 9, j = (n / 3) % 3, k = n % 3;
        if (cdf_incompatible(stile[pbase + (i3 * TS + j) * TS + k], pstate)) other |= 1u << n;
      }
      if (!other) continue;
      const float4 q0 = rp[icur * 4 + 0], q1 = rp[icur * 4 + 1], q3 = rp[icur * 4 + 3];
      const float4 nb0 = reinterpret_cast<const float4 *>(X.bnd)[icur * 2];
      const float bnn[3] = {nb0.x, nb0.y, nb0.z};
      const float mass = q3.w;
      float v[3] = {q0.w, q1.x, q1.y};
      if (P.particle_gravity) { v[0] = fmaf(P.g[0], P.dt, v[0]); v[1] = fmaf(P.g[1], P.dt, v[1]); v[2] = fmaf(P.g[2], P.dt, v[2]); }
      const float r0 = q0.x * P.idx - (float)px, r1 = q0.y * P.idx - (float)py, r2 = q0.z * P.idx - (float)pz;
      float w0[3], w1[3], w2[3];
      bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
      // d/dx of the quadratic B-spline in grid units (src/kernel.h:131-132): dw = (1, -2, 1) t + (-1.5, 0, 1.5)
      const float t0[3] = {r0, r0 - 1.0f, r0 - 2.0f}, t1[3] = {r1, r1 - 1.0f, r1 - 2.0f}, t2[3] = {r2, r2 - 1.0f, r2 - 2.0f};
      const float dw0[3] = {t0[0] - 1.5f, -2.0f * t0[1], t0[2] + 1.5f}, dw1[3] = {t1[0] - 1.5f, -2.0f * t1[1], t1[2] + 1.5f},
                  dw2[3] = {t2[0] - 1.5f, -2.0f * t2[1], t2[2] + 1.5f};
      const float4 h0 = rg[icur * 4 + 0], h1 = rg[icur * 4 + 1], h2 = rg[icur * 4 + 2];
      mat3 F;
      F.m[0] = h1.x; F.m[1] = h1.y; F.m[2] = h1.z; F.m[3] = h1.w; F.m[4] = h2.x; F.m[5] = h2.y; F.m[6] = h2.z; F.m[7] = h2.w; F.m[8] = h3.x;
      mat3 dtF = calculate_force<MATS>(groups[__float_as_uint(h3.y)], F, h0.w);  // delta_t * calculate_force()
#pragma unroll
      for (int e = 0; e < 9; e++) dtF.m[e] *= P.dt;
      while (other) {
        const int n = __ffs(other) - 1;
        other &= other - 1u;
        const int i3 = n / 9, j = (n / 3) % 3, k = n % 3;
        const int rid = (int)(stile[pbase + (i3 * TS + j) * TS + k] >> 24) - 1;
        if (rid < 0) continue;
        const RigidLite &B = srb[rid];
        // weights and their derivatives of node (i3, j, k) without indexing the per-axis arrays dynamically
        const float wa = i3 == 0 ? w0[0] : (i3 == 1 ? w0[1] : w0[2]), wb = j == 0 ? w1[0] : (j == 1 ? w1[1] : w1[2]),
                    wc = k == 0 ? w2[0] : (k == 1 ? w2[1] : w2[2]);
        const float da = i3 == 0 ? dw0[0] : (i3 == 1 ? dw0[1] : dw0[2]), db = j == 0 ? dw1[0] : (j == 1 ? dw1[1] : dw1[2]),
                    dc = k == 0 ? dw2[0] : (k == 1 ? dw2[1] : dw2[2]);
        const float w = (wa * wb) * wc;
        const float gp[3] = {(px + i3) * P.dx, (py + j) * P.dx, (pz + k) * P.dx};
        float rv[3];
        rigid_velocity_at(B, gp, rv);
        float pv[3] = {v[0], v[1], v[2]};
        friction_project(pv, rv, bnn, B.fric[(pstate >> (2 * rid)) & 1u]);
        const float gr[3] = {da * P.idx * wb * wc, wa * db * P.idx * wc, wa * wb * dc * P.idx};
        float imp[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
          imp[c] = mass * w * (v[c] - pv[c]) + (dtF(c, 0) * gr[0] + dtF(c, 1) * gr[1] + dtF(c, 2) * gr[2]);
        acc_add(ia, X.rb, rid, imp, gp, B.pos);
      }
    }
    acc_flush_wave(ia, X.rb);  // the wave's impulses: six atomics per body
    __syncthreads();
    for (int t = lane; t < TN; t += 64) tiles[(size_t)a * TN + t] = tile[t];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ G2P, rigid blocks
// block_op_rigid of resample_optimized (src/transfer.cpp:706-835): a node of the other colour contributes the
// particle's own velocity, or — when gather_cdf found a boundary for the particle — that velocity projected onto the
// body's surface motion plus a push along the boundary normal (:757-784); a particle near a boundary loses its affine
// momentum (:800-804) and is pushed back by the penalty term when it is slightly inside (:821-832), the body receiving
// the opposite impulse.  Everything after the gather is k_g2p's (same record layout, same key / deletion logic).
template <uint32_t MATS = MAT_ALL>
__global__ __launch_bounds__(256, MPM_RIGID_G2P_MINW) void k_g2p_rigid(Params P, const float4 *__restrict__ rg, float4 *__restrict__ rg_out,
                                                   float4 *__restrict__ rp_out, float4 *__restrict__ rb_out,
                                                   const Counters *__restrict__ cnt, const uint32_t *__restrict__ act_blk,
                                                   const uint32_t *__restrict__ act_start, const uint32_t *__restrict__ perm,
                                                   const GroupParams *__restrict__ groups, const float4 *__restrict__ gridv,
                                                   const uint32_t *__restrict__ fat_slot, Counters *cnt_w,
                                                   uint32_t *__restrict__ key, uint8_t *__restrict__ blk_flag,
                                                   const LevelSetDev *__restrict__ ls, RigidXfer X) {
  __shared__ float4 tile[TN];
  __shared__ uint32_t stile[TN];
  __shared__ RigidLite srb[MAX_RIGID];
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int tid = threadIdx.x;
  load_rigid_lite(srb, X.rb, tid, 256);  // (visible behind the first block's barriers)
  const float scale = -4.0f * P.idx * P.dt;
  const uint32_t nr = min(*X.n_rigid, na);
  for (uint32_t li = blockIdx.x; li < nr; li += gridDim.x) {
    const uint32_t a = X.rigid_list[li];
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    __syncthreads();
    for (int t = tid; t < TN; t += 256) {
      const int tx = t / (TS * TS), ty = (t / TS) % TS, tz = t % TS;
      const uint32_t fs = fat_slot[morton3(bx + (tx >> 2), by + (ty >> 2), bz + (tz >> 2))];
      tile[t] = gridv[(size_t)fs * BC + (((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3))];
    }
    load_state_tile<256>(X.C, bx, by, bz, stile, tid);
    __syncthreads();
    const float ox = (float)(bx * BS), oy = (float)(by * BS), oz = (float)(bz * BS);
    const uint32_t q0 = act_start[a] & ~ACT_RIGID_BIT, q1 = act_start[a + 1] & ~ACT_RIGID_BIT;
    for (uint32_t pb = q0; pb < q1; pb += 256) {  // uniform trip count: every lane reaches flag_block
      const uint32_t p = pb + tid;
      uint32_t bkey = INVALID;
      ImpulseAcc ia;
      acc_init(ia);
      if (p < q1) {
        const size_t i = perm[p];
        const float4 g0 = rg[i * 4 + 0], g1 = rg[i * 4 + 1], g2 = rg[i * 4 + 2], g3 = rg[i * 4 + 3];
        const GroupParams &g = groups[__float_as_uint(g3.y)];
        const uint32_t pstate = __float_as_uint(g3.w);
        const BndRec bn = X.bnd[i];
        // the particle's velocity before this G2P lives in its RecP record; the reference reads p.get_velocity() here,
        // i.e. the velocity AFTER rasterize added gravity to it (particle_gravity, src/transfer.cpp:393-395)
        const float4 pq0 = X.rp_in[i * 4 + 0], pq1 = X.rp_in[i * 4 + 1];
        float pv[3] = {pq0.w, pq1.x, pq1.y};
        if (P.particle_gravity) { pv[0] = fmaf(P.g[0], P.dt, pv[0]); pv[1] = fmaf(P.g[1], P.dt, pv[1]); pv[2] = fmaf(P.g[2], P.dt, pv[2]); }
        const float x0 = g0.x, x1 = g0.y, x2 = g0.z;
        const float X0 = x0 * P.idx - ox, X1 = x1 * P.idx - oy, X2 = x2 * P.idx - oz;
        const int c0 = min(max((int)(X0 - 0.5f), 0), BS - 1), c1 = min(max((int)(X1 - 0.5f), 0), BS - 1),
                  c2 = min(max((int)(X2 - 0.5f), 0), BS - 1);  // (clamped: see k_p2g_rigid)
        const float r0 = X0 - (float)c0, r1 = X1 - (float)c1, r2 = X2 - (float)c2;
        float w0[3], w1[3], w2[3];
        bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
        float v[3] = {0, 0, 0};
        mat3 b;
#pragma unroll
        for (int e = 0; e < 9; e++) b.m[e] = 0.0f;
        int rigid_id = -1;
        const int nbase = (c0 * TS + c1) * TS + c2;
        // pass 1: the nodes on the other side of a body; pass 2: the ordinary gather without them (small unrolled body);
        // pass 3 (particles at a boundary only): their share, with the projected velocity the reference substitutes
        uint32_t other = 0u;
#pragma unroll
        for (int n = 0; n < 27; n++)
          if (cdf_incompatible(stile[nbase + ((n / 9) * TS + (n / 3) % 3) * TS + n % 3], pstate)) other |= 1u << n;
#pragma unroll
        for (int n = 0; n < 27; n++) {
          const int i3 = n / 9, j = (n / 3) % 3, k = n % 3;
          const float4 gv4 = tile[nbase + (i3 * TS + j) * TS + k];
          const float w = ((other >> n) & 1u) ? 0.0f : (w0[i3] * w1[j]) * w2[k];
          const float d[3] = {r0 - (float)i3, r1 - (float)j, r2 - (float)k}, gv[3] = {gv4.x, gv4.y, gv4.z};
#pragma unroll
          for (int r = 0; r < 3; r++) {
            v[r] = fmaf(w, gv[r], v[r]);
#pragma unroll
            for (int c = 0; c < 3; c++) b(r, c) = fmaf(w * gv[r], d[c], b(r, c));
          }
        }
        while (other) {
          const int n = __ffs(other) - 1;
          other &= other - 1u;
          const int i3 = n / 9, j = (n / 3) % 3, k = n % 3;
          const uint32_t word = stile[nbase + (i3 * TS + j) * TS + k];
          float fake[3] = {pv[0], pv[1], pv[2]};
          const int rid = (int)(word >> 24) - 1;
          float vg[3] = {0, 0, 0}, friction = 0.0f;
          if (rid >= 0) {
            const float gp[3] = {(bx * BS + c0 + i3) * P.dx, (by * BS + c1 + j) * P.dx, (bz * BS + c2 + k) * P.dx};
            rigid_velocity_at(srb[rid], gp, vg);
            rigid_id = rid;
            friction = srb[rid].fric[(pstate >> (2 * rid)) & 1u];
          }
          if (bn.near) {
            friction_project(fake, vg, bn.n, friction);
            const float push = P.dt * P.dx * X.pushing_force;
            fake[0] += bn.n[0] * push; fake[1] += bn.n[1] * push; fake[2] += bn.n[2] * push;
          }
          const float wa = i3 == 0 ? w0[0] : (i3 == 1 ? w0[1] : w0[2]), wb = j == 0 ? w1[0] : (j == 1 ? w1[1] : w1[2]),
                      wc = k == 0 ? w2[0] : (k == 1 ? w2[1] : w2[2]);
          const float w = (wa * wb) * wc;
          const float d[3] = {r0 - (float)i3, r1 - (float)j, r2 - (float)k};
#pragma unroll
          for (int r = 0; r < 3; r++) {
            v[r] = fmaf(w, fake[r], v[r]);
#pragma unroll
            for (int c = 0; c < 3; c++) b(r, c) = fmaf(w * fake[r], d[c], b(r, c));
          }
        }
        mat3 cdg;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) cdg(r, c) = fmaf(scale, b(r, c), (r == c) ? 1.0f : 0.0f);
        if (bn.near) {
#pragma unroll
          for (int e = 0; e < 9; e++) b.m[e] = 0.0f;  // p.apic_b = Matrix(0), :800-801
        } else if (P.rpic_damping != 0.0f || P.apic_damping != 0.0f) {
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
        plasticity_and_force<MATS>(g, cdg, F, aux, stress);
        float nx0 = fmaf(v[0], P.dt, x0), nx1 = fmaf(v[1], P.dt, x1), nx2 = fmaf(v[2], P.dt, x2);
        if (bn.near && bn.dist < -0.05f * P.dx && bn.dist > -P.dx * 0.3f) {  // :821-832
          const float dv[3] = {bn.dist * bn.n[0] * X.penalty, bn.dist * bn.n[1] * X.penalty, bn.dist * bn.n[2] * X.penalty};
          v[0] -= dv[0]; v[1] -= dv[1]; v[2] -= dv[2];
          if (rigid_id != -1) {
            const float imp[3] = {dv[0] * g.p[0], dv[1] * g.p[0], dv[2] * g.p[0]}, at[3] = {nx0, nx1, nx2};
            acc_add(ia, X.rb, rigid_id, imp, at, srb[rigid_id].pos);
          }
        }
        if (P.clamp_pos) {
          nx0 = fminf(fmaxf(nx0 * P.idx, 0.0f), (float)P.res[0] - 1e-6f) * P.dx;
          nx1 = fminf(fmaxf(nx1 * P.idx, 0.0f), (float)P.res[1] - 1e-6f) * P.dx;
          nx2 = fminf(fmaxf(nx2 * P.idx, 0.0f), (float)P.res[2] - 1e-6f) * P.dx;
        }
        if (P.particle_collision) {
          const LevelSetDev &LS = *ls;
          const float xw[3] = {nx0, nx1, nx2};
          float phi, gr[3] = {0, 0, 0};
          if (levelset_eval(LS, P.t, xw, P.idx, phi, gr) && phi < 0.0f) {
            const float vn = gr[0] * v[0] + gr[1] * v[1] + gr[2] * v[2];
            nx0 -= gr[0] * phi * P.dx; nx1 -= gr[1] * phi * P.dx; nx2 -= gr[2] * phi * P.dx;
            v[0] -= vn * gr[0]; v[1] -= vn * gr[1]; v[2] -= vn * gr[2];
          }
        }
        const float m4 = 4.0f * g.p[0];
        float A[9];
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = fmaf(stress.m[e], scale, b.m[e] * m4);
        const float nxp[3] = {nx0, nx1, nx2};
        const uint32_t kk = particle_key(P, nxp, v, bkey);
        int32_t pid = __float_as_int(g3.z);
        if (kk == INVALID) {
          pid = -1;
          atomicAdd(&cnt_w->n_dead, 1u);
        }
        key[p] = kk;
        if (P.pidc) P.pidc[p] = (uint32_t)pid;
        const size_t o = p;
        rg_out[o * 4 + 0] = make_float4(nx0, nx1, nx2, aux);
        rg_out[o * 4 + 1] = make_float4(F.m[0], F.m[1], F.m[2], F.m[3]);
        rg_out[o * 4 + 2] = make_float4(F.m[4], F.m[5], F.m[6], F.m[7]);
        rg_out[o * 4 + 3] = make_float4(F.m[8], g3.y, __int_as_float(pid), g3.w);
        rp_out[o * 4 + 0] = make_float4(nx0, nx1, nx2, v[0]);
        rp_out[o * 4 + 1] = make_float4(v[1], v[2], A[0], A[1]);
        rp_out[o * 4 + 2] = make_float4(A[2], A[3], A[4], A[5]);
        rp_out[o * 4 + 3] = make_float4(A[6], A[7], A[8], g.p[0]);
        if (P.store_b) {
          rb_out[o * 3 + 0] = make_float4(b.m[0], b.m[1], b.m[2], b.m[3]);
          rb_out[o * 3 + 1] = make_float4(b.m[4], b.m[5], b.m[6], b.m[7]);
          rb_out[o * 3 + 2] = make_float4(b.m[8], 0.0f, 0.0f, 0.0f);
        }
      }
      acc_flush_wave(ia, X.rb);
      flag_block(blk_flag, bkey);
    }
  }
}

}  // namespace mpm
