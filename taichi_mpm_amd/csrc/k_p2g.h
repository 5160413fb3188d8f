// taichi_mpm_amd/csrc/k_p2g.h — P2G (rasterize_optimized, src/transfer.cpp:467-569)
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

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
// MC: merge chains.  The merge below is 27 read-modify-writes of the wave's tile that must stay in program order (two offsets of two
// lanes can name the same node): 27 x (LDS read latency + add + write) = 1.3 us of a block's ~15.  With MC = 3 every x-plane of the
// stencil merges into its OWN tile (tile + c * TN): three independent chains of nine steps whose reads, adds and writes interleave;
// the write-out sums the three tiles.
template <int N0, int N1, int MC, class AfterParticles>
__device__ __forceinline__ void p2g_cell(const Params &P, const float4 *__restrict__ rp,
                                         const uint32_t *__restrict__ perm,
                                         const GroupParams *__restrict__ groups, uint32_t p0, uint32_t p1, uint32_t i0,
                                         uint32_t i1, float ox, float oy, float oz, int nbase, float4 *tile,
                                         AfterParticles &&after_particles) {
  constexpr int NN = N1 - N0;
  float acc[NN][4];
#pragma unroll
  for (int n = 0; n < NN; n++) { acc[n][0] = 0.0f; acc[n][1] = 0.0f; acc[n][2] = 0.0f; acc[n][3] = 0.0f; }
  // software pipeline: the records of the next TWO particles and the index of the third are in flight while
  // one particle is computed (one particle's arithmetic is shorter than the loaded HBM latency; a third record in
  // flight was measured: no change)
  float4 n0, n1, n2, n3, m0, m1, m2, m3;
  uint32_t inext = 0;
  if (p0 < p1) {  // (i0 = perm[p0], i1 = perm[p0 + 1]: loaded by the caller while the previous block was merged)
    const size_t i = i0;
    n0 = rp[i * 4 + 0]; n1 = rp[i * 4 + 1]; n2 = rp[i * 4 + 2]; n3 = rp[i * 4 + 3];
    if (p0 + 1 < p1) {
      const size_t j = i1;
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
    if constexpr (N0 == 0 && N1 == 27) {
      // contrib(i, j, k) = affine (r - (i, j, k)) + mass v is affine in the node offset: start from the node (0, 0, 0)
      // and step by one column of the affine matrix per node — (x, y) and (z, m) as packed fp32 pairs, the mass riding
      // along with a zero step: 2 packed adds + 2 packed multiply-adds per node instead of 9 + 4 scalar ones (256 -> 212
      // vector instructions per particle; the kernel is latency-bound, so only 0.171 -> 0.168 ms at C3, 0.206 -> 0.201 after
      // impact).  Same terms as :535-541, the offsets subtracted column by column instead of before the product.
      const f2 a0xy = {A00, A10}, a0zw = {A20, 0.0f}, a1xy = {A01, A11}, a1zw = {A21, 0.0f}, a2xy = {A02, A12},
               a2zw = {A22, 0.0f};
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
            const f2 w = splat2(wij * w2[k]);
            f2 axy = {acc[n][0], acc[n][1]}, azw = {acc[n][2], acc[n][3]};
            axy = fma2(w, ckxy, axy); azw = fma2(w, ckzw, azw);
            acc[n][0] = axy.x; acc[n][1] = axy.y; acc[n][2] = azw.x; acc[n][3] = azw.y;
            if (k < 2) { ckxy -= a2xy; ckzw -= a2zw; }
          }
          if (j < 2) { cjxy -= a1xy; cjzw -= a1zw; }
        }
        if (i3 < 2) { cixy -= a0xy; cizw -= a0zw; }
      }
      continue;
    }
#pragma unroll
    for (int n = N0; n < N1; n++) {  // (node-split variants: every node from scratch)
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
  after_particles();  // (the caller's look-ahead loads for the next block: in flight during the merge and the write-out)
  // Merge the per-cell sums into this wave's tile.  The tile belongs to this wavefront alone, and within one
  // stencil-offset step all 64 lanes address distinct nodes (same offset, different cells), so a plain float4
  // read-modify-write is race-free as long as the steps stay in program order: LDS operations of one wave
  // execute in order, the wave_barrier keeps the compiler from interleaving them.  (DS float atomics cost
  // ~2 LDS cycles per LANE on gfx950 even without conflicts: measured 145 cycles per ds_add_f32.)
  if constexpr (MC == 3 && N0 == 0 && N1 == 27) {
#pragma unroll
    for (int s = 0; s < 9; s++) {
      float4 t[3];
#pragma unroll
      for (int c = 0; c < 3; c++) t[c] = tile[c * TN + nbase + (c * TS + s / 3) * TS + s % 3];
      if (p1 > p0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          t[c].x += acc[c * 9 + s][0]; t[c].y += acc[c * 9 + s][1]; t[c].z += acc[c * 9 + s][2]; t[c].w += acc[c * 9 + s][3];
          tile[c * TN + nbase + (c * TS + s / 3) * TS + s % 3] = t[c];
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    }
    return;
  }
#pragma unroll
  for (int n = N0; n < N1; n++) {
    const int node = nbase + ((n / 9) * TS + (n / 3) % 3) * TS + n % 3;
    if (MPM_ABLATE(P, 8) && acc[n - N0][3] != 1.2345e-30f) continue;
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
template <int NS, int PS, int MINW, bool RIGID = false>  // RIGID: skip the blocks flagged in blk_rigid (k_p2g_rigid takes them)
__global__ __launch_bounds__(64 * NS * PS, MINW) void k_p2g(Params P, const float4 *__restrict__ rp,
                                                            const Counters *__restrict__ cnt,
                                                            const uint32_t *__restrict__ act_blk,
                                                            const uint32_t *__restrict__ cell_start,
                                                            const uint32_t *__restrict__ perm,
                                                            const GroupParams *__restrict__ groups,
                                                            float4 *__restrict__ tiles, Tiling T, int phase,
                                                            const uint8_t *__restrict__ blk_rigid
#ifdef MPMHIP_TIMING_BUILD  // (variant library lib/libmpmhip_timing.so only, profiles/p2g_block_times.py: per-block wall-clock stamps)
                                                            , unsigned long long *__restrict__ tlog
#endif
                                                            ) {
  constexpr int NW = NS * PS, NT = 64 * NW;
  constexpr int MC = (NS == 1 && PS == 1) ? 3 : 1;  // merge chains (p2g_cell): three in the one-wave form; measured -2..-4 us of 166 at C3
  __shared__ float4 tile[NW * MC][TN];  // per wave (and merge chain): (m*vx, m*vy, m*vz, m) per node of the block's 6^3 tile
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npart = wave % NS, ppart = wave / NS;
  const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
  const int nbase = (cx * TS + cy) * TS + cz;
  // The start of a block is a chain of dependent loads (block key, cell table, first indices, first records).  The
  // first three links are taken one block ahead: key and cell range of the NEXT block are requested before this block's
  // particle loop, its first two indices before this block's tile is merged and written out — a block then starts with
  // its record loads.
  struct Ahead { uint32_t key, c0, c1, p0, p1, i0, i1; };
  auto range_of = [&](Ahead &h) {
    const uint32_t n = h.c1 - h.c0;
    h.p0 = h.c0 + (n * ppart + PS - 1) / PS;
    h.p1 = h.c0 + (n * (ppart + 1) + PS - 1) / PS;
    // (ablation builds only, results invalid: at most 8 / 6 particles per cell — the upper bound of what ANY balancing of the cells of a
    // block could give k_p2g after impact, where a wave takes as long as its fullest cell: profiles/r05_m_sort_front_and_p2g_cap.txt)
    if (MPM_ABLATE(P, 16)) h.p1 = min(h.p1, h.p0 + 8u);
    if (MPM_ABLATE(P, 32)) h.p1 = min(h.p1, h.p0 + 6u);
  };
  auto load_table = [&](uint32_t a, Ahead &h) {
    h.key = 0; h.c0 = 0; h.c1 = 0;
    if (a < na) { h.key = act_blk[a]; h.c0 = cell_start[a * BC + lane]; h.c1 = cell_start[a * BC + lane + 1]; }
  };
  auto load_indices = [&](Ahead &h) {
    range_of(h);
    h.i0 = 0; h.i1 = 0;
    if (h.p0 < h.p1) h.i0 = perm[h.p0];
    if (h.p0 + 1 < h.p1) h.i1 = perm[h.p0 + 1];
  };
  Ahead cur, nxt;
  load_table(blockIdx.x, cur);
  load_indices(cur);
  for (uint32_t a = blockIdx.x; a < na; a += gridDim.x) {
    load_table(a + gridDim.x, nxt);
    int bx, by, bz;
    demorton3(cur.key, bx, by, bz);
    bool mine = in_phase(T, phase, bx * BS, by * BS, bz * BS, TS);  // workgroup-uniform
    if constexpr (RIGID) { if (blk_rigid[a]) mine = false; }  // near a rigid body: k_p2g_rigid takes the block (CPIC colour test)
    if (!mine) {
      load_indices(nxt);
      cur = nxt;
      continue;
    }
#ifdef MPMHIP_TIMING_BUILD
    const unsigned long long t_begin = wall_clock64();
#endif
    for (int t = threadIdx.x; t < NW * MC * TN; t += NT) (&tile[0][0])[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    const float ox = (float)(bx * BS + cx), oy = (float)(by * BS + cy), oz = (float)(bz * BS + cz);
    auto ahead = [&]() { load_indices(nxt); };
    if constexpr (NS == 1) {
      p2g_cell<0, 27, MC>(P, rp, perm, groups, cur.p0, cur.p1, cur.i0, cur.i1, ox, oy, oz, nbase, tile[wave * MC], ahead);
    } else {
      if (npart == 0) p2g_cell<0, 14, 1>(P, rp, perm, groups, cur.p0, cur.p1, cur.i0, cur.i1, ox, oy, oz, nbase, tile[wave], ahead);
      else p2g_cell<14, 27, 1>(P, rp, perm, groups, cur.p0, cur.p1, cur.i0, cur.i1, ox, oy, oz, nbase, tile[wave], ahead);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TN; t += NT) {
      // (ablation builds only, results invalid: 150 of the 216 nodes written — what a 10 x 10 x 6 union tile per 2 x 2 x 1 quad of blocks
      // would write per block, 600 / 4: the upper bound of experiments/p2g_quad_tiles/)
      if (MPM_ABLATE(P, 64) && t >= 150) continue;
      float4 u = tile[0][t];
#pragma unroll
      for (int w = 1; w < NW * MC; w++) {
        const float4 q = tile[w][t];
        u.x += q.x; u.y += q.y; u.z += q.z; u.w += q.w;
      }
      tiles[(size_t)a * TN + t] = u;
    }
    __syncthreads();
#ifdef MPMHIP_TIMING_BUILD
    {
      unsigned cmax = cur.c1 - cur.c0, csum = cmax;  // the fullest cell of the block and its particle count
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, off)); csum += (unsigned)__shfl_xor((int)csum, off); }
      if (tlog && threadIdx.x == 0) {
        tlog[3 * (size_t)a] = t_begin; tlog[3 * (size_t)a + 1] = wall_clock64();
        tlog[3 * (size_t)a + 2] = ((unsigned long long)cmax << 32) | csum;
      }
    }
#endif
    cur = nxt;
  }
}

}  // namespace mpm
