// taichi_mpm_amd/csrc/k_grid.h — grid normalise + gravity + level-set boundary (+ halo sums, dense views, kinetic energy)
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

// ------------------------------------------------------------------------------------------------ grid
// The tile of active block a overlaps the 8 grid blocks c = block(a) + o, o in {0,1}^3 ("candidates").  A grid block c is
// processed by its "owner": the candidate with the smallest o among the active blocks c - o'.  The owner sums the
// overlapping tiles (<= 8), then
//   mode 0: normalize_grid_and_apply_external_force + apply_grid_boundary_conditions (src/mpm.cpp:277-372)
//           -> gridv[slot = 8a+o], fat_slot[morton(c)] = slot
//   mode 1: raw (m v, m) sums written to a dense node-major array (parity / download only)
//   mode 2: dense (v, m) array -> gridv (upload_grid)        mode 3: gridv -> dense (download_grid)
//   mode 4: grid kinetic energy (calculate_energy)
// Two kernels over the same arithmetic:
//   k_grid_list    (modes 0, 4) one wavefront per entry of the OWNER LIST the sort compacted (k_sort.h): the work items are exactly
//                  the touched grid blocks, the 27 neighbour slots of the owning block arrive as one coalesced 128-byte row, the halo
//                  boxes of a tiled ctx are tested per grid block with scalars.  Small problems and every tiled ctx (do_sort):
//                  17 -> 7.5 us at 1 M particles, 31 -> 19 us per rank at two bricks of C3 (profiles/r05_e_*_census.txt).
//   k_grid_blocks  the walks of rounds 1-4: one wavefront per active block walking its 8 candidates (PER_CAND = false), or one per
//                  (block, candidate); lanes 0..26 look the neighbours up in the bitmap, owner election by ballot.  Still the pass
//                  of an untiled ctx of 2 M slots and more — bound by its 180 MB of tile reads either way (30.4 against 30.5 us at
//                  8 M particles), without what the list costs the sort — and of the dense views (modes 1, 2, 3).
__device__ __forceinline__ constexpr int nb27(int dx, int dy, int dz) { return ((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1); }

// total of one halo node: the contributors' partial sums in RANK order (every holder of the node computes the bit-identical
// value).  `acc` = this rank's partial sum; (c0, c1, c2) = first node of the wave's grid block: a box that misses the block is
// skipped by a scalar test (adding its +0 would not change a bit: the running total is never -0), the others' values are
// fetched eight boxes at a time (independent loads, one round trip) and then added in order.
// (The box table is read with wave-uniform indices: scalar loads.  Handing it over as a kernel argument instead was built and
// dropped in round 5: the compiler keeps all eight boxes in SGPRs and spills 175 of them.)
// lower_has_mass (out): a contributor of LOWER rank holds mass on this node — the kinetic energy of a node is counted by the lowest
// rank that holds mass on it (mode 4), so that the ranks' shares add up to the one-ctx energy.
__device__ __forceinline__ float4 halo_total(const float4 acc, const int n_boxes, const int rank, const int gi, const int gj,
                                             const int gk, const int c0, const int c1, const int c2,
                                             const DevBox *__restrict__ boxes, bool &lower_has_mass) {
  float4 tot = make_float4(0, 0, 0, 0);
  bool own = false;
  lower_has_mass = false;
  for (int b0 = 0; b0 < n_boxes; b0 += 8) {
    float4 r[8];
    uint32_t hit = 0;  // wave-uniform
#pragma unroll
    for (int u = 0; u < 8; u++) {
      r[u] = make_float4(0, 0, 0, 0);
      if (b0 + u < n_boxes) {
        const DevBox &B = boxes[b0 + u];
        if (c0 + BS > B.lo[0] && c0 < B.lo[0] + B.dim[0] && c1 + BS > B.lo[1] && c1 < B.lo[1] + B.dim[1] && c2 + BS > B.lo[2] &&
            c2 < B.lo[2] + B.dim[2]) {
          hit |= 1u << u;
          const int x = gi - B.lo[0], y = gj - B.lo[1], z = gk - B.lo[2];
          if ((unsigned)x < (unsigned)B.dim[0] && (unsigned)y < (unsigned)B.dim[1] && (unsigned)z < (unsigned)B.dim[2])
            r[u] = B.recv[((size_t)x * B.dim[1] + y) * B.dim[2] + z];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (b0 + u < n_boxes) {
        if (!own && boxes[b0 + u].peer > rank) {
          tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w;
          own = true;
        }
        if ((hit >> u) & 1u) {
          tot.x += r[u].x; tot.y += r[u].y; tot.z += r[u].z; tot.w += r[u].w;
          if (!own && r[u].w != 0.0f) lower_has_mass = true;  // (boxes sorted by peer: before `own` = peers of lower rank)
        }
      }
    }
  }
  if (!own) { tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w; }
  return tot;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_grid_list(Params P, const Counters *__restrict__ cnt,
                                                   const uint32_t *__restrict__ nbr, const uint32_t *__restrict__ own_list,
                                                   const float4 *__restrict__ tiles, float4 *__restrict__ gridv,
                                                   uint32_t *__restrict__ fat_slot, double *__restrict__ energy, Tiling T,
                                                   const DevBox *__restrict__ boxes, LevelSetDev LS, int phase) {
  static_assert(MODE == 0 || MODE == 4, "the list walk serves the substep's pass and the energy");
  const int l = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  // (the first work item is requested before the number of work items is known: the list has 8 max_blocks entries)
  uint32_t first = 0;
  if (wave < P.max_blocks * 8u) first = own_list[wave];
  const int lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
  const uint32_t nwork = min(cnt->n_own, P.max_blocks * 8u);
  for (uint32_t work = wave; work < nwork; work += nwaves) {
    const uint32_t e = work == wave ? first : own_list[work];
    const uint32_t a = e >> 3;
    const int o = (int)(e & 7u), ox = o >> 2, oy = (o >> 1) & 1, oz = o & 1;
    // the owning block's row: lanes 0..26 its neighbours' slots (INVALID: not active), lane 27 its Morton key
    const uint32_t row = l < 32 ? nbr[(size_t)a * 32 + l] : INVALID;
    const uint32_t nslot = l < 27 ? row : INVALID;
    const uint32_t amask = (uint32_t)__ballot(nslot != INVALID);
    int bx, by, bz;
    demorton3(__shfl(row, 27), bx, by, bz);
    const int cx = bx + ox, cy = by + oy, cz = bz + oz;
    if (!in_phase(T, phase, cx * BS, cy * BS, cz * BS, BS)) continue;  // wave-uniform
    const uint32_t slot = e;  // = 8 a + o
    const int gi = cx * BS + lx, gj = cy * BS + ly, gk = cz * BS + lz;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) {  // sources of c = b + o are c - q = b + (o - q), q in {0,1}^3
      const int qx = q >> 2, qy = (q >> 1) & 1, qz = q & 1;
      const int nidx = nb27(ox - qx, oy - qy, oz - qz);
      const uint32_t sslot = __shfl(nslot, nidx);
      const int tx = lx + 4 * qx, ty = ly + 4 * qy, tz = lz + 4 * qz;
      if (((amask >> nidx) & 1u) && tx < TS && ty < TS && tz < TS) {
        const float4 t = tiles[(size_t)sslot * TN + (tx * TS + ty) * TS + tz];
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
    }
    bool counted_elsewhere = false;  // (mode 4, tiled: see halo_total)
    if (T.n_boxes > 0) {  // tiled: add the other ranks' partial sums, contributors in rank order
      // (decided per grid block, from scalars: the block lies inside the node box no halo box intersects)
      const int c0 = cx * BS, c1 = cy * BS, c2 = cz * BS;
      const bool interior = c0 >= T.int_lo[0] && c0 + BS <= T.int_hi[0] && c1 >= T.int_lo[1] && c1 + BS <= T.int_hi[1] &&
                            c2 >= T.int_lo[2] && c2 + BS <= T.int_hi[2];
      if (!interior) {
        const float own_mass = acc.w;
        bool lower;
        acc = halo_total(acc, T.n_boxes, T.rank, gi, gj, gk, c0, c1, c2, boxes, lower);
        counted_elsewhere = lower || own_mass == 0.0f;
      }
    }
    if (MODE == 4) {  // grid kinetic energy sum 1/2 m |v|^2 with v = (m v)/m (calculate_energy, src/mpm.cpp:1078-1096)
      double ke = (acc.w != 0.0f && !counted_elsewhere)
                      ? 0.5 * ((double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z) / acc.w : 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ke += __shfl_xor(ke, off);
      if (l == 0) atomicAdd(energy, ke);
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
      float phi, dphidt, nrm[3] = {0, 0, 0};
      levelset_eval(LS, P.t, xw, P.idx, phi, nrm, &dphidt);
      if (!(phi < -3.0f || 0.0f < phi)) {
        // boundary_velocity = -levelset.get_temporal_derivative(pos, t) * n * delta_x   (src/mpm.cpp:340-342)
        const float vb[3] = {-dphidt * nrm[0] * P.dx, -dphidt * nrm[1] * P.dx, -dphidt * nrm[2] * P.dx};
        friction_project(v, vb, nrm, LS.friction);
      }
    }
    if (LS.dirichlet && (float)gj * P.dx > 0.525f) { v[0] = 0.0f; v[1] = 0.0f; v[2] = 0.0f; }  // src/mpm.cpp:401-412 (behind the BC, :541-544)
    gridv[(size_t)slot * BC + l] = make_float4(v[0], v[1], v[2], m);
    if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
  }
}

// ---- the walks of rounds 1-4 (see above)
template <int MODE, bool PER_CAND>
__global__ __launch_bounds__(256) void k_grid_blocks(Params P, const Counters *__restrict__ cnt,
                                              const uint32_t *__restrict__ act_blk,
                                              const uint32_t *__restrict__ bits,
                                              const uint32_t *__restrict__ wprefix,
                                              const float4 *__restrict__ tiles, float4 *__restrict__ gridv,
                                              uint32_t *__restrict__ fat_slot, float4 *__restrict__ dense, Tiling T,
                                              const DevBox *__restrict__ boxes, LevelSetDev LS, int phase) {
  const uint32_t na_dev = cnt->n_active;
  const int l = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  // (the first work item's block key is requested before the number of active blocks is known: act_blk has max_blocks entries)
  const uint32_t a_first = PER_CAND ? wave >> 3 : wave;
  uint32_t key_first = 0;
  if (a_first < P.max_blocks) key_first = act_blk[a_first];
  const uint32_t na = min(na_dev, P.max_blocks);
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
    demorton3(work == wave ? key_first : act_blk[a], bx, by, bz);
    // neighbour table: lane n < 27 holds (active?, slot) of block b + (n/9-1, n/3%3-1, n%3-1)
    uint32_t nslot = INVALID;
    if (l < 27) {
      const int sx = bx + l / 9 - 1, sy = by + (l / 3) % 3 - 1, sz = bz + l % 3 - 1;
      if (sx >= 0 && sy >= 0 && sz >= 0) {
        const uint32_t bk = morton3(sx, sy, sz);
        if (block_active(bits, bk)) {
          // (a slot beyond max_blocks has no tile: the block table overflowed, the sticky capacity error is set)
          const uint32_t ns = block_slot(bits, wprefix, bk);
          if (ns < P.max_blocks) nslot = ns;
        }
      }
    }
    const uint32_t amask = (uint32_t)__ballot(nslot != INVALID);
    // One candidate: with a compile-time `o` (the unrolled loop below) all masks and neighbour indices are constants; with a
    // run-time `o` (PER_CAND: this wave's one candidate) the body exists ONCE — the eight-fold unrolled kernel is 13 000
    // instructions = 100 KB, more than the instruction cache holds, and a wave that handles a single candidate streams through
    // all of it (small problems: 28 -> 21 us at 1 M particles; at 8 M, where a wave walks all eight, the unrolled form wins:
    // 35 against 49 us — profiles/r03_q_ab_grid.txt).
    auto candidate = [&](const int o) __attribute__((always_inline)) {
      const int ox = o >> 2, oy = (o >> 1) & 1, oz = o & 1;
      // sources of c = b + o are c - q = b + (o - q), q in {0,1}^3; owner <=> none of them active for q < o
      uint32_t lower = 0;
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (q < o) lower |= 1u << nb27(ox - (q >> 2), oy - ((q >> 1) & 1), oz - (q & 1));
      if (amask & lower) return;  // wave-uniform
      const int cx = bx + ox, cy = by + oy, cz = bz + oz;
      if (!in_phase(T, phase, cx * BS, cy * BS, cz * BS, BS)) return;  // wave-uniform
      const uint32_t slot = a * 8u + (uint32_t)o;
      const int gi = cx * BS + lx, gj = cy * BS + ly, gk = cz * BS + lz;
      const bool in_grid = gi <= P.res[0] && gj <= P.res[1] && gk <= P.res[2];
      const size_t dense_idx = ((size_t)gi * (P.res[1] + 1) + gj) * (P.res[2] + 1) + gk;
      if (MODE == 2) {
        gridv[(size_t)slot * BC + l] = in_grid ? dense[dense_idx] : make_float4(0, 0, 0, 0);
        if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
        return;
      }
      if (MODE == 3) {
        if (in_grid) dense[dense_idx] = gridv[(size_t)slot * BC + l];
        return;
      }
      float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int qx = q >> 2, qy = (q >> 1) & 1, qz = q & 1;
        const int nidx = nb27(ox - qx, oy - qy, oz - qz);
        const uint32_t sslot = __shfl(nslot, nidx);
        const int tx = lx + 4 * qx, ty = ly + 4 * qy, tz = lz + 4 * qz;
        // (ablation builds only, results invalid: 5 of the 8 overlapping tiles read — the fold union tiles of 2 x 2 x 1 quads would
        // leave: experiments/p2g_quad_tiles/)
        if (MPM_ABLATE(P, 128) && q >= 5) continue;
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
        return;
      }
      if (MODE == 4) {  // grid kinetic energy sum 1/2 m |v|^2 with v = (m v)/m (calculate_energy, src/mpm.cpp:1078-1096)
        double e = (acc.w != 0.0f) ? 0.5 * ((double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z) / acc.w : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if (l == 0) atomicAdd(reinterpret_cast<double *>(dense), e);
        return;
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
        float phi, dphidt, nrm[3] = {0, 0, 0};
        levelset_eval(LS, P.t, xw, P.idx, phi, nrm, &dphidt);
        if (!(phi < -3.0f || 0.0f < phi)) {
          // boundary_velocity = -levelset.get_temporal_derivative(pos, t) * n * delta_x   (src/mpm.cpp:340-342)
          const float vb[3] = {-dphidt * nrm[0] * P.dx, -dphidt * nrm[1] * P.dx, -dphidt * nrm[2] * P.dx};
          friction_project(v, vb, nrm, LS.friction);
        }
      }
      if (LS.dirichlet && (float)gj * P.dx > 0.525f) { v[0] = 0.0f; v[1] = 0.0f; v[2] = 0.0f; }  // src/mpm.cpp:401-412 (behind the BC, :541-544)
      gridv[(size_t)slot * BC + l] = make_float4(v[0], v[1], v[2], m);
      if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
    };
    if constexpr (PER_CAND) {
      candidate(o_mine);
    } else {
#pragma unroll
      for (int o = 0; o < 8; o++) candidate(o);
    }
  }
}


}  // namespace mpm
