// taichi_mpm_amd/csrc/k_g2p.h — G2P (resample_optimized, src/transfer.cpp:837-954) with the fused constitutive update
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

// One particle of G2P: the 27-tap gather from the LDS tile `tile` (whose node (0, 0, 0) is the grid node (ox, oy, oz)), the
// constitutive update, the advection, the next substep's P2G matrix and sort key, and the outgoing records.  `pos` = the particle's
// position in the sorted index (where its records go).  Shared by k_g2p and k_g2p_packed (k_g2p_packed.h).
template <uint32_t MATS, bool STORE_B, bool RIGID>
__device__ __forceinline__ void g2p_particle(const Params &P, const float scale, const float4 *tile, const float ox, const float oy,
                                             const float oz, const float4 g0, const float4 g1, const float4 g2, const float4 g3,
                                             const GroupParams &g, const LevelSetDev *__restrict__ ls, Counters *cnt_w,
                                             uint32_t *__restrict__ key, const uint32_t pos, uint32_t &bkey, uint32_t &out_slot,
                                             float4 &G0, float4 &G1, float4 &G2, float4 &G3, float4 &Q0, float4 &Q1, float4 &Q2,
                                             float4 &Q3, float4 &B0, float4 &B1, float4 &B2, float4 *lane_lds) {
    // (lane_lds: five float4 of LDS private to this lane — its row of the wave's store-staging slab, idle until the records
    // are staged: scratch of the rare refinement of an ill-conditioned F, mpm_math.h: sym_eig3_refine)
    const float x0 = g0.x, x1 = g0.y, x2 = g0.z;
    const float X0 = x0 * P.idx - ox, X1 = x1 * P.idx - oy, X2 = x2 * P.idx - oz;
    const int c0 = (int)(X0 - 0.5f), c1 = (int)(X1 - 0.5f), c2 = (int)(X2 - 0.5f);
    const float r0 = X0 - (float)c0, r1 = X1 - (float)c1, r2 = X2 - (float)c2;
    float w0[3], w1[3], w2[3];
    bspline_weights(r0, w0); bspline_weights(r1, w1); bspline_weights(r2, w2);
    // 27-tap gather, :888-904:  v = sum w g,  b[:, c] = sum (w d_c) g  with  w = w0[i] w1[j] w2[k],  d = r - (i, j, k).
    // The weights factor along the axes, so the sums are taken axis by axis — 4 multiply-adds per node on the (x, y) and
    // (z, m) register pairs of the tile's float4 (packed fp32: v_pk_fma_f32) instead of 15 scalar ones per node:
    //   S0 = sum_k w2[k] g,  S1 = sum_k (w2 d2)[k] g;   T0 = sum_j w1[j] S0,  T1 = sum_j (w1 d1)[j] S0,  T2 = sum_j w1[j] S1;
    //   v = sum_i w0[i] T0,  b[:,0] = sum_i (w0 d0)[i] T0,  b[:,1] = sum_i w0[i] T1,  b[:,2] = sum_i w0[i] T2
    // (the m lanes ride along unused).  Same terms as the reference's loop, summed in a different order.
    const float e0[3] = {w0[0] * r0, w0[1] * (r0 - 1.0f), w0[2] * (r0 - 2.0f)};
    const float e1[3] = {w1[0] * r1, w1[1] * (r1 - 1.0f), w1[2] * (r1 - 2.0f)};
    const float e2[3] = {w2[0] * r2, w2[1] * (r2 - 1.0f), w2[2] * (r2 - 2.0f)};
    const f2 z2 = {0.0f, 0.0f};
    f2 vxy = z2, vzw = z2, b0xy = z2, b0zw = z2, b1xy = z2, b1zw = z2, b2xy = z2, b2zw = z2;
    const int nbase = (c0 * TS + c1) * TS + c2;
    auto plane = [&](int i3, float w0i, float e0i) __attribute__((always_inline)) {
      f2 T0xy = z2, T0zw = z2, T1xy = z2, T1zw = z2, T2xy = z2, T2zw = z2;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        f2 S0xy = z2, S0zw = z2, S1xy = z2, S1zw = z2;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float4 gv = tile[nbase + (i3 * TS + j) * TS + k];
          const f2 gxy = {gv.x, gv.y}, gzw = {gv.z, gv.w};
          S0xy = fma2(splat2(w2[k]), gxy, S0xy); S0zw = fma2(splat2(w2[k]), gzw, S0zw);
          S1xy = fma2(splat2(e2[k]), gxy, S1xy); S1zw = fma2(splat2(e2[k]), gzw, S1zw);
        }
        T0xy = fma2(splat2(w1[j]), S0xy, T0xy); T0zw = fma2(splat2(w1[j]), S0zw, T0zw);
        T1xy = fma2(splat2(e1[j]), S0xy, T1xy); T1zw = fma2(splat2(e1[j]), S0zw, T1zw);
        T2xy = fma2(splat2(w1[j]), S1xy, T2xy); T2zw = fma2(splat2(w1[j]), S1zw, T2zw);
      }
      vxy = fma2(splat2(w0i), T0xy, vxy); vzw = fma2(splat2(w0i), T0zw, vzw);
      b0xy = fma2(splat2(e0i), T0xy, b0xy); b0zw = fma2(splat2(e0i), T0zw, b0zw);
      b1xy = fma2(splat2(w0i), T1xy, b1xy); b1zw = fma2(splat2(w0i), T1zw, b1zw);
      b2xy = fma2(splat2(w0i), T2xy, b2xy); b2zw = fma2(splat2(w0i), T2zw, b2zw);
    };
    if (!MPM_ABLATE(P, 4)) {
      plane(0, w0[0], e0[0]); plane(1, w0[1], e0[1]); plane(2, w0[2], e0[2]);
    }
    float v0 = vxy.x, v1 = vxy.y, v2 = vzw.x;
    mat3 b;
    b(0, 0) = b0xy.x; b(1, 0) = b0xy.y; b(2, 0) = b0zw.x;
    b(0, 1) = b1xy.x; b(1, 1) = b1xy.y; b(2, 1) = b1zw.x;
    b(0, 2) = b2xy.x; b(1, 2) = b2xy.y; b(2, 2) = b2zw.x;
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
    if (!MPM_ABLATE(P, 2)) plasticity_and_force<MATS, true>(g, cdg, F, aux, stress, reinterpret_cast<float *>(lane_lds));  // :950 + next substep's :509
    else stress = cdg;
    float nx0 = fmaf(v0, P.dt, x0), nx1 = fmaf(v1, P.dt, x1), nx2 = fmaf(v2, P.dt, x2);  // :951
    if (P.clamp_pos) {  // generic path only (optimized = false): p.pos clamped into [0, res - eps], :668-670
      nx0 = fminf(fmaxf(nx0 * P.idx, 0.0f), (float)P.res[0] - 1e-6f) * P.dx;
      nx1 = fminf(fmaxf(nx1 * P.idx, 0.0f), (float)P.res[1] - 1e-6f) * P.dx;
      nx2 = fminf(fmaxf(nx2 * P.idx, 0.0f), (float)P.res[2] - 1e-6f) * P.dx;
    }
    if (P.particle_collision) {  // particle_collision_resolution, src/mpm.cpp:414-426 (runs after G2P, :566-569)
      const LevelSetDev &LS = *ls;  // in device memory: by value it would sit in ~130 SGPRs for a rarely used path
      const float xw[3] = {nx0, nx1, nx2};
      float phi, gr[3] = {0, 0, 0};
      if (levelset_eval(LS, P.t, xw, P.idx, phi, gr) && phi < 0.0f) {
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
    key[pos] = kk;
    G0 = make_float4(nx0, nx1, nx2, aux);
    G1 = make_float4(F.m[0], F.m[1], F.m[2], F.m[3]);
    G2 = make_float4(F.m[4], F.m[5], F.m[6], F.m[7]);
    G3 = make_float4(F.m[8], g3.y, __int_as_float(pid), RIGID ? g3.w : 0.0f);  // (.w: the particle's CPIC colour word travels with it)
    Q0 = make_float4(nx0, nx1, nx2, v0);
    Q1 = make_float4(v1, v2, A[0], A[1]);
    Q2 = make_float4(A[2], A[3], A[4], A[5]);
    Q3 = make_float4(A[6], A[7], A[8], g.p[0]);
    if constexpr (STORE_B) {
      B0 = make_float4(b.m[0], b.m[1], b.m[2], b.m[3]);
      B1 = make_float4(b.m[4], b.m[5], b.m[6], b.m[7]);
      B2 = make_float4(b.m[8], 0.0f, 0.0f, 0.0f);
    }
    out_slot = MPM_ABLATE(P, 1) ? INVALID : pos;
}

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
      g2p_particle<MATS, STORE_B, RIGID>(P, scale, tile, ox, oy, oz, g0, g1, g2, g3, g, ls, cnt_w, key, cur.p + tid, bkey, out_slot, G0, G1,
                                         G2, G3, Q0, Q1, Q2, Q3, B0, B1, B2, xp + lane * 5);
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
    // deterministic mode: the creation id beside the key (mpm_common.h: Params::pidc).  Here, behind the record stores, not next to the
    // key's store in g2p_particle: there the two address registers raised the packed kernels from 167 to 169 VGPRs (a workgroup per CU less)
    if (P.pidc && out_slot != INVALID) P.pidc[out_slot] = __float_as_uint(G3.z);
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
