// k_grid_fused: the grid kernel of the fused G2P -> P2G experiment (was inside csrc/k_grid.h under MPMHIP_WITH_FUSED)
#ifdef MPMHIP_WITH_FUSED  // (experimental variant library only: k_g2p2g.h)
// ------------------------------------------------------------------------------------------------ grid, fused path
// normalize_grid_and_apply_external_force + apply_grid_boundary_conditions (src/mpm.cpp:277-372) when the P2G result lives
// in the 8^3-node tiles of k_g2p2g (k_g2p2g.h): tile of block s = nodes 4s-1 .. 4s+6, written by the PREVIOUS substep for
// the PREVIOUS sort's active blocks, which are found through that sort's bitmap (pbits / pprefix).  Work distribution, owner
// election and output (gridv slot 8a+o, fat_slot) are k_grid's.  A node (lx, ly, lz) of grid block g is covered, per axis,
// by the tile of g itself (local coordinate l + 1) and by ONE neighbour: g - 1 for l <= 2 (coordinate l + 5), g + 1 for
// l = 3 (coordinate 0) — eight tiles per node, as before; the candidates b + o of block b draw on b + {-1..2}^3, one
// previous-table lookup per lane.  Sums are taken in a fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void k_grid_fused(Params P, const Counters *__restrict__ cnt,
                                                    const uint32_t *__restrict__ act_blk, const uint32_t *__restrict__ bits,
                                                    const uint32_t *__restrict__ wprefix, const uint32_t *__restrict__ pbits,
                                                    const uint32_t *__restrict__ pprefix, const float4 *__restrict__ tiles8,
                                                    float4 *__restrict__ gridv, uint32_t *__restrict__ fat_slot,
                                                    LevelSetDev LS) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const int l = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
  for (uint32_t a = wave; a < na; a += nwaves) {
    int bx, by, bz;
    demorton3(act_blk[a], bx, by, bz);
    bool cur_active = false;  // lane n < 27: is block b + (n/9-1, n/3%3-1, n%3-1) active in THIS sort (owner election)
    if (l < 27) {
      const int sx = bx + l / 9 - 1, sy = by + (l / 3) % 3 - 1, sz = bz + l % 3 - 1;
      if (sx >= 0 && sy >= 0 && sz >= 0) {
        const uint32_t bk = morton3(sx, sy, sz);
        cur_active = block_active(bits, bk) && block_slot(bits, wprefix, bk) < P.max_blocks;
      }
    }
    const uint32_t amask = (uint32_t)__ballot(cur_active);
    uint32_t pslot = INVALID;  // lane = (dx+1, dy+1, dz+1) in {0..3}^3: tile slot of block b + d in the PREVIOUS sort
    {
      const int sx = bx + lx - 1, sy = by + ly - 1, sz = bz + lz - 1;
      if (sx >= 0 && sy >= 0 && sz >= 0 && sx < (1 << P.kbits) && sy < (1 << P.kbits) && sz < (1 << P.kbits)) {
        const uint32_t bk = morton3(sx, sy, sz);
        if (block_active(pbits, bk)) {
          const uint32_t ps = block_slot(pbits, pprefix, bk);
          if (ps < P.max_blocks) pslot = ps;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; o++) {
      const int ox = o >> 2, oy = (o >> 1) & 1, oz = o & 1;
      uint32_t lower = 0;
#pragma unroll
      for (int q = 0; q < o; q++) lower |= 1u << nb27(ox - (q >> 2), oy - ((q >> 1) & 1), oz - (q & 1));
      if (amask & lower) continue;  // wave-uniform: a candidate with a smaller offset owns this grid block
      const int cx = bx + ox, cy = by + oy, cz = bz + oz;
      const uint32_t slot = a * 8u + (uint32_t)o;
      const int gi = cx * BS + lx, gj = cy * BS + ly, gk = cz * BS + lz;
      float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int qx = q >> 2, qy = (q >> 1) & 1, qz = q & 1;
        const int dx = qx ? (lx <= 2 ? -1 : 1) : 0, dy = qy ? (ly <= 2 ? -1 : 1) : 0, dz = qz ? (lz <= 2 ? -1 : 1) : 0;
        const int tx = qx ? (lx <= 2 ? lx + 5 : 0) : lx + 1, ty = qy ? (ly <= 2 ? ly + 5 : 0) : ly + 1,
                  tz = qz ? (lz <= 2 ? lz + 5 : 0) : lz + 1;
        const uint32_t sslot = __shfl(pslot, ((ox + dx + 1) * 4 + (oy + dy + 1)) * 4 + (oz + dz + 1));
        if (sslot != INVALID) {
          const float4 t = tiles8[(size_t)sslot * 512 + (tx * 8 + ty) * 8 + tz];
          acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
      }
      float v[3] = {acc.x, acc.y, acc.z};
      const float m = acc.w;
      if (m > 0.0f) {  // src/mpm.cpp:282-292
        const float im = 1.0f / m;
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = fmaf(v[k], im, P.particle_gravity ? 0.0f : P.g[k] * P.dt);
      }
      if (m != 0.0f && LS.n > 0) {  // src/mpm.cpp:313-368
        const float xw[3] = {gi * P.dx, gj * P.dx, gk * P.dx};
        float phi, dphidt, nrm[3] = {0, 0, 0};
        levelset_eval(LS, P.t, xw, P.idx, phi, nrm, &dphidt);
        if (!(phi < -3.0f || 0.0f < phi)) {
          const float vb[3] = {-dphidt * nrm[0] * P.dx, -dphidt * nrm[1] * P.dx, -dphidt * nrm[2] * P.dx};
          friction_project(v, vb, nrm, LS.friction);
        }
      }
      gridv[(size_t)slot * BC + l] = make_float4(v[0], v[1], v[2], m);
      if (l == 0) fat_slot[morton3(cx, cy, cz)] = slot;
    }
  }
}

#endif  // MPMHIP_WITH_FUSED
