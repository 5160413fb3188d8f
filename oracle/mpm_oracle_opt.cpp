// oracle/mpm_oracle_opt.cpp — timed CPU baseline (TEST/BENCH INFRASTRUCTURE ONLY).
//
// Restates the *optimised* CPU path of the reference so that bench.py can quote a
// "restated reference algorithm (CPU)" number next to the GPU one:
//   sort_particles_and_populate_grid   src/mpm.cpp:770-918   (64-bit key sort, block meta, fat blocks, memset)
//   rasterize_optimized/block_op_normal src/transfer.cpp:467-569 (6x6x10 scratch tile, 8-colour blocks)
//   normalize_grid + boundary           src/mpm.cpp:277-372
//   resample_optimized/block_op_normal  src/transfer.cpp:837-954
// Block shape = SPGrid's 4x4x8 nodes (external/SPGrid/Core/SPGrid_Mask.h:29-35 with
// sizeof(GridState<3>)=32 B), blocks ordered by SPGrid's Morton (bit-interleaved, z-x-y) key, cells inside a
// block lexicographic with z fastest — the order `Linear_Offset` produces (SPGrid_Mask.h:141-148).
// Particles are AoS records reached through a sorted index array, physically reordered every
// `reorder_interval`(=1000) substeps like sort_allocator (src/mpm.cpp:752-768,811-813).
// Threads: OpenMP instead of TBB/ThreadedTaskManager.  This is NOT the reference binary.

#include "oracle_core.h"

#include <chrono>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>
#endif

using namespace orc;

namespace {

struct alignas(16) PRec {  // AoS particle (reference: MPMParticle<3>, src/particles.h:16-50)
  float pos[4];
  float v_and_m[4];
  float F[9];
  float B[9];
  float aux;
  int32_t gid;
  int32_t id;
  float pad;
};

constexpr int BX = 4, BY = 4, BZ = 8;            // nodes per block
constexpr int SX = BX + 2, SY = BY + 2, SZ = BZ + 2;  // scratch 6x6x10 (src/transfer.cpp:59-63)

inline uint32_t spread3(uint32_t v) {  // bit-interleave helper (pdep equivalent)
  uint32_t r = 0;
  for (int b = 0; b < 10; b++) r |= ((v >> b) & 1u) << (3 * b);
  return r;
}
// block index = the page bits of SparseMask::Linear_Offset: per level z is the most significant bit, then x, then y
// (page_zmask / page_xmask / page_ymask of SPGrid_Mask.h:29-35 for block_bits = 7).  PINNED against the reference's own
// header through oracle/_ref/spgrid_keys (tests/golden/spgrid_keys.txt, tests/test_oracle_kernel.py).
inline uint32_t morton_block(int bx, int by, int bz) { return (spread3(bz) << 2) | (spread3(bx) << 1) | spread3(by); }
inline uint64_t spgrid_key(int i, int j, int k) {  // Linear_Offset(i, j, k) >> data_bits  (src/mpm.cpp:785-790)
  return ((uint64_t)morton_block(i / BX, j / BY, k / BZ) << 7) | (uint64_t)(((i % BX) * BY + (j % BY)) * BZ + (k % BZ));
}

struct Opt {
  const orc_config *c;
  int nbx, nby, nbz;
  std::vector<float> grid;  // [block][BX][BY][BZ][4]
  inline int64_t bidx(int bx, int by, int bz) const { return ((int64_t)bx * nby + by) * nbz + bz; }
  inline float *node(int i, int j, int k) {
    return &grid[(bidx(i / BX, j / BY, k / BZ) * (BX * BY * BZ) + ((i % BX) * BY + (j % BY)) * BZ + (k % BZ)) * 4];
  }
};

}  // namespace

extern "C" uint64_t orc_spgrid_key(int i, int j, int k) { return spgrid_key(i, j, k); }

extern "C" double orc_opt_run(const orc_config *c, int64_t n, float *x, float *v, float *B, float *F, float *aux,
                              const int32_t *gid, const float *gparams, const int32_t *gtype, int steps,
                              int threads, double phase_s[4]) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  Opt o;
  o.c = c;
  o.nbx = (c->res[0] + 1 + BX - 1) / BX + 1;
  o.nby = (c->res[1] + 1 + BY - 1) / BY + 1;
  o.nbz = (c->res[2] + 1 + BZ - 1) / BZ + 1;
  const int64_t nblocks = (int64_t)o.nbx * o.nby * o.nbz;
  o.grid.assign((size_t)nblocks * BX * BY * BZ * 4, 0.0f);
  const real idx = 1.0f / c->dx, dt = c->dt;

  std::vector<PRec> pool(n), pool_(n);
  for (int64_t p = 0; p < n; p++) {
    PRec &r = pool[p];
    for (int k = 0; k < 3; k++) { r.pos[k] = x[3 * p + k]; r.v_and_m[k] = v[3 * p + k]; }
    r.pos[3] = 0; r.v_and_m[3] = gparams[ORC_NPARAM * gid[p]];
    std::memcpy(r.F, F + 9 * p, 36); std::memcpy(r.B, B + 9 * p, 36);
    r.aux = aux[p]; r.gid = gid[p]; r.id = (int32_t)p;
  }
  std::vector<uint32_t> particles(n), particles_(n);
  for (int64_t p = 0; p < n; p++) particles[p] = (uint32_t)p;
  std::vector<uint64_t> sorter(n);
  std::vector<int64_t> block_off;       // particle offset per active block (+ sentinel)
  std::vector<uint32_t> block_key;      // morton key of each active block
  std::vector<int32_t> block_coord;     // bx,by,bz per active block
  std::vector<uint8_t> fat(nblocks, 0);
  std::vector<int64_t> fat_list;
  std::vector<uint16_t> cell_count;     // per active block: 128 counts (GridState::particle_count)
  for (int k = 0; k < 4; k++) phase_s[k] = 0;
  const int index_bits = 25;            // src/mpm.cpp:773
  int64_t cur_n = n;

  auto t_begin = clk::now();
  for (int step = 0; step < steps; step++) {
    auto t0 = clk::now();
    // ---- sort_particles_and_populate_grid (src/mpm.cpp:770-918)
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < cur_n; i++) {
      const PRec &r = pool[particles[i]];
      int b[3];
      for (int k = 0; k < 3; k++) b[k] = stencil_start(r.pos[k] * idx);
      uint64_t key = spgrid_key(b[0], b[1], b[2]);
      sorter[i] = (key << index_bits) + (uint64_t)i;
    }
#ifdef _OPENMP
    __gnu_parallel::sort(sorter.begin(), sorter.begin() + cur_n);
#else
    std::sort(sorter.begin(), sorter.begin() + cur_n);
#endif
    std::swap(particles, particles_);
    for (int64_t i = 0; i < cur_n; i++) particles[i] = particles_[sorter[i] & ((1ull << index_bits) - 1)];  // serial, :805-807
    if (step % 1000 == 0) {  // sort_allocator, serial (:752-768)
      std::swap(pool, pool_);
      for (int64_t i = 0; i < cur_n; i++) { pool[i] = pool_[particles[i]]; particles[i] = (uint32_t)i; }
    }
    // block meta (serial scan, :876-889) + page map
    block_off.clear(); block_key.clear(); block_coord.clear();
    uint64_t last = ~0ull;
    for (int64_t i = 0; i < cur_n; i++) {
      uint64_t bk = sorter[i] >> (index_bits + 7);
      if (bk != last) {
        block_off.push_back(i); block_key.push_back((uint32_t)bk);
        const PRec &r = pool[particles[i]];
        block_coord.push_back(stencil_start(r.pos[0] * idx) / BX);
        block_coord.push_back(stencil_start(r.pos[1] * idx) / BY);
        block_coord.push_back(stencil_start(r.pos[2] * idx) / BZ);
        last = bk;
      }
    }
    block_off.push_back(cur_n);
    const int64_t nab = (int64_t)block_key.size();
    // fat page map = 3x3x3 dilation (:831-865) and memset of every fat block (:867-874)
    for (int64_t f : fat_list) fat[f] = 0;
    fat_list.clear();
    for (int64_t b = 0; b < nab; b++) {
      int bx = block_coord[3 * b], by = block_coord[3 * b + 1], bz = block_coord[3 * b + 2];
      for (int i = -1 + (bx == 0); i < 2; i++)
        for (int j = -1 + (by == 0); j < 2; j++)
          for (int k = -1 + (bz == 0); k < 2; k++) {
            if (bx + i >= o.nbx || by + j >= o.nby || bz + k >= o.nbz) continue;
            int64_t f = o.bidx(bx + i, by + j, bz + k);
            if (!fat[f]) { fat[f] = 1; fat_list.push_back(f); }
          }
    }
    for (int64_t f : fat_list) std::memset(&o.grid[(size_t)f * BX * BY * BZ * 4], 0, BX * BY * BZ * 16);
    // per-cell particle counts (:891-911)
    cell_count.assign((size_t)nab * 128, 0);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t b = 0; b < nab; b++)
      for (int64_t i = block_off[b]; i < block_off[b + 1]; i++) cell_count[b * 128 + ((sorter[i] >> index_bits) & 127)]++;
    auto t1 = clk::now();

    // ---- P2G: 8 colour passes over blocks (src/mpm.h:429-463, src/transfer.cpp:467-577)
    const real S = -4.0f * idx * dt;
    for (int colour = 0; colour < 8; colour++) {
#pragma omp parallel for schedule(dynamic, 4)
      for (int64_t b = 0; b < nab; b++) {
        int bx = block_coord[3 * b], by = block_coord[3 * b + 1], bz = block_coord[3 * b + 2];
        if ((bx & 1) != (colour & 1) || (by & 1) != ((colour >> 1) & 1) || (bz & 1) != ((colour >> 2) & 1)) continue;
        alignas(64) float cache[SX][SY][SZ][4];
        for (int i = 0; i < SX; i++)
          for (int j = 0; j < SY; j++)
            for (int k = 0; k < SZ; k++) std::memcpy(cache[i][j][k], o.node(bx * BX + i, by * BY + j, bz * BZ + k), 16);
        int64_t pe = block_off[b];
        for (int t = 0; t < 128; t++) {
          int64_t pb = pe;
          pe += cell_count[b * 128 + t];
          int cx = t >> 5, cy = (t >> 3) & 3, cz = t & 7;
          real basef[3] = {(real)(bx * BX + cx), (real)(by * BY + cy), (real)(bz * BZ + cz)};
          for (int64_t pi = pb; pi < pe; pi++) {
            PRec &r = pool[particles[pi]];
            const float *gp = gparams + ORC_NPARAM * r.gid;
            if (c->particle_gravity) for (int k = 0; k < 3; k++) r.v_and_m[k] += c->gravity[k] * dt;
            real rela[3];
            for (int k = 0; k < 3; k++) rela[k] = r.pos[k] * idx - basef[k];
            real w[3][3];
            for (int d = 0; d < 3; d++) quad_w_fma(rela[d] - 0.5f, w[d]);
            const real mass = r.v_and_m[3];
            M3 stress = calculate_force(gtype[r.gid], gp, load3(r.F), r.aux);
            real affine[9];
            for (int i = 0; i < 9; i++) affine[i] = std::fmaf(stress.a[i], S, r.B[i] * (4.0f * mass));
            real mass_v[3] = {mass * r.v_and_m[0], mass * r.v_and_m[1], mass * r.v_and_m[2]};
            for (int i = 0; i < 3; i++)
              for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) {
                  real d0 = rela[0] - i, d1 = rela[1] - j, d2 = rela[2] - k;
                  real weight = (w[0][i] * w[1][j]) * w[2][k];
                  float *g = cache[cx + i][cy + j][cz + k];
                  real contrib[4];
                  for (int rr = 0; rr < 3; rr++)
                    contrib[rr] = std::fmaf(affine[3 * rr + 2], d2, std::fmaf(affine[3 * rr + 1], d1, std::fmaf(affine[3 * rr], d0, mass_v[rr])));
                  contrib[3] = mass;
                  for (int rr = 0; rr < 4; rr++) g[rr] += weight * contrib[rr];
                }
          }
        }
        for (int i = 0; i < SX; i++)
          for (int j = 0; j < SY; j++)
            for (int k = 0; k < SZ; k++) std::memcpy(o.node(bx * BX + i, by * BY + j, bz * BZ + k), cache[i][j][k], 16);
      }
    }
    auto t2 = clk::now();

    // ---- grid normalise + boundary (src/mpm.cpp:277-372) over fat blocks
    real inc[3] = {0, 0, 0};
    if (!c->particle_gravity) for (int k = 0; k < 3; k++) inc[k] = c->gravity[k] * dt;
    const int64_t nfat = (int64_t)fat_list.size();
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t fi = 0; fi < nfat; fi++) {
      int64_t f = fat_list[fi];
      int bz = (int)(f % o.nbz), by = (int)((f / o.nbz) % o.nby), bx = (int)(f / ((int64_t)o.nbz * o.nby));
      float *blk = &o.grid[(size_t)f * BX * BY * BZ * 4];
      for (int e = 0; e < BX * BY * BZ; e++) {
        float *g = blk + 4 * e;
        if (g[3] > 0) {
          real im = 1.0f / g[3];
          for (int rr = 0; rr < 3; rr++) g[rr] = std::fmaf(g[rr], im, inc[rr]);
        }
        if (g[3] == 0.0f || c->n_planes <= 0) continue;
        real pos[3] = {(real)(bx * BX + e / (BY * BZ)), (real)(by * BY + (e / BZ) % BY), (real)(bz * BZ + e % BZ)}, phi, nrm[3] = {0, 0, 0};
        if (!levelset_eval(c, pos, phi, nrm)) continue;
        if (phi < -3 || 0 < phi) continue;
        real vb[3] = {0, 0, 0}, out[3], vel[3] = {g[0], g[1], g[2]};
        friction_project(vel, vb, nrm, c->friction, out);
        g[0] = out[0]; g[1] = out[1]; g[2] = out[2];
      }
    }
    auto t3 = clk::now();

    // ---- G2P (src/transfer.cpp:837-966), blocks in parallel, no colouring
    const real scale = -4.0f * idx * dt;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b = 0; b < nab; b++) {
      int bx = block_coord[3 * b], by = block_coord[3 * b + 1], bz = block_coord[3 * b + 2];
      alignas(64) float cache[SX][SY][SZ][4];
      for (int i = 0; i < SX; i++)
        for (int j = 0; j < SY; j++)
          for (int k = 0; k < SZ; k++) std::memcpy(cache[i][j][k], o.node(bx * BX + i, by * BY + j, bz * BZ + k), 16);
      int64_t pe = block_off[b];
      for (int t = 0; t < 128; t++) {
        int64_t pb = pe;
        pe += cell_count[b * 128 + t];
        int cx = t >> 5, cy = (t >> 3) & 3, cz = t & 7;
        real basef[3] = {(real)(bx * BX + cx), (real)(by * BY + cy), (real)(bz * BZ + cz)};
        for (int64_t pi = pb; pi < pe; pi++) {
          PRec &r = pool[particles[pi]];
          const float *gp = gparams + ORC_NPARAM * r.gid;
          real rela[3];
          for (int k = 0; k < 3; k++) rela[k] = r.pos[k] * idx - basef[k];
          real w[3][3];
          for (int d = 0; d < 3; d++) quad_w_fma(rela[d] - 0.5f, w[d]);
          real v_[3] = {0, 0, 0};
          M3 b_ = m3_zero();
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
              for (int k = 0; k < 3; k++) {
                real dpos[3] = {rela[0] - i, rela[1] - j, rela[2] - k};
                real weight = (w[0][i] * w[1][j]) * w[2][k];
                const float *g = cache[cx + i][cy + j][cz + k];
                for (int rr = 0; rr < 3; rr++) {
                  v_[rr] = std::fmaf(g[rr], weight, v_[rr]);
                  real wgv = weight * g[rr];
                  for (int cc = 0; cc < 3; cc++) b_(rr, cc) = std::fmaf(wgv, dpos[cc], b_(rr, cc));
                }
              }
          M3 bd = b_;
          if (c->rpic_damping != 0 || c->apic_damping != 0) {
            M3 sym = 0.5f * (b_ + transposed(b_));
            bd = (1 - c->rpic_damping) * sym + (1 - c->apic_damping) * (b_ - sym);
          }
          store3(bd, r.B);
          for (int k = 0; k < 3; k++) r.v_and_m[k] = v_[k];
          M3 cdg;
          for (int rr = 0; rr < 3; rr++)
            for (int cc = 0; cc < 3; cc++) cdg(rr, cc) = std::fmaf(scale, b_(rr, cc), (rr == cc) ? 1.0f : 0.0f);
          M3 Fm = load3(r.F);
          real a = r.aux;
          plasticity(gtype[r.gid], gp, cdg, Fm, a);
          store3(Fm, r.F);
          r.aux = a;
          for (int k = 0; k < 3; k++) r.pos[k] = std::fmaf(v_[k], dt, r.pos[k]);
        }
      }
    }
    // clear_boundary_particles (src/mpm.cpp:582-633): flags in parallel, serial compaction
    {
      int64_t m = 0;
      for (int64_t i = 0; i < cur_n; i++) {
        const PRec &r = pool[particles[i]];
        if (particle_alive(c, r.pos, r.v_and_m)) particles[m++] = particles[i];
      }
      cur_n = m;
    }
    auto t4 = clk::now();
    phase_s[0] += secs(t0, t1); phase_s[1] += secs(t1, t2); phase_s[2] += secs(t2, t3); phase_s[3] += secs(t3, t4);
  }
  double total = secs(t_begin, clk::now());
  // write back (by original id) so tests can compare against the plain oracle
  for (int64_t i = 0; i < cur_n; i++) {
    const PRec &r = pool[particles[i]];
    int64_t p = r.id;
    for (int k = 0; k < 3; k++) { x[3 * p + k] = r.pos[k]; v[3 * p + k] = r.v_and_m[k]; }
    std::memcpy(F + 9 * p, r.F, 36); std::memcpy(B + 9 * p, r.B, 36);
    aux[p] = r.aux;
  }
  return total;
}
