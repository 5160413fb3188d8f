// taichi_mpm_amd/csrc/async_sched.h — the block scheduler of AsyncMPM<dim> (src/async/async_mpm.{h,cpp}), host side, for both
// dimensions.  Part of libmpmhip; pure C++ (no device code: tests/test_async_sched_cpu.py builds it with g++).
//
// AsyncMPM keeps, per scheduler block (an SPGrid block: 4 x 4 x 8 nodes in 3D, 8 x 16 nodes in 2D), three time-step limits
// (powers of two of unit_delta_t), the time of its particle pool and of its backup pool.  step() walks the power-of-two
// levels; advance(limit) gathers the pools of the level's blocks plus frozen copies of their neighbours, runs ONE ordinary
// substep and files the results back.  Everything that is a few integers per block lives here: the limit state machine
// (update_dt_limits, :90-164), the neighbour lists per level (:183-253), and the per-block ACTION TABLE of an advance
// (which pools to gather, re-tag, clear, and where results go: :255-373) that the device kernels of k_async.h (3D records)
// and k_async2d.h (2D arrays) execute.  The particle containers themselves never come here: they stay in HBM.
#pragma once
#include <xmmintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mpmhip.h"

namespace mpm {

// per-block action bits of one advance (host -> device table)
enum : uint8_t {
  AT_POOL0 = 1,        // gather the block's pool, first in line (a neighbour stepping with a smaller dt: ":263 Smaller neighbours")
  AT_POOL1 = 2,        // gather the block's pool (it steps with this advance's dt: ":277 Equal neighbours")
  AT_BACKUP = 4,       // gather the block's backup (a neighbour stepping with a larger dt: ":294 Larger neighbours")
  AT_SWAP = 8,         // backup_current_dt_limit (:317-325): drop the block's backup, its pool becomes the backup
  AT_CLEAR = 16,       // after the substep: drop the block's backup (:337-341)
  AT_DEST_POOL = 32,   // after the substep: results that land in this block go to its pool (:361-364)
  AT_DEST_BACKUP = 64  // ... to its backup (:365-371)
};

struct AsyncSched {
  mpmhip_async_config cfg{};
  int dim = 3;
  int nb[3] = {0, 0, 1};        // blocks per axis (block b = (bx nb[1] + by) nb[2] + bz; nb[2] = 1 in 2D)
  int shift[3] = {2, 2, 3};     // log2 of a block's extent in nodes (SPGrid_Mask<5, 5, dim>::block_{x,y,z}bits)
  int max_neigh = 26;
  std::vector<uint32_t> boundary;  // "left_boundary" blocks (src/async/async_mpm.cpp:43-53)
  int64_t current_t_int = 0, min_delta_t_int = 1, max_delta_t_int = 1;
  std::vector<int64_t> strength, cfl, continuous;  // strength_dt_limit, cfl_dt_limit, continuous_dt_limit
  std::vector<uint32_t> count;
  std::vector<int64_t> particle_t, backup_t, local_min;  // BlockInfo, src/async/async_mpm.h:93-110
  std::vector<uint8_t> has_copied, tbl;
  std::vector<int64_t> scratch;
  uint64_t limits_version = 0, lists_version = ~0ull;  // the neighbour lists are rebuilt only when a continuous limit changed
  std::vector<uint32_t> rank_of;                         // position of a block in the reference's block order
  std::vector<int32_t> neigh;                            // cached_neighbours: max_neigh per block, -1 terminated
  std::vector<std::vector<uint32_t>> larger, smaller;    // larger_neighbours / smaller_neighbours by log2(limit)
  int64_t update_counter = 0, step_counter = 0;
  float request_t = 0.0f, current_t = 0.0f;
  std::string sched_err;

  size_t nblk() const { return continuous.size(); }
  static uint64_t spread(uint32_t v, int stride) {  // bits of v `stride` apart
    uint64_t r = 0;
    for (int k = 0; k < 20; k++) r |= (uint64_t)((v >> k) & 1u) << (k * stride);
    return r;
  }
  static int log2i(int64_t v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
  static float inv_sqrt(float v) {  // src/async/async_mpm.cpp:77-80: the SSE reciprocal-square-root estimate
    return _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set1_ps(v)));
  }

  // AsyncMPM<dim>::initialize (src/async/async_mpm.cpp:13-55), the block table: strength = cfl = 2^31, continuous = 1
  void sched_enable(int dimension, const int *res, const mpmhip_async_config &c) {
    cfg = c;
    dim = dimension;
    if (dim == 3) { shift[0] = 2; shift[1] = 2; shift[2] = 3; max_neigh = 26; }
    else { shift[0] = 3; shift[1] = 4; shift[2] = 0; max_neigh = 8; }
    for (int k = 0; k < 3; k++) nb[k] = k < dim ? (res[k] >> shift[k]) + 1 : 1;
    const size_t n = (size_t)nb[0] * nb[1] * nb[2];
    strength.assign(n, 1ll << 31); cfl.assign(n, 1ll << 31); continuous.assign(n, 1);
    count.assign(n, 0);
    boundary.clear();
    if (cfg.left_boundary)  // :43-53 — blocks whose corner node lies in 0 <= x / res <= 0.2
      for (int bx = 0; bx < nb[0]; bx++) {
        if (!((float)(bx << shift[0]) / (float)res[0] <= 0.2f)) continue;
        for (int by = 0; by < nb[1]; by++)
          for (int bz = 0; bz < nb[2]; bz++) boundary.push_back((uint32_t)((bx * nb[1] + by) * nb[2] + bz));
      }
    current_t_int = 0; min_delta_t_int = 1; max_delta_t_int = 1;
  }

  // the rest of initialize for the resident stepper: block times, the reference's block order, cached_neighbours
  void sched_begin() {
    const size_t n = nblk();
    particle_t.assign(n, 0); backup_t.assign(n, 0); local_min.assign(n, 1);
    has_copied.assign(n, 0); tbl.assign(n, 0);
    larger.assign(64, {}); smaller.assign(64, {});
    update_counter = 0; step_counter = 0; request_t = 0.0f; current_t = 0.0f;
    // the reference's block number: the page bits of SparseMask::Linear_Offset — 3D: per level z, then x, then y
    // (external/SPGrid/Core/SPGrid_Mask.h:29-35, block_bits = 7); 2D: per level y, then x (:317-318); pools are walked in this order
    std::vector<std::pair<uint64_t, uint32_t>> order(n);
    for (int bx = 0; bx < nb[0]; bx++)
      for (int by = 0; by < nb[1]; by++)
        for (int bz = 0; bz < nb[2]; bz++) {
          const uint32_t b = ((uint32_t)bx * nb[1] + by) * nb[2] + bz;
          const uint64_t key = dim == 3 ? (spread(bz, 3) << 2) | (spread(bx, 3) << 1) | spread(by, 3) : (spread(by, 2) << 1) | spread(bx, 2);
          order[b] = {key, b};
        }
    std::sort(order.begin(), order.end());
    rank_of.assign(n, 0);
    for (size_t r = 0; r < n; r++) rank_of[order[r].second] = (uint32_t)r;
    // cached_neighbours (src/async/async_mpm.h:248-301): the blocks around a block
    neigh.assign(n * max_neigh, -1);
    for (int bx = 0; bx < nb[0]; bx++)
      for (int by = 0; by < nb[1]; by++)
        for (int bz = 0; bz < nb[2]; bz++) {
          const size_t b = ((size_t)bx * nb[1] + by) * nb[2] + bz;
          int m = 0;
          for (int i = -1; i < 2; i++)
            for (int j = -1; j < 2; j++)
              for (int k = (dim == 3 ? -1 : 0); k < (dim == 3 ? 2 : 1); k++) {
                const int x = bx + i, y = by + j, z = bz + k;
                if ((i || j || k) && x >= 0 && y >= 0 && z >= 0 && x < nb[0] && y < nb[1] && z < nb[2])
                  neigh[b * max_neigh + m++] = (int32_t)(((size_t)x * nb[1] + y) * nb[2] + z);
              }
        }
    limits_version = 0; lists_version = ~0ull;
  }

  // a block table read from a snapshot blob: every `continuous` limit must be a power of two in [1, 2^31] — limits_from_table's
  // doubling loop never ends on 0 at t = 0, and the alignment test (t & (limit - 1)) means nothing for other values
  static bool limits_are_sane(const char *continuous_bytes, size_t n_blocks) {
    for (size_t b = 0; b < n_blocks; b++) {
      int64_t v;
      memcpy(&v, continuous_bytes + 8 * b, 8);
      if (v < 1 || v > (1ll << 31) || (v & (v - 1)) != 0) return false;
    }
    return true;
  }

  // the block state machine of update_dt_limits (src/async/async_mpm.cpp:93-164) from the table the device reduced:
  // tab[3 b] = {bits of the smallest get_allowed_dt, bits of the largest |v|^2, number of pool containers}.  false: sched_err.
  bool limits_from_table(const uint32_t *tab, float dx) {
    const size_t n = nblk();
    const float inv_unit = 1.0f / cfg.unit_delta_t;
    const int64_t t = current_t_int;
    // non-empty blocks (:93-134)
    for (size_t b = 0; b < n; b++) {
      count[b] = tab[3 * b + 2];
      if (!count[b] || (t & (continuous[b] - 1)) != 0) continue;
      float min_dt, max_v2;
      memcpy(&min_dt, &tab[3 * b], 4); memcpy(&max_v2, &tab[3 * b + 1], 4);
      strength[b] = (int64_t)(cfg.strength_dt_mul * min_dt * inv_unit);
      cfl[b] = (int64_t)(cfg.cfl_dt_mul * dx * inv_unit * inv_sqrt(max_v2));
      const int64_t tmp = std::min(std::min(cfl[b], strength[b]), (int64_t)cfg.max_units);
      if (tmp < 1) {
        sched_err = "async stepping: a block's allowed time step is below unit_delta_t (particle types without a sound-speed bound, "
                    "e.g. linear / jelly, return 0: the reference stops here too, src/async/async_mpm.cpp:118-125)";
        return false;
      }
      int64_t &limit = continuous[b];
      while (tmp < limit) limit >>= 1;
      while (tmp >= (limit << 1) && (t & ((limit << 1) - 1)) == 0) limit <<= 1;
    }
    auto bounds = [&](bool non_empty) {  // update_dt_limit_boundary, src/async/async_mpm.h:178-189
      min_delta_t_int = 1ll << 31; max_delta_t_int = 1;
      for (size_t b = 0; b < n; b++) {
        if (non_empty && !count[b]) continue;
        min_delta_t_int = std::min(min_delta_t_int, continuous[b]);
        max_delta_t_int = std::max(max_delta_t_int, continuous[b]);
      }
    };
    bounds(true);
    for (size_t b = 0; b < n; b++) {  // empty blocks follow the largest step in use (:137-152)
      if (count[b] || (t & (continuous[b] - 1)) != 0) continue;
      int64_t &limit = continuous[b];
      while (max_delta_t_int < limit) limit >>= 1;
      while (max_delta_t_int >= (limit << 1) && (t & ((limit << 1) - 1)) == 0) limit <<= 1;
    }
    bounds(false);
    for (uint32_t b : boundary)  // "left_boundary" blocks follow the smallest step in use (:155-163; min / max stay as they are)
      if ((t & (continuous[b] - 1)) == 0) {
        int64_t &limit = continuous[b];
        while (min_delta_t_int < limit) limit >>= 1;
      }
    return true;
  }

  // larger / smaller neighbours per log2(limit) (:183-247), each in the reference's block order — they depend on the continuous
  // limits alone: rebuilt only when a limit has changed; then local_min_dt_limit (:165-182)
  void rebuild_lists() {
    const size_t n = nblk();
    if (lists_version != limits_version) {
      for (auto &v : larger) v.clear();
      for (auto &v : smaller) v.clear();
      for (size_t b = 0; b < n; b++) {
        const int64_t cb = continuous[b];
        for (int k = 0; k < max_neigh; k++) {
          const int32_t q = neigh[b * max_neigh + k];
          if (q < 0) break;
          if (cb < continuous[q]) {
            larger[log2i(cb)].push_back((uint32_t)q);
            smaller[log2i(continuous[q])].push_back((uint32_t)b);
          }
        }
      }
      auto tidy = [&](std::vector<uint32_t> &v) {
        std::sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return rank_of[x] < rank_of[y]; });
        v.erase(std::unique(v.begin(), v.end()), v.end());
      };
      for (auto &v : larger) tidy(v);
      for (auto &v : smaller) tidy(v);
      lists_version = limits_version;
    }
    // advance() reads local_min only for blocks it has copied with a LARGER limit than the level's, i.e. for members of the
    // larger-neighbour lists: computed for those (the value of the others is never looked at)
    for (const auto &v : larger)
      for (uint32_t b : v) {
        if (continuous[b] == min_delta_t_int) continue;
        int64_t m = 1ll << 31;
        for (int k = 0; k < max_neigh; k++) {
          const int32_t q = neigh[(size_t)b * max_neigh + k];
          if (q < 0) break;
          m = std::min(m, particle_t[q] + continuous[q]);
        }
        local_min[b] = m;
      }
  }

  // advance(limit), first half (:255-325): which pools the gather takes, which blocks back their pool up.  false: sched_err
  // (the reference's "particle_pool broken" / "backup_pool broken" assertions).
  bool plan_gather(int64_t limit) {
    const size_t n = nblk();
    const int64_t t = current_t_int;
    const int lg = log2i(limit);
    char msg[160];
    std::fill(has_copied.begin(), has_copied.end(), 0);
    std::fill(tbl.begin(), tbl.end(), 0);
    for (uint32_t b : smaller[lg]) {
      if (particle_t[b] != t) {
        snprintf(msg, sizeof msg, "async: particle_pool broken 2 (block %u at %lld, now %lld)", b, (long long)particle_t[b], (long long)t);
        sched_err = msg;
        return false;
      }
      has_copied[b] = 1; tbl[b] |= AT_POOL0;
    }
    for (size_t b = 0; b < n; b++)
      if (continuous[b] == limit) {
        if (particle_t[b] != t) {
          snprintf(msg, sizeof msg, "async: particle_pool broken 1 (block %zu)", b);
          sched_err = msg;
          return false;
        }
        tbl[b] |= AT_POOL1 | AT_SWAP;  // (backup_current_dt_limit: same condition, :317-325)
        backup_t[b] = t;
      }
    for (uint32_t b : larger[lg]) {
      if (backup_t[b] != t) {
        snprintf(msg, sizeof msg, "async: backup_pool broken (block %u at %lld, now %lld)", b, (long long)backup_t[b], (long long)t);
        sched_err = msg;
        return false;
      }
      has_copied[b] = 1; tbl[b] |= AT_BACKUP;
    }
    return true;
  }

  // advance(limit), second half (:331-343): update backup_t and particle_t, and where the results of the substep go.
  // Returns whether any backup is cleared.
  bool plan_file(int64_t limit) {
    const size_t n = nblk();
    const int64_t t = current_t_int;
    std::fill(tbl.begin(), tbl.end(), 0);
    bool any_clear = false;
    for (size_t b = 0; b < n; b++) {
      if (continuous[b] == limit) {
        particle_t[b] = t + limit;
        tbl[b] = AT_DEST_POOL;
      } else if (has_copied[b] && continuous[b] > limit && local_min[b] == t + limit) {
        backup_t[b] = t + limit;
        tbl[b] = AT_CLEAR | AT_DEST_BACKUP;
        any_clear = true;
      }
    }
    return any_clear;
  }

  // the end of one round of step() (:409-412)
  void finish_round() {
    current_t_int += min_delta_t_int - current_t_int % min_delta_t_int;
    current_t = cfg.unit_delta_t * (float)current_t_int;
  }
};

}  // namespace mpm
