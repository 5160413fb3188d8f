// tests/cpp/async_sched_host.cpp — host build of the block scheduler of the asynchronous stepper (taichi_mpm_amd/csrc/async_sched.h:
// pure C++, the same header libmpmhip compiles) as a small shared library for tests/test_async_sched_cpu.py — no GPU needed.
#include "../../taichi_mpm_amd/csrc/async_sched.h"

extern "C" {
// the block table of a `dim`-dimensional grid: per block b (dense index) its corner node, its position in the reference's block
// order, its neighbours (26 slots, -1 terminated) and the left_boundary flag.  Returns the number of blocks.
long sched_geometry(int dim, const int *res, int left_boundary, long cap, int *coord, int *rank, int *neigh, int *bnd, int *nb) {
  mpm::AsyncSched S;
  mpmhip_async_config cfg{};
  cfg.unit_delta_t = 1e-6f; cfg.max_units = 8192; cfg.cfl_dt_mul = 1.0f; cfg.strength_dt_mul = 1.0f; cfg.left_boundary = left_boundary;
  S.sched_enable(dim, res, cfg);
  S.sched_begin();
  const long n = (long)S.nblk();
  for (int k = 0; k < 3; k++) nb[k] = S.nb[k];
  if (cap < n) return n;
  for (int bx = 0; bx < S.nb[0]; bx++)
    for (int by = 0; by < S.nb[1]; by++)
      for (int bz = 0; bz < S.nb[2]; bz++) {
        const long b = ((long)bx * S.nb[1] + by) * S.nb[2] + bz;
        coord[3 * b] = bx << S.shift[0]; coord[3 * b + 1] = by << S.shift[1]; coord[3 * b + 2] = dim == 3 ? bz << S.shift[2] : 0;
        rank[b] = (int)S.rank_of[b];
        for (int k = 0; k < 26; k++) neigh[26 * b + k] = k < S.max_neigh ? S.neigh[b * S.max_neigh + k] : -1;
        bnd[b] = 0;
      }
  for (uint32_t b : S.boundary) bnd[b] = 1;
  return n;
}

// `rounds` rounds of AsyncMPM::step's loop over a FIXED reduced table (tab[3 b] = {bits of min allowed dt, bits of max |v|^2, pool
// size}): update_dt_limits, then advance(level) for every level due (the action tables are built and checked, nothing is
// executed).  out_t: per block {continuous, particle_t, backup_t}; advances[64]: how often each level advanced.
// Returns current_t_int, or -1 with the message in err.
long sched_walk(int dim, const int *res, float unit_delta_t, int max_units, float dx, const uint32_t *tab, int rounds, long *out_t,
                long *advances, char *err, int err_cap) {
  mpm::AsyncSched S;
  mpmhip_async_config cfg{};
  cfg.unit_delta_t = unit_delta_t; cfg.max_units = max_units; cfg.cfl_dt_mul = 1.0f; cfg.strength_dt_mul = 1.0f; cfg.left_boundary = 0;
  S.sched_enable(dim, res, cfg);
  S.sched_begin();
  for (int k = 0; k < 64; k++) advances[k] = 0;
  for (int r = 0; r < rounds; r++) {
    S.scratch = S.continuous;
    if (!S.limits_from_table(tab, dx)) { snprintf(err, err_cap, "%s", S.sched_err.c_str()); return -1; }
    if (S.scratch != S.continuous) S.limits_version++;
    S.rebuild_lists();
    for (int64_t d = S.max_delta_t_int; d >= S.min_delta_t_int; d >>= 1)
      if (S.current_t_int % d == 0) {
        if (!S.plan_gather(d)) { snprintf(err, err_cap, "%s", S.sched_err.c_str()); return -1; }
        // a block is gathered from its pool OR its backup, never both; a block that steps backs its pool up
        for (size_t b = 0; b < S.nblk(); b++) {
          const uint8_t a = S.tbl[b];
          if ((a & mpm::AT_BACKUP) && (a & (mpm::AT_POOL0 | mpm::AT_POOL1))) { snprintf(err, err_cap, "block %zu gathered twice", b); return -1; }
          if (((a & mpm::AT_POOL1) != 0) != ((a & mpm::AT_SWAP) != 0)) { snprintf(err, err_cap, "block %zu: POOL1 without SWAP", b); return -1; }
        }
        S.plan_file(d);
        advances[mpm::AsyncSched::log2i(d)]++;
      }
    S.finish_round();
  }
  for (size_t b = 0; b < S.nblk(); b++) { out_t[3 * b] = S.continuous[b]; out_t[3 * b + 1] = S.particle_t[b]; out_t[3 * b + 2] = S.backup_t[b]; }
  return (long)S.current_t_int;
}
}
